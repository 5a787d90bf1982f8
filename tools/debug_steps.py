import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "da-sac_amd"))
import torch, torch.nn as nn
import bench, models, driver
size = int(sys.argv[1]) if len(sys.argv) > 1 else 769
cfg = bench.model_cfg()
net = models.get_model(cfg, 0, num_classes=19, criterion=nn.CrossEntropyLoss(ignore_index=255, reduction="none"))
driver.init_synthetic_weights(net, 0, float(os.environ.get('GAIN', '6.0')))
cfg.LR *= float(os.environ.get('LRS', '1'))
net.cuda().train(); net.running_conf.fill_(0.05)
optim = driver.make_optimizer(net, cfg)
src, tgt = driver.synthetic_batches(8, 2, 4, (size, size), "cuda", seed=0)
src = (src[0], driver.self_consistent_labels(net, src[0]))
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    t = (tgt[0], tgt[1].clone(), tgt[2], tgt[3], tgt[4])
    ls, lt, outs = driver.sac_train_iteration(net, optim, src, t, 4, i == 0, cfg.LR_TARGET)
    gn = sum(float(p.grad.norm()) ** 2 for p in net.backbone.parameters()) ** 0.5
    pm = max(float(p.abs().max()) for p in net.backbone.parameters())
    lg = outs["logits"]
    print(i, "src", float(ls["loss_ce"]), {k: float(v) for k, v in lt.items()}, "gradnorm", gn, "pmax", pm, "logit std", float(lg.std()), "lab", float((outs["teacher_labels"] != 255).float().mean()), flush=True)
