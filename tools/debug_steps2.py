import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "da-sac_amd"))
import torch, torch.nn as nn
import bench, models, driver
size = int(sys.argv[1]); B = int(sys.argv[2])
cfg = bench.model_cfg()
net = models.get_model(cfg, 0, num_classes=19, criterion=nn.CrossEntropyLoss(ignore_index=255, reduction="none"))
driver.init_synthetic_weights(net, 0)
net.cuda().train(); net.running_conf.fill_(0.05)
optim = driver.make_optimizer(net, cfg)
src, tgt = driver.synthetic_batches(B, max(B // 4, 1), 4, (size, size), "cuda", seed=0)
src = (src[0], driver.self_consistent_labels(net, src[0]))
before = {k: v.clone() for k, v in net.backbone.state_dict().items()}
t = (tgt[0], tgt[1].clone(), tgt[2], tgt[3], tgt[4])
# source only
ls, _ = net(*src)
optim.zero_grad(); ls["loss_ce"].mean().backward()
gs = {n: p.grad.clone() for n, p in net.backbone.named_parameters()}
lt, outs = net(t[0], t[1], t[2], t[3], t[4], use_teacher=True, update_teacher=True, T=4)
(5.0 * lt["self_ce"].mean()).backward()
rows = []
for n, p in net.backbone.named_parameters():
    gt_ = p.grad - gs[n]
    rows.append((float(gt_.norm()), float(gs[n].norm()), float(p.detach().norm()), n))
rows.sort(reverse=True)
print("src loss", float(ls["loss_ce"]), "self_ce", float(lt["self_ce"]), "logit std", float(outs["logits"].std()))
print("largest target-grad norms: (tgt_grad, src_grad, weight_norm, name)")
for r in rows[:10]: print(r)
print("dlogits-related: logits_up abs max", float(outs["logits_up"].abs().max()), "conf mean", float(outs["teacher_conf"].mean()), "labels frac", float((outs["teacher_labels"]!=255).float().mean()))
