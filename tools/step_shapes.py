"""Per-shape breakdown of the GEMM kernels inside one cfg-3 step (which layers the time goes to).
Usage (GPU box): python tools/step_shapes.py [steps] [two|fused]  ->  table sorted by time (default: the two-call train.py order,
the bench headline since round 6; "fused" = SAC.forward_fused)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "da-sac_amd"))
import torch
import bench
import driver
from dasac_hip import ops
from models import get_model

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
FUSE = len(sys.argv) > 2 and sys.argv[2] == "fused"
dev = torch.device("cuda:0")
cfg = bench.model_cfg("deeplabv2_resnet101", False)
sys.stdout, out = open(os.devnull, "w"), sys.stdout
net = get_model(cfg, 0, num_classes=19, criterion=torch.nn.CrossEntropyLoss(ignore_index=255, reduction="none"))
driver.init_synthetic_weights(net, seed=0)
net.cuda(0).train()
net.running_conf.fill_(0.05)
opt = driver.make_optimizer(net, cfg)
src, tgt = driver.synthetic_batches(8, 2, 4, (769, 769), dev, seed=0)
src = (src[0], driver.self_consistent_labels(net, src[0]))
sys.stdout = out


def step(i):
    tgt_i = (tgt[0], tgt[1].clone(), tgt[2], tgt[3], tgt[4])
    return driver.sac_train_iteration(net, opt, src, tgt_i, 4, update_teacher=(i == 0), lr_target=cfg.LR_TARGET, fuse_passes=FUSE)


step(0)
ops.PROFILE.start()
for i in range(steps):
    step(1 + i)
prof = ops.PROFILE.stop(by_shape=True)
rows = sorted(prof.items(), key=lambda kv: -kv[1]["seconds"])
tot = sum(v["seconds"] for v in prof.values())
print("{:28s} {:>5s} {:>6s} {:>8s} {:>2s} {:>2s} {:>3s} {:>3s} {:>6s} {:>8s} {:>7s} {:>6s}".format(
    "kernel", "M", "K", "Npix", "s", "os", "res", "msk", "n/step", "ms/step", "TF", "cum%"))
cum = 0.0
for (name, tag), v in rows[:60]:
    cum += v["seconds"]
    M, K, Np, st, os_, res, msk = tag if tag else (0, 0, 0, 0, 0, False, False)
    print("{:28s} {:5d} {:6d} {:8d} {:2d} {:2d} {:3d} {:3d} {:6d} {:8.2f} {:7.1f} {:6.1f}".format(
        name, M, K, Np, st, os_, int(res), int(msk), v["launches"] // steps, v["seconds"] / steps * 1e3,
        v["flops"] / v["seconds"] / 1e12, 100 * cum / tot))
