"""Where do the small device-to-device copies of a cfg-3 step come from?  Runs ONE step under torch.profiler and prints, for
every aten op that launched a Memcpy DtoD, the python call sites.  Usage (GPU box): python tools/count_copies.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "da-sac_amd"))
import collections
import torch
import bench, driver
from models import get_model

dev = torch.device("cuda:0")
cfg = bench.model_cfg("deeplabv2_resnet101", False)
sys.stdout, out = open(os.devnull, "w"), sys.stdout
net = get_model(cfg, 0, num_classes=19, criterion=torch.nn.CrossEntropyLoss(ignore_index=255, reduction="none"))
driver.init_synthetic_weights(net, seed=0)
net.cuda(0).train()
net.running_conf.fill_(0.05)
opt = driver.make_optimizer(net, cfg)
src, tgt = driver.synthetic_batches(2, 1, 2, (193, 193), dev, seed=0)
src = (src[0], driver.self_consistent_labels(net, src[0]))
sys.stdout = out


def step(i):
    tgt_i = (tgt[0], tgt[1].clone(), tgt[2], tgt[3], tgt[4])
    return driver.sac_train_iteration(net, opt, src, tgt_i, 2, update_teacher=(i == 0), lr_target=cfg.LR_TARGET)


step(0); step(1)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(2)
    torch.cuda.synchronize()
ev = prof.events()
cnt = collections.Counter()
for e in ev:
    n = e.name
    if n.startswith("aten::") and n in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::add_", "aten::add", "aten::fill_", "aten::zero_", "aten::zeros"):
        stack = [f for f in (e.stack or []) if "da-sac_amd" in f or "bench" in f or "autograd" in f][:2]
        cnt[(n, tuple(stack))] += 1
for (n, st), c in cnt.most_common(25):
    print(c, n, " <- ".join(s.strip()[-90:] for s in st))
print("memcpy kernels:", sum(1 for e in ev if "Memcpy" in e.name or "copyBuffer" in e.name))
