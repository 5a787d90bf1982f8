"""Which ATen ops (and how many) one two-call cfg-3-shaped iteration dispatches -- to find the ~437 `__amd_rocclr_copyBuffer` launches per
step of the kernel trace.  Small crops: the count does not depend on the size.  Usage: python tools/experiments/r6_count_copies.py"""
import os, sys, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "da-sac_amd"))
import torch, torch.nn as nn
from torch.utils._python_dispatch import TorchDispatchMode
import bench, driver, models
cfg = bench.model_cfg()
net = models.get_model(cfg, 0, num_classes=19, criterion=nn.CrossEntropyLoss(ignore_index=255, reduction="none"))
driver.init_synthetic_weights(net, seed=0)
net.cuda().train(); net.running_conf.fill_(0.05)
optim = driver.make_optimizer(net, cfg)
src, tgt = driver.synthetic_batches(2, 2, 4, (129, 129), "cuda", seed=0)
clone = lambda: (tgt[0], tgt[1].clone(), tgt[2], tgt[3], tgt[4])
for i in range(2):
    driver.sac_train_iteration(net, optim, src, clone(), 4, i == 0, cfg.LR_TARGET)
counts, where = collections.Counter(), collections.Counter()


class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__ if hasattr(func, "overloadpacket") else str(func)
        counts[name] += 1
        if name in ("copy_", "_to_copy", "clone", "cat", "fill_", "zero_", "zeros", "full", "add", "mul", "div"):
            fr = [f for f in traceback.extract_stack()[:-1] if "da-sac_amd" in f.filename or "bench.py" in f.filename]
            if fr:
                where[(name, os.path.basename(fr[-1].filename), fr[-1].lineno)] += 1
        return func(*args, **(kwargs or {}))


with Rec():
    driver.sac_train_iteration(net, optim, src, clone(), 4, False, cfg.LR_TARGET)
    torch.cuda.synchronize()
print(counts.most_common(25))
for k, v in where.most_common(25):
    print(v, k)
