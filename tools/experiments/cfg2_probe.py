"""cfg-2 probe (round 5): cost of two no-grad batch-statistics forwards of 2 crops against ONE of 4 crops -- the ceiling of the
cross-iteration forward fusion VERDICT r4 item 4 asks for.  Usage (GPU box): python tools/cfg2_probe.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "da-sac_amd"))
import torch, torch.nn as nn
import bench, driver, models
dev = torch.device("cuda:0")
cfg = bench.model_cfg("deeplabv2_resnet101", True)
net = models.get_model(cfg, 0, num_classes=19, criterion=nn.CrossEntropyLoss(**bench.CRITERION))
driver.init_synthetic_weights(net, seed=0)
net.cuda(0).train()
x2 = torch.randn(2, 3, 769, 769, device=dev)
x4 = torch.randn(4, 3, 769, 769, device=dev)
def fwd(x):
    with torch.no_grad():
        d = torch.zeros(x.shape[0], x.shape[2], x.shape[3], dtype=torch.int64, device=dev)
        net(x, d)
def t(fn, n=5):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
a = t(lambda: (fwd(x2), fwd(x2)))
b = t(lambda: fwd(x4))
print("two 2-crop no-grad train-BN forwards %.2f ms, one 4-crop %.2f ms" % (a, b))
