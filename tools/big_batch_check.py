"""Does a 16-crop pass of VGG16-FCN8s at 512x1024 (activations of exactly 2 GiB) compute what two 8-crop passes compute?
Forward logits and every parameter gradient, full batch against the two halves.  Usage (GPU box): python tools/big_batch_check.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "da-sac_amd"))
import torch
import torch.nn as nn
import bench, driver, models
cfg = bench.model_cfg("fcn_vgg16_bn")
sys.stdout, out = open(os.devnull, "w"), sys.stdout
net = models.get_model(cfg, 0, num_classes=19, criterion=nn.CrossEntropyLoss(ignore_index=255, reduction="none"))
driver.init_synthetic_weights(net, seed=0)
sys.stdout = out
net.cuda(0).train()
for m in net.modules():
    if isinstance(m, nn.Dropout2d):
        m.p = 0.0
bb = net.backbone
g = torch.Generator(device="cuda").manual_seed(1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x = torch.randn(N, 3, 512, 1024, device="cuda", generator=g)
go = torch.randn(N, 19, 64, 128, device="cuda", generator=g)


def run(xs, gs):
    for p in bb.parameters():
        p.grad = None
    y = bb._logits(xs)
    (y * gs[:, :, :y.shape[2], :y.shape[3]]).sum().backward()
    return y.detach(), {n: p.grad.detach().clone() for n, p in bb.named_parameters() if p.grad is not None}


y_full, g_full = run(x, go)
h = N // 2
y_a, g_a = run(x[:h].contiguous(), go[:h].contiguous())
y_b, g_b = run(x[h:].contiguous(), go[h:].contiguous())
y_halves = torch.cat([y_a, y_b])
print("logits: max |full - halves| / max", float((y_full - y_halves).abs().max() / y_halves.abs().max()), "finite", bool(torch.isfinite(y_full).all()))
worst = 0.0
for n in g_full:
    ref = g_a[n] + g_b[n]
    e = float((g_full[n] - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
    worst = max(worst, e)
    if e > 1e-4:
        print("  grad", n, e)
print("worst gradient error over tensor max:", worst)
