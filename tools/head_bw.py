"""Achieved HBM bandwidth of the SAC-head / pointwise kernels at the cfg-3 shape (8 crops, 19 classes, 769x769):
algorithmic bytes (each operand once) / time.  Usage (GPU box): python tools/head_bw.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "da-sac_amd"))
import torch
from dasac_hip import ops
import driver


def timeit(fn, iters=10):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


B, C, h, H = 8, 19, 97, 769
T = B * C * H * H * 4            # one [8,19,769,769] fp32 tensor
P = B * H * H
low = torch.randn(B, C, h, h, device="cuda") * 3
ign = torch.rand(B, H, H, device="cuda") < 0.05
up, probs, _ = ops.upsample_softmax(low, (H, H), ign, want_probs=True)
y = torch.randint(0, C, (B, H, H), device="cuda")
conf = torch.rand(B, 1, H, H, device="cuda")
cw = torch.rand(C, device="cuda")
ONLY = os.environ.get("HEAD_BW_ONLY", "")        # substring filter (tools/head_exp.py times one kernel through variant libraries)
rows = []


def row(name, nbytes, fn):
    if ONLY in name:
        rows.append((name, nbytes, timeit(fn)))


row("upsample_softmax (logits_up)", T, lambda: ops.upsample_softmax(low, (H, H)))
row("upsample_softmax (probs+sums+mask)", T + P, lambda: ops.upsample_softmax(low, (H, H), ign, want_up=False, want_probs=True, want_sums=True))
row("ce_loss forward (focal+conf)", T + P * 12, lambda: ops.ce_loss(up, y, cw, conf))
row("ce_loss backward -> low-res (fused)", T + P * 12, lambda: ops.ce_loss_bwd_low(up, y, (h, h), cw, conf))
row("ce_loss dlogits + upsample_bwd (old path)", 3 * T + P * 12, lambda: ops.upsample_bwd(ops.ce_loss(up, y, cw, conf, want_grad=True)[1], (h, h)))
row("pseudo_labels", T + P * (1 + 8 + 4), lambda: ops.pseudo_labels(probs, ign, 0.75, 0.2, cw))
theta, inv = driver.view_affines(driver.BENCH_VIEWS, H, H)
theta, inv = theta.repeat(2, 1, 1).cuda(), inv.repeat(2, 1, 1).cuda()
row("warp_pool (T=4, +aligned)", 2 * T + T // 4, lambda: ops.warp_pool(probs, theta, inv, 4))
pooled, mask, _ = ops.warp_pool(probs, theta, inv, 4)
row("warp_back", T // 4 + T, lambda: ops.warp_back(pooled, mask, inv, 4))
x = torch.randn(B, 64, 385, 385, device="cuda")
yp, arg = ops.maxpool_fwd(x, 3, 2, 1, True)
dy = torch.randn_like(yp)
row("maxpool_fwd 3x3/2 ceil", x.numel() * 4 + yp.numel() * 5, lambda: ops.maxpool_fwd(x, 3, 2, 1, True))
row("maxpool_bwd (+ReLU mask)", x.numel() * 4 + yp.numel() * 5, lambda: ops.maxpool_bwd(dy, yp, arg, (385, 385), 3, 2, 1, True))
a = torch.randn(B, 1024, 97, 97, device="cuda")
row("relu_mask", 3 * a.numel() * 4, lambda: ops.relu_mask(a, a))
if not ONLY:
    print("{:44s} {:>9s} {:>9s} {:>8s} {:>7s}".format("kernel", "MB", "us", "TB/s", "of 8"))
for name, nbytes, t in rows:
    print("{:44s} {:9.1f} {:9.1f} {:8.2f} {:6.0f}%".format(name, nbytes / 1e6, t * 1e6, nbytes / t / 1e12, 100 * nbytes / t / 8e12))
