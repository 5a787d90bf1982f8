"""Achieved HBM bandwidth of the SAC-head / pointwise kernels at the cfg-3 shape (8 crops, 19 classes, 769x769) and of the
SURVEY 8(f) kernels (device-side views at the reference's 512x1024 crop, validation counts, fused inference, SGD, teacher EMA):
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
# ---- SURVEY 8(f) kernels: next-1 views, next-2 optimiser / teacher EMA, next-3 validation, next-4 inference
import views as V
from dasac_hip.optim import FusedSGD
Hc, Wc, Lv = 512, 1024, 4                         # the reference's own target crop (configs: 512 x 1024, GROUP_SIZE 4)
tv = V.TargetViews((Hc, Wc), Lv, seed=3)
img = torch.randint(0, 256, (3, Hc, Wc), dtype=torch.uint8, device="cuda")
lab = torch.randint(0, 19, (Hc, Wc), dtype=torch.uint8, device="cuda")
vw = tv.sample()
# per crop pixel: 3 + 1 B read (+ resampling taps from L2), per view pixel 12 B frames + 8 B labels written (tf_target.py:33-239)
# (the kernel alone: TargetViews.make also builds the resampling tables on the host, ~1 ms of numpy per call -- timed as part of
# bench.py's ms_per_step_with_device_views, not here)
from dasac_hip import lib as L_
lib_ = L_.load()
tabs = torch.from_numpy(V.view_tables(vw, Hc, Wc)).cuda()
fr_o = torch.empty((Lv, 3, Hc, Wc), dtype=torch.float32, device="cuda")
gt_o = torch.empty((Lv, Hc, Wc), dtype=torch.int64, device="cuda")
row("make_views (4 views of 512x1024)", Hc * Wc * (4 + Lv * 20),
    lambda: L_.check(lib_.dasac_make_views(img.data_ptr(), lab.data_ptr(), 0, Hc, Wc, Lv, tabs.data_ptr(), tv.mean.ctypes.data,
                                            tv.std.ctypes.data, -1, fr_o.data_ptr(), gt_o.data_ptr(), 0, L_.stream_ptr()), "dasac_make_views"))
tvp = V.TargetViews((Hc, Wc), Lv, seed=3, blur=(.1, 2.), jitter=0.4, jitter_p=1.0, grey_p=0.0)
u8 = torch.randint(0, 256, (Lv, 3, Hc, Wc), dtype=torch.uint8, device="cuda")
gtv = torch.randint(0, 19, (Lv, Hc, Wc), dtype=torch.int64, device="cuda")
ph = tvp.sample_photometric()
# u8 views in (3 B), fp32 frames out (12 B) per view pixel; the blur's two box passes and the jitter chain stay in the workspace
row("view_photometric (blur + jitter, 4 views)", Lv * Hc * Wc * 15, lambda: tvp.augment(u8, gtv, ph))
row("iou_counts (argmax + tp/fp/fn)", T + P * 8, lambda: ops.iou_counts(up, y))
lut = torch.tensor(driver.CITYSCAPES_TRAIN_TO_ID, dtype=torch.uint8, device="cuda")
row("infer_labels (upsample+softmax+argmax+LUT)", low.numel() * 4 + P, lambda: ops.infer_labels(low, (H, H), lut))
# ResNet-101 DeepLabv2 parameter set: 43.9 M floats = 175.6 MB.  SGD: p, g read, momentum read + written, p written;
# EMA (update): student read, teacher read + written
nparam = 43_900_000
ps = [torch.nn.Parameter(torch.randn(n, device="cuda")) for n in (nparam // 2, nparam // 4, nparam // 4)]
for q_ in ps:
    q_.grad = torch.randn_like(q_)
sgd = FusedSGD([{"params": ps}], lr=1e-4, momentum=0.9, weight_decay=5e-4)
sgd.step()                                          # first step creates the momentum buffers (a different kernel path)
row("sgd_chunks (FusedSGD.step, 175.6 MB params)", 5 * nparam * 4, lambda: sgd.step())
fast = [q_.detach() for q_ in ps]
slow = [torch.randn_like(q_) for q_ in fast]
ema = ops.EmaPlan(fast, slow)
row("ema_chunks (EMA + distance)", 3 * nparam * 4, lambda: ema.run(0.99, True))
row("ema_chunks (distance only)", 2 * nparam * 4, lambda: ema.run(0.99, False))
if not ONLY:
    print("{:44s} {:>9s} {:>9s} {:>8s} {:>7s}".format("kernel", "MB", "us", "TB/s", "of 8"))
for name, nbytes, t in rows:
    print("{:44s} {:9.1f} {:9.1f} {:8.2f} {:6.0f}%".format(name, nbytes / 1e6, t * 1e6, nbytes / t / 1e12, 100 * nbytes / t / 8e12))
