"""Inference throughput of the fused path (infer_val.py:160-163 + argmax + id LUT): full-resolution Cityscapes frames
(1024x2048, batch 1 as the reference's loader) through RN101-DeepLabv2 -> dasac_infer_labels.
Usage (GPU box): python tools/infer_bench.py [precision] [H W]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "da-sac_amd"))
import torch
import torch.nn as nn
import bench
import driver
import models
from dasac_hip import ops

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1024, 2048)
out, sys.stdout = sys.stdout, open(os.devnull, "w")
net = models.get_model(bench.model_cfg(), 0, num_classes=19, criterion=nn.CrossEntropyLoss(ignore_index=255, reduction="none"))
driver.init_synthetic_weights(net, seed=0)
net.cuda(0).eval()
sys.stdout = out
ops.set_precision(prec)
x = torch.randn(1, 3, H, W, device="cuda")
for name, fn in (("fused labels", lambda: driver.infer_label_maps(net, x, lut=driver.CITYSCAPES_TRAIN_TO_ID)),
                 ("reference-style (logits_up, softmax, argmax)", lambda: net(x, teacher=False)[1].softmax(1).argmax(1))):
    with torch.no_grad():
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("{:8s} {}x{} {:46s} {:7.1f} ms/frame  {:6.2f} frames/s".format(prec, H, W, name, dt * 1e3, 1.0 / dt))
