"""wgrad_reduce alone: time of dasac_conv_wgrad_finish over a prepared slab workspace, per layer shape of cfg-3 (8- and 16-crop launches).
Usage: python tools/experiments/r6_wgrad_reduce_bw.py   (DASAC_LIB selects another build)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "da-sac_amd"))
import torch
from dasac_hip import lib as L
lib = L.load()
SHAPES = [("l3 3x3", 256, 256, 9, 23), ("l3 1x1a", 256, 1024, 1, 22), ("l3 1x1b", 1024, 256, 1, 23), ("l4 3x3", 512, 512, 9, 3),
          ("l4 1x1b", 2048, 512, 1, 3), ("l2 3x3", 128, 128, 9, 4), ("l2 1x1b", 512, 128, 1, 4)]
tot = {}
for crops in (8, 16):
    Npix = crops * 97 * 97
    for name, M, Cin, taps, count in SHAPES:
        K = Cin * taps
        nbytes = lib.dasac_conv_wgrad_workspace(crops, 97, 97, M, K)
        ws = torch.randn(nbytes // 4, device="cuda")
        w = torch.randn(M, Cin, 3 if taps == 9 else 1, 3 if taps == 9 else 1, device="cuda")
        dw = torch.empty_like(w)
        scale = torch.rand(M, device="cuda")
        rows = lib.dasac_conv_wgrad_dot_rows(Cin, taps)
        dot = torch.empty(rows, M, device="cuda")
        sums = torch.empty(M, device="cuda")
        run = lambda: L.check(lib.dasac_conv_wgrad_finish(ws.data_ptr(), crops, 97, 97, M, K, w.data_ptr(), scale.data_ptr(), dw.data_ptr(),
                                                          dot.data_ptr(), sums.data_ptr(), Cin, taps, 0, L.stream_ptr()), "finish")
        run(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            run()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / 50
        tot[crops] = tot.get(crops, 0.0) + us * count * (2 if crops == 8 else 1)
        print("{:2d} crops {:8s} M={:5d} K={:5d} slabs {:7.1f} MB  {:7.1f} us  {:5.2f} TB/s  checksum {:.6e}".format(
            crops, name, M, K, nbytes / 1e6, us, nbytes / us / 1e6, float(dw.double().sum())))
print("per step (these shapes): two-call {:.2f} ms, fused {:.2f} ms".format(tot[8] / 1e3, tot[16] / 1e3))
