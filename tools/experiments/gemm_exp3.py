"""K sweep at fixed tile count (1x1 conv, cout 1024, B=8, 97x97 -> 4712 tiles of 128x128): per-K-step cost vs per-tile overhead."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "da-sac_amd"))
import torch
from dasac_hip import ops
from gemm_exp import timeit

B, H, W = 8, 97, 97
for cout in (1024, 512):
    for cin in (64, 128, 256, 512, 1024, 2048):
        spec = ops.ConvSpec(cin, cout, [(1, 1, 1, 0)], 1)
        x = torch.randn(B, cin, H, W, device="cuda")
        w = [torch.randn(cout, cin, 1, 1, device="cuda") * 0.05]
        flops = 2.0 * B * H * W * cout * spec.K
        tab, pk = ops.conv_table(spec, H, W, False, x.device, 0), ops.conv_pack(spec, w, False)
        y = torch.empty(B, cout, H, W, device="cuda")
        t0 = timeit(lambda: ops.conv_gemm(x, pk, tab, y, (H, W), 1, cout, spec.K), 20)
        print("M={} K={:5d}  {:8.1f} us  {:6.1f} TF".format(cout, cin, t0 * 1e6, flops / t0 / 1e12), flush=True)
