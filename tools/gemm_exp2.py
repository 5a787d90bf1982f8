"""Short-K 1x1 conv (256 -> 1024 @97x97): what bounds it?  batch (working set vs the 256 MB Infinity Cache), residual on/off."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "da-sac_amd"))
import torch
from dasac_hip import ops
from gemm_exp import timeit

for (cin, cout) in ((256, 1024), (1024, 256), (512, 2048)):
    for B in (1, 2, 4, 8):
        H = W = 97
        spec = ops.ConvSpec(cin, cout, [(1, 1, 1, 0)], 1)
        x = torch.randn(B, cin, H, W, device="cuda")
        w = [torch.randn(cout, cin, 1, 1, device="cuda") * 0.05]
        res = torch.randn(B, cout, H, W, device="cuda")
        shift = torch.randn(cout, device="cuda")
        flops = 2.0 * B * H * W * cout * spec.K
        tab, pk = ops.conv_table(spec, H, W, False, x.device, 0), ops.conv_pack(spec, w, False)
        y = torch.empty(B, cout, H, W, device="cuda")
        t0 = timeit(lambda: ops.conv_gemm(x, pk, tab, y, (H, W), 1, cout, spec.K, 1, None, None, None, False))
        t1 = timeit(lambda: ops.conv_gemm(x, pk, tab, y, (H, W), 1, cout, spec.K, 1, shift, None, None, True))
        t2 = timeit(lambda: ops.conv_gemm(x, pk, tab, y, (H, W), 1, cout, spec.K, 1, shift, res, None, True))
        t3 = timeit(lambda: ops.conv_gemm(x, pk, tab, y, (H, W), 1, cout, spec.K, 1, shift, res, res, True))
        mb = (x.numel() + y.numel()) * 4 / 1e6
        print("{}->{} B={}  x+y {:5.0f} MB | plain {:6.1f}  shift+relu {:6.1f}  +res {:6.1f}  +res+mask {:6.1f} TF | plain {:.0f} us".format(
            cin, cout, B, mb, flops / t0 / 1e12, flops / t1 / 1e12, flops / t2 / 1e12, flops / t3 / 1e12, t0 * 1e6), flush=True)
