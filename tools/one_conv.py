"""Runs one conv shape repeatedly (for rocprofv3 --pmc).  Usage: python tools/one_conv.py <shape-name> <mode> [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "da-sac_amd"))
import torch
from dasac_hip import ops
SH = {"l3_3x3": (256, 256, [(3, 3, 2, 2)], 1, 97, 97), "l3_1x1a": (1024, 256, [(1, 1, 1, 0)], 1, 97, 97),
      "l3_1x1b": (256, 1024, [(1, 1, 1, 0)], 1, 97, 97), "l4_3x3": (512, 512, [(3, 3, 4, 4)], 1, 97, 97)}
name, mode = sys.argv[1], sys.argv[2]
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
cin, cout, br, stride, H, W = SH[name]
B = 8
spec = ops.ConvSpec(cin, cout, br, stride)
x = torch.randn(B, cin, H, W, device="cuda")
ws = [torch.randn(cout, cin, b[0], b[1], device="cuda") * 0.05 for b in br]
OH, OW = spec.out_hw(H, W)
dz = torch.randn(B, cout, OH, OW, device="cuda")
order = ops.gemm_order(spec, False) if mode == "fwd" else 0      # DASAC_PRECISION=bf16x3 selects the split-bf16 kernels
tab = ops.conv_table(spec, H, W, False, x.device, order); pk = ops.conv_pack(spec, ws, False, order=order)
y = torch.empty(B, cout, OH, OW, device="cuda")
for _ in range(iters):
    if mode == "fwd":
        ops.conv_gemm(x, pk, tab, y, (OH, OW), stride, cout, spec.K)
    else:
        ops.conv_wgrad(spec, dz, x, ws, table=tab)
torch.cuda.synchronize()
print("done")
