import os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "da-sac_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from dasac_hip import ops
from gemm_exp import timeit
B, C, h, H = 8, 19, 97, 769
low = torch.randn(B, C, h, h, device="cuda") * 3
up, _, _ = ops.upsample_softmax(low, (H, H))
y = torch.randint(0, C, (B, H, H), device="cuda")
conf = torch.rand(B, 1, H, H, device="cuda")
cw = torch.rand(C, device="cuda")
for name, fn in [("fused mode1", lambda: ops.ce_loss_bwd_low(up, y, (h, h), cw, conf)), ("fused mode0", lambda: ops.ce_loss_bwd_low(up, y, (h, h), cw, None)),
                 ("ce_loss fwd", lambda: ops.ce_loss(up, y, cw, conf)), ("ce_loss fwd+grad", lambda: ops.ce_loss(up, y, cw, conf, want_grad=True))]:
    print(name, "%.1f us" % (timeit(fn) * 1e6))
