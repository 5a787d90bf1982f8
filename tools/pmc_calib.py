"""Known-byte kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 in the access patterns this library uses
(MI355X_MICROARCH.md, HBM section: only the wide 16 B/lane stream is calibrated there).  Every tensor is 308 MB (> the
256 MB Infinity Cache), every kernel runs 5 times:
    relu_mask        dword loads (2 tensors) + dword stores (1 tensor), 64 lanes x 4 B contiguous   -> read 616.6 MB, write 308.3 MB
    add              the same with two dword read streams                                             -> read 616.6 MB, write 308.3 MB
    torch copy       dwordx4 loads + stores (ATen vectorised copy)                                    -> read 308.3 MB, write 308.3 MB
    upsample (up)    dwordx4 stores only, reads 5.7 MB                                                -> write 359.5 MB
Usage: rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o p -- python tools/pmc_calib.py   (and again with WRITE_SIZE)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "da-sac_amd"))
import torch
from dasac_hip import ops
a = torch.randn(8, 1024, 97, 97, device="cuda")
b = torch.randn(8, 1024, 97, 97, device="cuda")
c = torch.empty_like(a)
low = torch.randn(8, 19, 97, 97, device="cuda")
for _ in range(5):
    ops.relu_mask(a, b)
    ops.add(a, b, out=c)
    c.copy_(a)
    ops.upsample_softmax(low, (769, 769))
torch.cuda.synchronize()
print("done")
