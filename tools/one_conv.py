"""Runs one conv shape repeatedly (for rocprofv3 --pmc / --kernel-trace).
Usage: python tools/one_conv.py <shape-name> <mode> [iters] [batch]
modes: fwd | fwd_res (residual + ReLU epilogue) | fwd_res_bits (same, recording the ReLU bit mask) | dgrad_res_mask (data gradient
with residual + fp32 mask) | dgrad_res_bits (same with the bit mask) | wgrad.  Prints the algorithmic MB per launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "da-sac_amd"))
import torch
from dasac_hip import ops
SH = {"l3_3x3": (256, 256, [(3, 3, 2, 2)], 1, 97, 97), "l3_1x1a": (1024, 256, [(1, 1, 1, 0)], 1, 97, 97),
      "l3_1x1b": (256, 1024, [(1, 1, 1, 0)], 1, 97, 97), "l4_3x3": (512, 512, [(3, 3, 4, 4)], 1, 97, 97)}
name, mode = sys.argv[1], sys.argv[2]
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
B = int(sys.argv[4]) if len(sys.argv) > 4 else 8
cin, cout, br, stride, H, W = SH[name]
spec = ops.ConvSpec(cin, cout, br, stride)
torch.manual_seed(0)
x = torch.randn(B, cin, H, W, device="cuda")
ws = [torch.randn(cout, cin, b[0], b[1], device="cuda") * 0.05 for b in br]
OH, OW = spec.out_hw(H, W)
dz = torch.randn(B, cout, OH, OW, device="cuda")
MB = lambda *ts: sum(t.numel() * t.element_size() for t in ts) / 1e6
if mode.startswith("fwd"):
    order = ops.gemm_order(spec, False)
    tab, pk = ops.conv_table(spec, H, W, False, x.device, order), ops.conv_pack(spec, ws, False, order=order)
    y = torch.empty(B, cout, OH, OW, device="cuda")
    res = torch.randn_like(y) if "res" in mode else None
    bits = ops.ReluBits(B, cout, OH, OW, x.device) if "bits" in mode else None
    alg = MB(x, pk, y) + (MB(res) if res is not None else 0) + (MB(bits.words) if bits is not None else 0)
    run = lambda: ops.conv_gemm(x, pk, tab, y, (OH, OW), stride, cout, spec.K, 1, None, res, None, res is not None, bits_out=bits)
elif mode.startswith("dgrad"):
    order = ops.gemm_order(spec, True)
    tab, pk = ops.conv_table(spec, OH, OW, True, x.device, order), ops.conv_pack(spec, ws, True, order=order)
    res = torch.randn(B, cin, H, W, device="cuda")
    act = torch.randn(B, cin, H, W, device="cuda")
    mask = act
    if "bits" in mode:                       # record the pattern of `act` with a ReLU forward of an identity-like conv: fill the words directly
        mask = ops.ReluBits(B, cin, H, W, x.device)
        mask.words.random_(-2 ** 31, 2 ** 31 - 1)
    alg = MB(dz, pk, res, res) + (MB(mask.words) if "bits" in mode else MB(act))
    run = lambda: ops.conv_dgrad(spec, dz, None, (H, W), res=res, mask=mask, table=tab, packed=pk)
else:
    tab = ops.conv_table(spec, H, W, False, x.device, 0)
    alg = MB(dz, x) + sum(MB(w) for w in ws)
    run = lambda: ops.conv_wgrad(spec, dz, x, ws, table=tab)
run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(iters):
    run()
b.record()
torch.cuda.synchronize()
out = run()
out = out[0] if isinstance(out, (list, tuple)) else out
print("done {} {} B={} algorithmic_MB_per_launch {:.1f} us_per_call {:.1f} checksum {:.6e} {:.6e}".format(
    name, mode, B, alg, a.elapsed_time(b) * 1e3 / iters, float(out.double().sum()), float(out.double().abs().sum())))
