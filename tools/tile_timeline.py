"""Per-workgroup timeline of the tile-per-block conv GEMM on one layer shape (diagnostic; MI355X).
Builds a SEPARATE library with -DDASAC_TRACE_TILES (four s_memtime stamps per workgroup: start, first tile in LDS, end of the K
loop, epilogue stores acknowledged), runs one launch and prints how long the phases take and how many workgroups are in each
phase over time.  Usage: python tools/tile_timeline.py [shape] [mode] [batch]      (shapes / modes of tools/one_conv.py)"""
import ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "da-sac_amd")
out = "/tmp/libdasac_trace.so"
srcs = sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")))
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result",
                       "-DDASAC_TRACE_TILES", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc"), "-shared"] + srcs + ["-o", out],
                      stderr=subprocess.DEVNULL)
os.environ["DASAC_LIB"] = out
sys.path.insert(0, ROOT); sys.path.insert(0, PKG)
import numpy as np
import torch
from dasac_hip import ops, lib as L
raw = ctypes.CDLL(out)
name = sys.argv[1] if len(sys.argv) > 1 else "l3_1x1b"
mode = sys.argv[2] if len(sys.argv) > 2 else "fwd_res"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
SH = {"l3_3x3": (256, 256, [(3, 3, 2, 2)], 97), "l3_1x1a": (1024, 256, [(1, 1, 1, 0)], 97), "l3_1x1b": (256, 1024, [(1, 1, 1, 0)], 97)}
cin, cout, br, H = SH[name]
spec = ops.ConvSpec(cin, cout, br, 1)
x = torch.randn(B, cin, H, H, device="cuda")
ws = [torch.randn(cout, cin, b[0], b[1], device="cuda") * 0.05 for b in br]
order = ops.gemm_order(spec, False)
tab, pk = ops.conv_table(spec, H, H, False, x.device, order), ops.conv_pack(spec, ws, False, order=order)
y = torch.empty(B, cout, H, H, device="cuda")
res = torch.randn_like(y) if "res" in mode else None
run = lambda: ops.conv_gemm(x, pk, tab, y, (H, H), 1, cout, spec.K, 1, None, res, None, res is not None)
run(); torch.cuda.synchronize()
n_blocks = ((B * H * H + 127) // 128 + 7) // 8 * 8 * ((cout + 127) // 128)
trace = torch.zeros(n_blocks * 8, dtype=torch.int64, device="cuda")
assert raw.dasac_debug_set_tile_trace(ctypes.c_void_p(trace.data_ptr())) == 0
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); run(); b.record(); torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(-1, 8)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
span = (t[:, 3].max() - t0)
us = a.elapsed_time(b) * 1e3
tick = us / span                     # microseconds per s_memtime tick, from the launch's own duration
st, ld, kl, ep = (t[:, 0] - t0) * tick, (t[:, 1] - t[:, 0]) * tick, (t[:, 2] - t[:, 1]) * tick, (t[:, 3] - t[:, 2]) * tick
print("{} {} B={}: {} workgroups, launch {:.1f} us ({:.4f} us/tick)".format(name, mode, B, len(t), us, tick))
for nm, v in (("prologue (first tile -> LDS)", ld), ("K loop", kl), ("epilogue (stores acknowledged)", ep), ("whole workgroup", ld + kl + ep)):
    print("  {:32s} mean {:7.2f}  p10 {:7.2f}  p50 {:7.2f}  p90 {:7.2f}  max {:7.2f} us".format(nm, v.mean(), *np.percentile(v, [10, 50, 90]), v.max()))
# occupancy of the phases over time: how many workgroups sit in prologue / K loop / epilogue at 40 sample times
ts = np.linspace(0, us, 41)[1:-1]
print("  time us : in prologue / in K loop / in epilogue  (of {} resident slots)".format(256 * 4))
for x_ in ts[::3]:
    tt = x_ / tick + t0
    print("  {:8.1f} : {:5d} {:5d} {:5d}".format(x_, int(((t[:, 0] <= tt) & (tt < t[:, 1])).sum()), int(((t[:, 1] <= tt) & (tt < t[:, 2])).sum()),
                                                  int(((t[:, 2] <= tt) & (tt < t[:, 3])).sum())))
xcc = t[:, 4]
print("  block id % 8 == hardware XCC id for {:.1f} % of the workgroups".format(100.0 * float((xcc == (np.arange(len(trace) // 8)[trace.cpu().numpy().reshape(-1, 8)[:, 0] > 0] % 8)).mean())))
