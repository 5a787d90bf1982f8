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
                       "-DDASAC_TRACE_TILES"] + os.environ.get("DASAC_TRACE_DEFS", "").split() + ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc"), "-shared"] + srcs + ["-o", out],
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
HW_ARG = int(sys.argv[4]) if len(sys.argv) > 4 else 0           # plane side (default 97: cfg-3's stride-8 maps)
SH = {"l3_3x3": (256, 256, [(3, 3, 2, 2)], 97), "l3_1x1a": (1024, 256, [(1, 1, 1, 0)], 97), "l3_1x1b": (256, 1024, [(1, 1, 1, 0)], 97)}
cin, cout, br, H = SH[name]
H = HW_ARG or H
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
us = a.elapsed_time(b) * 1e3
# s_memtime counters of different XCDs have different bases: normalise per XCC, calibrate the tick on the per-XCC launch span
xcc = t[:, 4].astype(np.int64)
spans = []
for c in np.unique(xcc):
    m = xcc == c
    base = t[m, 0].min()
    t[m, :4] -= base
    t[m, 5] -= base
    spans.append(t[m, 3].max())
tick = us / float(np.median(spans))
st, ld, kl, ep = t[:, 0] * tick, (t[:, 1] - t[:, 0]) * tick, (t[:, 2] - t[:, 1]) * tick, (t[:, 3] - t[:, 2]) * tick
# the counters are not comparable across CUs; only differences inside one workgroup are used below, scaled so that the mean
# workgroup lifetime equals launch time x resident slots / workgroups (all slots busy)
life = (t[:, 3] - t[:, 0]).astype(np.float64)
tick = (us * min(1024, len(t)) / len(t)) / life.mean()
st, ld, kl, ep = t[:, 0] * tick, (t[:, 1] - t[:, 0]) * tick, (t[:, 2] - t[:, 1]) * tick, (t[:, 3] - t[:, 2]) * tick
ep_issue, ep_ack = (t[:, 5] - t[:, 2]) * tick, (t[:, 3] - t[:, 5]) * tick      # round 5: loads + arithmetic + store issue | waiting for the acknowledgements
print("{} {} B={} {}x{}: {} workgroups, launch {:.1f} us, {:.1f} TFLOP/s".format(name, mode, B, H, H, len(t), us, 2.0 * B * H * H * cout * spec.K / us / 1e6))
for nm, v in (("prologue (first tile -> LDS)", ld), ("K loop", kl), ("epilogue (stores acknowledged)", ep), ("  of which: until stores issued", ep_issue),
              ("  of which: waiting for acks", ep_ack), ("whole workgroup", ld + kl + ep)):
    print("  {:32s} mean {:7.2f}  p10 {:7.2f}  p50 {:7.2f}  p90 {:7.2f}  max {:7.2f} us".format(nm, v.mean(), *np.percentile(v, [10, 50, 90]), v.max()))
print("  sum over workgroups / (launch x 1024 slots): prologue {:.3f}  K loop {:.3f}  epilogue {:.3f}".format(
    ld.sum() / (us * 1024), kl.sum() / (us * 1024), ep.sum() / (us * 1024)))
