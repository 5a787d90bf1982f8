"""Where the waves of the M-sweep kernel spend their time (diagnostic build with -DDASAC_TRACE_TILES; MI355X).
Usage: python tools/ms_timeline.py [mode] [extra -D flags]     modes of tools/one_conv.py: fwd | fwd_res | fwd_res_bits | dgrad_res_bits"""
import ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "da-sac_amd")
out = "/tmp/libdasac_mstrace.so"
srcs = sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")))
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
flags = sys.argv[2].split() if len(sys.argv) > 2 else []
o = "/tmp/ms_trace.o"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result",
                       "-DDASAC_TRACE_TILES"] + flags + ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc"), "-c",
                                                         os.path.join(PKG, "csrc", "gemm1x1_msweep.hip"), "-o", o], stderr=subprocess.DEVNULL)
objs = [os.path.join(PKG, "build", os.path.basename(s)[:-4] + ".o") for s in srcs if not s.endswith("gemm1x1_msweep.hip")] + [o]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
os.environ["DASAC_LIB"] = out
sys.path.insert(0, ROOT); sys.path.insert(0, PKG)
import numpy as np
import torch
from dasac_hip import ops
raw = ctypes.CDLL(out)
B, H = 16, 97
dgrad = mode.startswith("dgrad")
K, M = 256, 1024
spec = ops.ConvSpec(K, M, [(1, 1, 1, 0)], 1)
x = torch.randn(B, K, H, H, device="cuda")
w = torch.randn(M, K, 1, 1, device="cuda") * 0.05
order = ops.gemm_order(spec, False)
tab, pk = ops.conv_table(spec, H, H, False, x.device, order), ops.conv_pack(spec, [w], False, order=order)
y = torch.empty(B, M, H, H, device="cuda")
res = torch.randn_like(y) if "res" in mode else None
bits = ops.ReluBits(B, M, H, H, x.device) if "bits" in mode else None
if dgrad:
    bits.words.random_(-2 ** 31, 2 ** 31 - 1)
    run = lambda: ops.conv_gemm(x, pk, tab, y, (H, H), 1, M, K, 1, None, res, bits, False)
else:
    run = lambda: ops.conv_gemm(x, pk, tab, y, (H, H), 1, M, K, 1, None, res, None, res is not None, bits_out=bits)
run(); torch.cuda.synchronize()
trace = torch.zeros(256 * 8 * 12, dtype=torch.int64, device="cuda")
assert raw.dasac_debug_set_ms_trace(ctypes.c_void_p(trace.data_ptr())) == 0
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); run(); b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) * 1e3
t = trace.cpu().numpy().reshape(256, 8, 12).astype(np.float64)
tick = us / t[:, :, 4].max()          # the longest-living wave spans (almost) the whole launch
print("{} {}: launch {:.1f} us, {:.1f} TFLOP/s; passes per wave {:.0f}..{:.0f}".format(
    mode, " ".join(flags), us, 2.0 * B * H * H * M * K / us / 1e6, t[:, :, 6].min(), t[:, :, 6].max()))
for nm, k in (("waiting for a tile", 0), ("waiting for the matrix pipe", 1), ("K loop", 2), ("publish + prefetch issue", 3), ("epilogue, pixel columns 0-31", 7),
              ("epilogue, pixel columns 32-63", 8), ("wave lifetime", 4)):
    v = t[:, :, k] * tick
    print("  {:32s} mean {:7.1f}  p10 {:7.1f}  p50 {:7.1f}  p90 {:7.1f}  max {:7.1f} us   per pass {:6.2f} us".format(
        nm, v.mean(), *np.percentile(v, [10, 50, 90]), v.max(), (v / t[:, :, 6]).mean()))
print("  SIMD ids of waves 0..7 in workgroup 0:", t[0, :, 5].astype(int).tolist())
