import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
sys.path.insert(0, "%s"); sys.path.insert(0, os.path.join("%s", "da-sac_amd"))
import torch
from dasac_hip import ops
spec = ops.ConvSpec(256, 256, [(3, 3, 2, 2)], 1)
x = torch.randn(8, 256, 97, 97, device="cuda"); w = torch.randn(256, 256, 3, 3, device="cuda") * 0.05
tab = ops.conv_table(spec, 97, 97, False, x.device); pk = ops.conv_pack(spec, [w], False)
y = torch.empty(8, 256, 97, 97, device="cuda")
f = lambda: ops.conv_gemm(x, pk, tab, y, (97, 97), 1, 256, spec.K)
f(); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): f()
b.record(); torch.cuda.synchronize()
t = a.elapsed_time(b) / 10 * 1e-3
print(os.environ.get("DASAC_LIB", "default").split("/")[-1], os.environ.get("DASAC_BK", "16"), "%%.3f ms %%.1f TF" %% (t * 1e3, 2.0 * 8 * 97 * 97 * 256 * 2304 / t / 1e12))
''' % (ROOT, ROOT)
for lib in [None, "libabl7.so", "libabl15.so", "libabl23.so", "libabl31.so"]:
    for bk in ("16", "32"):
        env = dict(os.environ, DASAC_BK=bk)
        if lib:
            env["DASAC_LIB"] = os.path.join(ROOT, "tools", lib)
        subprocess.run([sys.executable, "-c", code], env=env)
