"""The hot GEMM shapes of a cfg-3 step in isolation (B=8, 97x97): one line per shape with the TFLOP/s of the forward, the data
gradient (residual accumulate + ReLU mask) and the weight gradient -- the A/B harness of the experiments in DESIGN.md 5a (the
library reads its switches once per process: DASAC_STREAMK, DASAC_HYBRID, DASAC_WGRAD_QUAD, ...; 10 iterations per number:
compare within ONE run, boxes differ by a few per cent).  Usage (GPU box): python tools/gemm_exp.py [tag]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "da-sac_amd"))
import torch
from dasac_hip import ops

B, H, W = int(os.environ.get("DASAC_EXP_B", "8")), 97, 97
SHAPES = [  # name, cin, cout, branch, with residual epilogue
    ("1x1_256_1024+res", 256, 1024, (1, 1, 1, 0), True),
    ("1x1_1024_256", 1024, 256, (1, 1, 1, 0), False),
    ("3x3d2_256", 256, 256, (3, 3, 2, 2), False),
    ("3x3d4_512", 512, 512, (3, 3, 4, 4), False),
    ("1x1_512_2048+res", 512, 2048, (1, 1, 1, 0), True),
]


def timeit(fn, iters=10):
    fn()
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


if __name__ != "__main__":
    SHAPES = []
tag = sys.argv[1] if len(sys.argv) > 1 else "streamk=%s quad=%s" % (os.environ.get("DASAC_STREAMK", "1"), os.environ.get("DASAC_WGRAD_QUAD", "1"))
out = []
for name, cin, cout, br, with_res in SHAPES:
    spec = ops.ConvSpec(cin, cout, [br], 1)
    x = torch.randn(B, cin, H, W, device="cuda")
    w = [torch.randn(cout, cin, br[0], br[1], device="cuda") * 0.05]
    dz = torch.randn(B, cout, H, W, device="cuda")
    res = torch.randn(B, cout, H, W, device="cuda") if with_res else None
    shift = torch.randn(cout, device="cuda")
    flops = 2.0 * B * H * W * cout * spec.K
    o = ops.gemm_order(spec, False)
    tab, pk = ops.conv_table(spec, H, W, False, x.device, o), ops.conv_pack(spec, w, False, order=o)
    y = torch.empty(B, cout, H, W, device="cuda")
    tf = timeit(lambda: ops.conv_gemm(x, pk, tab, y, (H, W), 1, cout, spec.K, 1, shift, res, None, True))
    ot = ops.gemm_order(spec, True)
    tabt, pkt = ops.conv_table(spec, H, W, True, x.device, ot), ops.conv_pack(spec, w, True, order=ot)
    acc, msk = torch.randn_like(x), torch.randn_like(x)
    td = timeit(lambda: ops.conv_dgrad(spec, dz, None, (H, W), res=acc, mask=msk, table=tabt, packed=pkt))
    tabw = ops.conv_table(spec, H, W, False, x.device, 0)
    tw = timeit(lambda: ops.conv_wgrad(spec, dz, x, w, table=tabw))
    out.append("{:18s} fwd {:6.1f}  dgrad(res,mask) {:6.1f}  wgrad {:6.1f} TF".format(name, flops / tf / 1e12, flops / td / 1e12, flops / tw / 1e12))
print("== " + tag)
print("\n".join(out), flush=True)
