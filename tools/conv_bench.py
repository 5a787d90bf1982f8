"""Micro-benchmark of the implicit-GEMM conv kernels on the RN101 @769^2 layer shapes (B=8).
Usage (GPU box): python tools/conv_bench.py [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "da-sac_amd"))
import torch
from dasac_hip import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SHAPES = [
    # name, cin, cout, branches, stride, H, W, count in RN101
    ("stem7x7", 3, 64, [(7, 7, 1, 3)], 2, 769, 769, 1),
    ("l1_1x1_64_256", 64, 256, [(1, 1, 1, 0)], 1, 193, 193, 4),
    ("l1_1x1_256_64", 256, 64, [(1, 1, 1, 0)], 1, 193, 193, 2),
    ("l1_3x3_64", 64, 64, [(3, 3, 1, 1)], 1, 193, 193, 3),
    ("l2_3x3_128", 128, 128, [(3, 3, 1, 1)], 1, 97, 97, 4),
    ("l2_1x1_128_512", 128, 512, [(1, 1, 1, 0)], 1, 97, 97, 4),
    ("l3_1x1_1024_256", 1024, 256, [(1, 1, 1, 0)], 1, 97, 97, 22),
    ("l3_3x3_256_d2", 256, 256, [(3, 3, 2, 2)], 1, 97, 97, 23),
    ("l3_1x1_256_1024", 256, 1024, [(1, 1, 1, 0)], 1, 97, 97, 23),
    ("l4_1x1_2048_512", 2048, 512, [(1, 1, 1, 0)], 1, 97, 97, 2),
    ("l4_3x3_512_d4", 512, 512, [(3, 3, 4, 4)], 1, 97, 97, 3),
    ("l4_1x1_512_2048", 512, 2048, [(1, 1, 1, 0)], 1, 97, 97, 3),
    ("aspp", 2048, 19, [(3, 3, r, r) for r in (6, 12, 18, 24)], 1, 97, 97, 1),
]


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


tot = {"fwd": [0, 0], "dgrad": [0, 0], "wgrad": [0, 0]}
for name, cin, cout, br, stride, H, W, cnt in SHAPES:
    spec = ops.ConvSpec(cin, cout, br, stride)
    x = torch.randn(B, cin, H, W, device="cuda")
    ws = [torch.randn(cout, cin, b[0], b[1], device="cuda") * 0.05 for b in br]
    OH, OW = spec.out_hw(H, W)
    dz = torch.randn(B, cout, OH, OW, device="cuda")
    flops = 2.0 * B * OH * OW * cout * spec.K
    order = ops.gemm_order(spec, False)
    tab = ops.conv_table(spec, H, W, False, x.device, order)
    tabw = ops.conv_table(spec, H, W, False, x.device, 0)
    pk = ops.conv_pack(spec, ws, False, order=order)
    y = torch.empty(B, cout, OH, OW, device="cuda")
    tf = timeit(lambda: ops.conv_gemm(x, pk, tab, y, (OH, OW), stride, cout, spec.K))
    line = "{:18s} fwd {:7.3f} ms {:6.1f} TF".format(name, tf * 1e3, flops / tf / 1e12)
    tot["fwd"][0] += flops * cnt; tot["fwd"][1] += tf * cnt
    if name != "stem7x7":
        tabt = ops.conv_table(spec, OH, OW, True, x.device, ops.gemm_order(spec, True))
        pkt = ops.conv_pack(spec, ws, True, order=ops.gemm_order(spec, True))
        td = timeit(lambda: ops.conv_dgrad(spec, dz, ws, (H, W), table=tabt, packed=pkt))
        line += " | dgrad {:7.3f} ms {:6.1f} TF".format(td * 1e3, flops / td / 1e12)
        tot["dgrad"][0] += flops * cnt; tot["dgrad"][1] += td * cnt
    tw = timeit(lambda: ops.conv_wgrad(spec, dz, x, ws, table=tabw))
    line += " | wgrad {:7.3f} ms {:6.1f} TF".format(tw * 1e3, flops / tw / 1e12)
    tot["wgrad"][0] += flops * cnt; tot["wgrad"][1] += tw * cnt
    print(line, flush=True)
for k, (f, t) in tot.items():
    print("RN101 total {:6s}: {:8.1f} GFLOP/img  {:7.2f} ms/batch  {:6.1f} TF".format(k, f / B / 1e9, t * 1e3, f / t / 1e12))
