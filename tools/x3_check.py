"""Split-bf16 (bf16x3) conv GEMM against the exact-fp32 kernel: error and speed per layer shape.
Usage (GPU box): python tools/x3_check.py [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "da-sac_amd"))
import torch
import torch.nn.functional as F
from dasac_hip import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SHAPES = [
    ("stem7x7", 3, 64, [(7, 7, 1, 3)], 2, 769, 769),
    ("l1_1x1_64_256", 64, 256, [(1, 1, 1, 0)], 1, 193, 193),
    ("l1_3x3_64", 64, 64, [(3, 3, 1, 1)], 1, 193, 193),
    ("l2_1x1_s2", 256, 128, [(1, 1, 1, 0)], 2, 193, 193),
    ("l2_3x3_128", 128, 128, [(3, 3, 1, 1)], 1, 97, 97),
    ("l3_1x1_1024_256", 1024, 256, [(1, 1, 1, 0)], 1, 97, 97),
    ("l3_3x3_256_d2", 256, 256, [(3, 3, 2, 2)], 1, 97, 97),
    ("l3_1x1_256_1024", 256, 1024, [(1, 1, 1, 0)], 1, 97, 97),
    ("l4_3x3_512_d4", 512, 512, [(3, 3, 4, 4)], 1, 97, 97),
    ("l4_1x1_512_2048", 512, 2048, [(1, 1, 1, 0)], 1, 97, 97),
    ("aspp_expanded_720", 2048, 720, [(1, 1, 1, 0)], 1, 97, 97),
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


def run(mode, spec, x, ws, dz, H, W, OH, OW, shift, res, mask):
    ops.set_precision(mode)
    order, ordert = ops.gemm_order(spec, False), ops.gemm_order(spec, True)
    tab = ops.conv_table(spec, H, W, False, x.device, order)
    pk = ops.conv_pack(spec, ws, False, order=order)
    y = torch.empty(x.shape[0], spec.cout, OH, OW, device="cuda")
    f = lambda: ops.conv_gemm(x, pk, tab, y, (OH, OW), spec.stride, spec.cout, spec.K, shift=shift, res=res, relu=True)
    tf = timeit(f)
    out = [y.clone(), tf]
    if spec.cin >= 64:
        tabt = ops.conv_table(spec, OH, OW, True, x.device, ordert)
        pkt = ops.conv_pack(spec, ws, True, order=ordert)
        g = lambda: ops.conv_dgrad(spec, dz, ws, (H, W), table=tabt, packed=pkt, mask=mask)
        td = timeit(g)
        out += [g().clone(), td]
    else:
        out += [None, 0.0]
    tabw = ops.conv_table(spec, H, W, False, x.device, 0)
    h = lambda: ops.conv_wgrad(spec, dz, x, ws, table=tabw)
    tw = timeit(h)
    out += [h()[0].clone(), tw]
    return out


torch.manual_seed(0)
for name, cin, cout, br, stride, H, W in SHAPES:
    spec = ops.ConvSpec(cin, cout, br, stride)
    x = torch.randn(B, cin, H, W, device="cuda")
    ws = [torch.randn(cout, cin, b[0], b[1], device="cuda") * (2.0 / (cin * b[0] * b[1])) ** 0.5 for b in br]
    OH, OW = spec.out_hw(H, W)
    dz = torch.randn(B, cout, OH, OW, device="cuda")
    shift = torch.randn(cout, device="cuda") * 0.1
    res = torch.randn(B, cout, OH, OW, device="cuda")
    mask = torch.randn(B, cin, H, W, device="cuda")
    flops = 2.0 * B * OH * OW * cout * spec.K
    a = run("fp32", spec, x, ws, dz, H, W, OH, OW, shift, res, mask)
    b = run("bf16x3", spec, x, ws, dz, H, W, OH, OW, shift, res, mask)
    e_f = float((a[0] - b[0]).abs().max() / a[0].abs().max())
    line = "{:18s} fwd fp32 {:6.1f} TF  x3 {:6.1f} TF-eq ({:4.2f}x)  err {:.1e}".format(
        name, flops / a[1] / 1e12, flops / b[1] / 1e12, a[1] / b[1], e_f)
    if a[2] is not None:
        e_d = float((a[2] - b[2]).abs().max() / a[2].abs().max())
        line += " | dgrad fp32 {:6.1f}  x3 {:6.1f} ({:4.2f}x) err {:.1e}".format(flops / a[3] / 1e12, flops / b[3] / 1e12, a[3] / b[3], e_d)
    e_w = float((a[4] - b[4]).abs().max() / a[4].abs().max())
    line += " | wgrad fp32 {:6.1f}  x3 {:6.1f} ({:4.2f}x) err {:.1e}".format(flops / a[5] / 1e12, flops / b[5] / 1e12, a[5] / b[5], e_w)
    print(line, flush=True)

# against float64 on a small case
ops.set_precision("fp32")
spec = ops.ConvSpec(256, 256, [(3, 3, 2, 2)], 1)
x = torch.randn(1, 256, 33, 33, device="cuda")
w = torch.randn(256, 256, 3, 3, device="cuda") * (2.0 / 2304) ** 0.5
ref = F.conv2d(x.double().cpu(), w.double().cpu(), padding=2, dilation=2)
for mode in ("fp32", "bf16x3"):
    ops.set_precision(mode)
    y = ops.conv_forward(spec, x, [w])
    d = (y.double().cpu() - ref).abs()
    print("vs float64, 3x3 d2 256->256 @33x33: {:7s} max {:.2e}  rms {:.2e}  (ref rms {:.2e})".format(
        mode, float(d.max()), float(d.pow(2).mean().sqrt()), float(ref.pow(2).mean().sqrt())))
