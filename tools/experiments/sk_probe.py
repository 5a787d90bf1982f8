"""Fixed cost of a stream-K remainder launch: time the remainder launch alone (154 tiles of 128x128) for several K."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "da-sac_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from dasac_hip import ops, lib as L
from gemm_exp import timeit

B, H, W = 8, 97, 97
lib = L.load()
for cin, k in ((256, 1), (512, 1), (1024, 1), (2048, 1), (256, 3), (512, 3)):
    cout = 256
    spec = ops.ConvSpec(cin, cout, [(k, k, 1, k // 2)], 1)
    x = torch.randn(B, cin, H, W, device="cuda")
    w = [torch.randn(cout, cin, k, k, device="cuda") * 0.05]
    o = ops.gemm_order(spec, False)
    tab, pk = ops.conv_table(spec, H, W, False, x.device, o), ops.conv_pack(spec, w, False, order=o)
    y = torch.empty(B, cout, H, W, device="cuda")
    ws = L.workspace(lib.dasac_conv_gemm_workspace(), x.device, owner="conv_gemm")
    npix = B * H * W
    lead = 1024 // 2 * 128          # 512 pixel tiles x 2 M tiles = the 1024 tiles of the leading launch

    def run(pb, pc, sched):
        L.check(lib.dasac_conv_gemm(x.data_ptr(), pk.data_ptr(), tab.data_ptr(), y.data_ptr(), B, cin, H, W, H, W, 1, cout, spec.K, H, W, 1,
                                    0, 0, 0, 0, 0, 0, pb, pc, sched, ws.data_ptr(), ws.numel(), L.stream_ptr()), "gemm")
    t_lead = timeit(lambda: run(0, lead, 1), 20)
    t_sk = timeit(lambda: run(lead, 0, 2), 20)
    t_tail_plain = timeit(lambda: run(lead, 0, 1), 20)
    tiles_rem = 2 * ((npix - lead + 127) // 128)
    gf = lambda tiles: 2.0 * tiles * 128 * 128 * spec.K / 1e9
    print("K={:5d}  lead(1024 tiles) {:7.1f} us {:6.1f} TF | remainder {} tiles: stream-K {:7.1f} us {:6.1f} TF, tile-per-block {:7.1f} us {:6.1f} TF".format(
        spec.K, t_lead * 1e6, gf(1024) / t_lead / 1e3, tiles_rem, t_sk * 1e6, gf(tiles_rem) / t_sk / 1e3, t_tail_plain * 1e6, gf(tiles_rem) / t_tail_plain / 1e3), flush=True)
