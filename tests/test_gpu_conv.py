"""Implicit-GEMM convolution kernels (forward / data-gradient / weight-gradient) against ATen CPU
conv2d + autograd (the oracle's conv arithmetic), through the C ABI.  Tolerance: 1e-3 relative to
the tensor's max magnitude (BASELINE.json north_star), in practice ~1e-6."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4

CASES = [
    # name, cin, cout, branches[(kh,kw,dil,pad)], stride, (N,H,W)
    ("1x1", 64, 256, [(1, 1, 1, 0)], 1, (2, 25, 33)),
    ("1x1_s2", 256, 128, [(1, 1, 1, 0)], 2, (2, 25, 33)),
    # the four-pixels-per-lane weight-gradient loader (conv_wgrad<..., QUAD>: 1x1, stride 1, Cin % 128 == 0, Cout % 128 == 0):
    # 9x13 = 117 pixels per image (quads straddle images), 351 pixels in all (a ragged last step); then two k tiles x two m tiles
    ("1x1_quad_ragged", 128, 128, [(1, 1, 1, 0)], 1, (3, 9, 13)),
    ("1x1_quad", 256, 256, [(1, 1, 1, 0)], 1, (2, 25, 33)),
    ("3x3_d2", 32, 64, [(3, 3, 2, 2)], 1, (2, 19, 23)),
    ("3x3_d4_big", 128, 128, [(3, 3, 4, 4)], 1, (1, 33, 31)),
    # one tap per k tile with 128-row M tiles (the layer3 / layer4 weight-gradient path): a partial last M tile (200 of 256 rows)
    # over 117-pixel images, and a strided 1x1 (dZ grid != input grid)
    ("3x3_d2_fast_ragged", 128, 200, [(3, 3, 2, 2)], 1, (3, 9, 13)),
    ("1x1_s2_fast", 128, 256, [(1, 1, 1, 0)], 2, (2, 25, 33)),
    # round 6: the quad loader with per-pixel tap validity (conv_wgrad<..., QUAD, QTAP>: "same"-padded k x k, stride 1, Cin % 128 == 0,
    # Cout % 128 == 0, W >= 4): images smaller than a step with quads that straddle them, the narrowest maps (every quad wraps a
    # row; dilation beyond the map: most taps fall outside), a 5x5, and a launch with several pixel splits and two m / k tiles per tap
    ("3x3_qtap_images", 128, 256, [(3, 3, 2, 2)], 1, (3, 9, 13)),
    ("3x3_qtap_w4", 128, 128, [(3, 3, 1, 1)], 1, (2, 6, 4)),
    ("3x3_qtap_w5_d3", 256, 128, [(3, 3, 3, 3)], 1, (2, 7, 5)),
    ("5x5_qtap", 128, 128, [(5, 5, 1, 2)], 1, (1, 12, 10)),
    ("3x3_qtap_splits", 256, 256, [(3, 3, 2, 2)], 1, (2, 40, 37)),
    ("3x3_d1_c19", 48, 19, [(3, 3, 1, 1)], 1, (3, 9, 13)),
    ("7x7_s2_stem", 3, 64, [(7, 7, 1, 3)], 2, (2, 65, 49)),
    ("aspp4", 96, 19, [(3, 3, 6, 6), (3, 3, 12, 12), (3, 3, 18, 18), (3, 3, 24, 24)], 1, (2, 17, 21)),
    ("7x7_fcn_head", 16, 160, [(7, 7, 1, 3)], 1, (1, 8, 12)),
    ("7x7_fcn_head_cfg5", 512, 4096, [(7, 7, 1, 3)], 1, (1, 16, 32)),     # fcn.py:49 at its cfg-5 size (512x1024 crop / 32)
]


def _make(case, seed=0):
    name, cin, cout, br, stride, (N, H, W) = case
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, cin, H, W, generator=g)
    ws = [torch.randn(cout, cin, b[0], b[1], generator=g) / (cin * b[0] * b[1]) ** 0.5 for b in br]
    return x, ws


def _ref_forward(x, ws, br, stride):
    out = None
    for w, (kh, kw, d, p) in zip(ws, br):
        o = F.conv2d(x, w, None, stride, p, d)
        out = o if out is None else out + o
    return out


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_forward_dgrad_wgrad(case):
    from dasac_hip import ops
    name, cin, cout, br, stride, _ = case
    spec = ops.ConvSpec(cin, cout, br, stride)
    x, ws = _make(case)
    xr = x.clone().requires_grad_(True)
    wr = [w.clone().requires_grad_(True) for w in ws]
    yr = _ref_forward(xr, wr, br, stride)
    g = torch.Generator().manual_seed(1)
    dz = torch.randn(yr.shape, generator=g)
    yr.backward(dz)

    xd, wd, dzd = x.cuda(), [w.cuda() for w in ws], dz.cuda()
    y = ops.conv_forward(spec, xd, wd)
    assert y.shape == yr.shape
    assert rel_err(y, yr) < TOL
    gw = ops.conv_wgrad(spec, dzd, xd, wd)
    for a, b in zip(gw, wr):
        assert rel_err(a, b.grad) < TOL
    if stride == 1 or spec.taps == 1:
        dx = ops.conv_dgrad(spec, dzd, wd, x.shape[2:])
        assert rel_err(dx, xr.grad) < TOL


def test_epilogue_bn_residual_relu_and_mask():
    """y = relu(scale*conv + shift + res): the fused 'ABN' epilogue, and the masked dgrad epilogue."""
    from dasac_hip import ops
    case = ("e", 40, 72, [(3, 3, 2, 2)], 1, (2, 15, 18))
    spec = ops.ConvSpec(40, 72, case[3], 1)
    x, ws = _make(case, 3)
    g = torch.Generator().manual_seed(4)
    scale, shift = torch.rand(72, generator=g) + 0.5, torch.randn(72, generator=g)
    res = torch.randn(2, 72, 15, 18, generator=g)
    ref = F.relu(F.conv2d(x, ws[0], None, 1, 2, 2) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res)
    y = ops.conv_forward(spec, x.cuda(), [ws[0].cuda()], scale.cuda(), shift.cuda(), res.cuda(), relu=True)
    assert rel_err(y, ref) < TOL
    # dgrad with per-channel scale folded into the packed weights, accumulate + mask
    dz = torch.randn(ref.shape, generator=g)
    acc = torch.randn(x.shape, generator=g)
    msk = torch.randn(x.shape, generator=g)
    dref = F.conv_transpose2d(dz * scale.view(1, -1, 1, 1), ws[0], None, 1, 2, 0, 1, 2) + acc
    dref = torch.where(msk > 0, dref, torch.zeros_like(dref))
    dx = ops.conv_dgrad(spec, dz.cuda(), [ws[0].cuda()], x.shape[2:], scale=scale.cuda(), res=acc.cuda(), mask=msk.cuda())
    assert rel_err(dx, dref) < TOL
    # wgrad with scale and the gamma-gradient dot term
    dot = torch.full((ops.dot_rows(spec), 72), float("nan"), device="cuda")       # partial rows: every element is written
    (gw,) = ops.conv_wgrad(spec, dz.cuda(), x.cuda(), [ws[0].cuda()], scale=scale.cuda(), dot=dot)
    xr, wr = x.clone(), ws[0].clone().requires_grad_(True)
    z = F.conv2d(xr, wr, None, 1, 2, 2)
    (z * scale.view(1, -1, 1, 1) * dz).sum().backward()
    assert rel_err(gw, wr.grad) < TOL
    assert rel_err(dot.sum(0), (z.detach() * dz).sum((0, 2, 3))) < TOL


def test_strided_1x1_dgrad_accumulates_off_lattice():
    from dasac_hip import ops
    spec = ops.ConvSpec(24, 40, [(1, 1, 1, 0)], 2)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 24, 13, 17, generator=g, requires_grad=True)
    w = torch.randn(40, 24, 1, 1, generator=g)
    y = F.conv2d(x, w, None, 2)
    dz = torch.randn(y.shape, generator=g)
    y.backward(dz)
    acc = torch.randn(x.shape, generator=g)
    dx = ops.conv_dgrad(spec, dz.cuda(), [w.cuda()], (13, 17), res=acc.cuda())
    assert rel_err(dx, x.grad + acc) < TOL


def test_resnet_shapes_full_size_linearity():
    """cfg-3 layer3 shape [8,256,97,97] 3x3 d2: too big for the CPU oracle in seconds, so check
    linearity conv(a*x1 + x2) == a*conv(x1) + conv(x2) and a sampled direct evaluation."""
    from dasac_hip import ops
    spec = ops.ConvSpec(256, 256, [(3, 3, 2, 2)], 1)
    g = torch.Generator(device="cuda").manual_seed(7)
    x1 = torch.randn(8, 256, 97, 97, device="cuda", generator=g)
    x2 = torch.randn(8, 256, 97, 97, device="cuda", generator=g)
    w = torch.randn(256, 256, 3, 3, device="cuda", generator=g) / 48
    table = ops.conv_table(spec, 97, 97, False, x1.device)
    packed = ops.conv_pack(spec, [w], False)
    y1 = ops.conv_forward(spec, x1, [w], table=table, packed=packed)
    y2 = ops.conv_forward(spec, x2, [w], table=table, packed=packed)
    y3 = ops.conv_forward(spec, 0.5 * x1 + x2, [w], table=table, packed=packed)
    assert rel_err(y3, 0.5 * y1 + y2) < 1e-5
    # direct evaluation of 64 sampled outputs on the CPU in float64
    xs, wc, yc = x1.cpu().double(), w.cpu().double(), y1.cpu()
    xp = F.pad(xs, (2, 2, 2, 2))
    for i in range(64):
        n, co, oh, ow = i % 8, (37 * i) % 256, (11 * i) % 97, (29 * i + 3) % 97
        patch = xp[n, :, oh:oh + 5:2, ow:ow + 5:2]
        assert abs(float((patch * wc[co]).sum()) - float(yc[n, co, oh, ow])) < 1e-4


SMALL_BATCH = [
    # name, cin, cout, branch, (N, H, W): fewer 128x128 tiles than persistent workers -> several ranges contribute to a tile
    ("l3_3x3_b2", 256, 256, (3, 3, 2, 2), (2, 97, 97)),          # 296 tiles x 144 K-steps over 768 workers
    ("l3_3x3_b1", 256, 256, (3, 3, 2, 2), (1, 97, 97)),          # 148 tiles: ~5 ranges per tile
    ("l3_1x1_b1", 1024, 256, (1, 1, 1, 0), (1, 97, 97)),         # 64 K-steps
    ("l3_1x1_short_k_b1", 256, 1024, (1, 1, 1, 0), (1, 97, 97)),  # 16 K-steps, 592 tiles: stays tile-per-block
    ("l4_3x3_b1_odd", 512, 512, (3, 3, 4, 4), (1, 61, 53)),      # ragged last pixel tile
]


@pytest.mark.parametrize("case", SMALL_BATCH, ids=[c[0] for c in SMALL_BATCH])
def test_small_batch_stream_k_with_several_contributors(case):
    """Inference / cfg-2 sized launches: the persistent schedule cuts a tile into more than two ranges; the worker
    holding its first K-steps sums the deposits of all the following ones."""
    from dasac_hip import ops
    name, cin, cout, br, (N, H, W) = case[:5]
    sched = case[5] if len(case) > 5 else None
    spec = ops.ConvSpec(cin, cout, [br], 1)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, br[0], br[1], generator=g) / (cin * br[0] * br[1]) ** 0.5
    shift = torch.randn(cout, generator=g)
    res = torch.randn(N, cout, H, W, generator=g)
    dz = torch.randn(N, cout, H, W, generator=g)
    yr = torch.relu(F.conv2d(x.double(), w.double(), None, 1, br[3], br[2]) + shift.double().view(1, -1, 1, 1) + res.double())
    dxr = F.conv_transpose2d(dz.double(), w.double(), None, 1, br[3], 0, 1, br[2])
    y = ops.conv_forward(spec, x.cuda(), [w.cuda()], shift=shift.cuda(), res=res.cuda(), relu=True)
    assert rel_err(y.double().cpu(), yr) < TOL
    dx = ops.conv_dgrad(spec, dz.cuda(), [w.cuda()], (H, W))
    assert rel_err(dx.double().cpu(), dxr) < TOL
    ops.set_precision("bf16x3")
    try:
        y3 = ops.conv_forward(spec, x.cuda(), [w.cuda()], shift=shift.cuda(), res=res.cuda(), relu=True)
    finally:
        ops.set_precision("fp32")
    assert rel_err(y3.double().cpu(), yr) < TOL


def test_whole_network_refresh_equals_the_per_layer_folds_and_packs():
    """Engine.refresh (dasac_bn_fold_multi + dasac_conv_pack_multi: every frozen-BN fold and every packed weight operand of
    ResNet-101 in two launches) against the per-layer kernels it replaces -- bit for bit, forward and data-gradient layouts,
    both K orders, K / M padding zeroed -- and that an optimiser-style in-place update makes the next forward refresh again."""
    from types import SimpleNamespace as NS
    import torch.nn as nn
    import models
    from dasac_hip import ops
    from oracle import nets_ref as N
    from oracle.step_ref import DEFAULT_CFG
    cfg = NS(**dict(DEFAULT_CFG, INIT_MODEL="", OPT_NESTEROV=False))
    net = models.get_model(cfg, 0, num_classes=19, criterion=nn.CrossEntropyLoss(ignore_index=255, reduction="none"))
    net.backbone.load_state_dict(N.resnet101_state(seed=2, randomize_bn=True, he_init=True), strict=True)
    net.cuda().train()
    bb = net.backbone
    x = torch.randn(1, 3, 33, 49, device="cuda")
    calls = []
    real_pack, real_fold = ops.conv_pack, ops.bn_fold
    ops.conv_pack = lambda *a, **k: (calls.append("pack"), real_pack(*a, **k))[1]
    ops.bn_fold = lambda *a, **k: (calls.append("fold"), real_fold(*a, **k))[1]
    try:
        out = bb._logits(x)
        out.sum().backward()
        eng = bb._engine
        assert not calls, calls[:4]                      # nothing went through the per-layer path
        n = 0
        for op in eng.plan.ops:
            if op.kind != "conv" or op.expanded is not None:
                continue
            scale, shift, invstd = eng._folds[id(op)][1]
            ws, wh, wi = real_fold(op.bn.weight.detach(), op.bn.bias.detach(), op.bn.running_mean, op.bn.running_var, op.bn.eps, None)
            assert torch.equal(scale, ws) and torch.equal(shift, wh) and torch.equal(invstd, wi)
            for tr in (False, True):
                got = eng._packs[(id(op), tr)][1]
                want = real_pack(op.spec, [op.convs[0].weight.detach()], tr, scale, order=ops.gemm_order(op.spec, tr))
                assert got.shape == want.shape and torch.equal(got, want), (op.spec.cout, op.spec.cin, tr)
                n += 1
        assert n == 2 * 104
        # in-place parameter update (what FusedSGD does): everything is stale again and is rebuilt by the next forward
        with torch.no_grad():
            for p in bb.parameters():
                p.mul_(1.01)
        before = eng._packs[(id(eng.plan.ops[0]), False)][1].clone()
        bb._logits(x)
        assert not calls and not torch.equal(before, eng._packs[(id(eng.plan.ops[0]), False)][1])
        # one stale layer only: the per-layer path handles it
        with torch.no_grad():
            eng.plan.ops[5].convs[0].weight.mul_(0.5)
        bb._logits(x)
        assert calls == ["pack"]
    finally:
        ops.conv_pack, ops.bn_fold = real_pack, real_fold


@pytest.mark.parametrize("cin,cout,k,dil,shape", [
    (64, 256, 1, 1, (2, 25, 33)),          # 1650 pixels: a ragged last word, a ragged last tile
    (32, 200, 3, 2, (3, 19, 23)),          # M = 200 -> padded to 256: whole 8-row groups beyond M
    (48, 130, 3, 1, (1, 31, 17)),          # M % 8 != 0: per-lane row test
    (256, 256, 3, 2, (8, 97, 97)),         # the layer3 3x3 at its cfg-3 size: leading rounds + stream-K remainder launches
])
def test_relu_bit_masks_equal_the_fp32_pattern(cin, cout, k, dil, shape):
    """ReLU patterns as one bit per element (dasac_conv_gemm relu_bits_out / mask_bits): the forward epilogue's bits are exactly
    (out > 0) in the documented layout, and a data-gradient GEMM masked by the bits equals the one masked by the fp32 activation
    bit for bit -- the round-3 replacement for reading a whole activation back just to test its sign."""
    from dasac_hip import ops
    N_, H, W = shape
    g = torch.Generator().manual_seed(cin + cout)
    spec = ops.ConvSpec(cin, cout, [(k, k, dil, dil * (k // 2))], 1)
    assert ops.bits_ok(cout, cin) and ops.bits_ok(cin, cout) == (cin > 64 and cout % 16 == 0)
    x = torch.randn(N_, cin, H, W, generator=g).cuda()
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).cuda()
    shift = torch.randn(cout, generator=g).cuda() * 0.1
    order = ops.gemm_order(spec, False)
    table, packed = ops.conv_table(spec, H, W, False, x.device, order), ops.conv_pack(spec, [w], False, None, order=order)
    y0 = torch.empty(N_, cout, H, W, device="cuda")
    ops.conv_gemm(x, packed, table, y0, (H, W), 1, cout, spec.K, 1, shift, None, None, True)
    y1 = torch.empty_like(y0)
    bits = ops.ReluBits(N_, cout, H, W, x.device)
    bits.words.fill_(0x55555555)
    ops.conv_gemm(x, packed, table, y1, (H, W), 1, cout, spec.K, 1, shift, None, None, True, bits_out=bits)
    assert torch.equal(y0, y1)
    npix = N_ * H * W
    w32 = (npix + 31) // 32
    pos = (y1 > 0).permute(1, 0, 2, 3).reshape(cout, npix)                   # [m][flattened (n, oh, ow)]
    pad = torch.zeros(cout, w32 * 32 - npix, dtype=torch.bool, device="cuda")
    want = torch.cat([pos, pad], 1).view(cout, w32, 32).to(torch.int64)
    want = (want << torch.arange(32, device="cuda")).sum(-1)
    got = bits.words.view(cout, w32).to(torch.int64) & 0xFFFFFFFF
    valid = torch.full((w32,), 0xFFFFFFFF, dtype=torch.int64, device="cuda")
    if npix % 32:
        valid[-1] = (1 << (npix % 32)) - 1                                   # bits past the last pixel are unspecified
    assert torch.equal(got & valid, want & valid)
    assert 0.2 < float(pos.float().mean()) < 0.8
    # consumer: a 1x1 data-gradient GEMM into this tensor's shape, masked by bits vs by the activation itself
    if ops.bits_ok(cout, 64):
        spec2 = ops.ConvSpec(cout, 64, [(1, 1, 1, 0)], 1)
        w2 = (torch.randn(64, cout, 1, 1, generator=g) / cout ** 0.5).cuda()
        dz = torch.randn(N_, 64, H, W, generator=g).cuda()
        res = torch.randn(N_, cout, H, W, generator=g).cuda()
        a = ops.conv_dgrad(spec2, dz, [w2], (H, W), res=res, mask=y1)
        b = ops.conv_dgrad(spec2, dz, [w2], (H, W), res=res, mask=bits)
        assert torch.equal(a, b)
        assert float((a == 0).float().mean()) > 0.2


@pytest.mark.parametrize("cin,cout,k,dil,shape,want_split", [
    (256, 256, 3, 2, (8, 97, 97), 6),        # layer3 3x3, one student pass: 1178 tiles = 1024 + 154 x 6 K-ranges of 24 steps
    (1024, 256, 1, 1, (8, 97, 97), 6),       # conv1 of layer3: 64 K-steps
    (512, 512, 3, 4, (8, 97, 97), 3),        # layer4 3x3: 2356 tiles = 2048 + 308 x 3
    (2048, 512, 1, 1, (16, 97, 97), 3),      # 4708 tiles = 4096 + 612 x 3: the pieces take TWO rounds of resident blocks
])
def test_split_k_tail_in_the_tile_launch(cin, cout, k, dil, shape, want_split):
    """Round 6: a long-K convolution whose tile count leaves a ragged last round runs as ONE launch -- whole rounds one block per
    tile, the remaining tiles cut into K-ranges (conv_gemm<SK = 2>: later ranges deposit, the first range's block adds and runs the
    epilogue).  Against the plain one-block-per-tile launch (summation order of the cut tiles differs: 1e-6), for every epilogue
    the network uses; two runs give the same bits (fixed ranges, fixed order of the deposits); the recorded ReLU bits follow
    the stored values; the hand-off flags are left clean."""
    from dasac_hip import ops
    lib = ops.L.load()
    N_, H, W = shape
    assert lib.dasac_conv_gemm_tail_split(N_, H, W, cout, cin * k * k) == want_split
    g = torch.Generator().manual_seed(cin + cout + k)
    spec = ops.ConvSpec(cin, cout, [(k, k, dil, dil * (k // 2))], 1)
    x = torch.randn(N_, cin, H, W, generator=g).cuda()
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).cuda()
    shift = (torch.randn(cout, generator=g) * 0.1).cuda()
    res = torch.randn(N_, cout, H, W, generator=g).cuda()
    order = ops.gemm_order(spec, False)
    table, packed = ops.conv_table(spec, H, W, False, x.device, order), ops.conv_pack(spec, [w], False, None, order=order)
    mask = ops.ReluBits(N_, cout, H, W, x.device)
    mask.words.random_(-2 ** 31, 2 ** 31 - 1)

    def run(schedule, shift_, res_, relu, want_bits=False, mask_=None):
        out = torch.full((N_, cout, H, W), float("nan"), device="cuda")
        bits = None
        if want_bits:
            bits = ops.ReluBits(N_, cout, H, W, x.device)
            bits.words.fill_(0x55555555)
        ops.conv_gemm(x, packed, table, out, (H, W), 1, cout, spec.K, 1, shift_, res_, mask_, relu, bits_out=bits, schedule=schedule)
        return out, bits

    for shift_, res_, relu, want_bits, mask_ in ((None, None, False, False, None), (shift, res, True, True, None), (None, res, False, False, mask)):
        a, ba = run(None, shift_, res_, relu, want_bits, mask_)
        b, bb = run(1, shift_, res_, relu, want_bits, mask_)
        a2, _ = run(None, shift_, res_, relu, want_bits, mask_)
        assert not torch.isnan(a).any()
        assert rel_err(a, b) < 1e-5                           # fp32 sums of up to 4608 products in another order
        assert torch.equal(a, a2)
        assert float((a != b).float().mean()) < 0.5          # the leading rounds run the very same code: equal bits there
        if want_bits:
            pos = (a > 0).permute(1, 0, 2, 3).reshape(cout, -1)
            npix = N_ * H * W
            idx = torch.arange(npix, device="cuda")
            got = (ba.words.view(cout, -1)[:, idx >> 5].to(torch.int64) >> (idx & 31)) & 1
            assert torch.equal(got.bool(), pos)
    ws = ops.L.workspace(lib.dasac_conv_gemm_workspace(), x.device, owner="conv_gemm")
    torch.cuda.synchronize()
    flags = ws[ws.numel() - 4 * 2049:].view(torch.int32)
    assert int(flags.abs().sum()) == 0                       # self-cleaning flags: every raised flag was taken down by its owner


STATS_CASES = [
    # name, cin, cout, branch, (N,H,W): one block per tile / the persistent stream-K schedule (long K, few tiles: the worker that
    # holds a tile's first K-steps runs the epilogue and must own the tile's statistics slot) / a ragged last M tile and pixel tile
    ("tiles_1x1", 64, 256, (1, 1, 1, 0), (2, 25, 33)),
    ("streamk_3x3", 128, 256, (3, 3, 2, 2), (2, 33, 41)),
    ("ragged_m200", 128, 200, (3, 3, 2, 2), (3, 9, 13)),
    # forced stream-K with MORE tiles than workers (1178 tiles x 4 K-steps over 768 workers: a worker finishes one tile's epilogue and
    # starts the next tile): the statistics scratch aliases the operand LDS of the next tile -- the loop-end barrier orders them
    ("streamk_consecutive_tiles", 64, 256, (1, 1, 1, 0), (8, 97, 97), 2),
]


@pytest.mark.parametrize("case", STATS_CASES, ids=[c[0] for c in STATS_CASES])
def test_gemm_epilogue_channel_statistics(case):
    """dasac_conv_gemm_stats: the raw conv (+ bias) in front of a batch-statistics BatchNorm (deeplabv2.py:15 in baseline mode)
    also leaves per-tile channel sums / sums of squares.  Output bit-equal to the plain kernel's; statistics against float64
    sums of that output; two runs bit-identical (every slot has one writer); and bn_train_forward on them equals the
    stand-alone statistics pass."""
    import torch.nn as nn
    from dasac_hip import ops
    name, cin, cout, br, (N, H, W) = case[:5]
    sched = case[5] if len(case) > 5 else None
    spec = ops.ConvSpec(cin, cout, [br], 1)
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(N, cin, H, W, generator=g) + 0.3).cuda()
    w = (torch.randn(cout, cin, br[0], br[1], generator=g) / (cin * br[0] * br[1]) ** 0.5).cuda()
    bias = torch.randn(cout, generator=g).cuda()
    assert ops.stats_ok(cout, cin)
    order = ops.gemm_order(spec, False)
    table, packed = ops.conv_table(spec, H, W, False, x.device, order), ops.conv_pack(spec, [w], False, None, order=order)
    OH, OW = spec.out_hw(H, W)
    plain = torch.empty((N, cout, OH, OW), device="cuda")
    ops.conv_gemm(x, packed, table, plain, (OH, OW), 1, cout, spec.K, 1, bias, schedule=sched)
    runs = []
    for _ in range(2):
        out = torch.empty_like(plain)
        ts = ops.tile_stats_buffer(N, cout, OH, OW, x.device).fill_(float("nan"))
        ops.conv_gemm(x, packed, table, out, (OH, OW), 1, cout, spec.K, 1, bias, stats=ts, schedule=sched)
        runs.append((out, ts))
    (out, ts), (out2, ts2) = runs
    assert torch.equal(out, plain) and torch.equal(out2, plain) and torch.equal(ts, ts2)
    assert not torch.isnan(ts).any() and float(ts[:, :, cout:].abs().max() if ts.shape[2] > cout else 0.0) == 0.0
    s, q = ts[:, 0, :cout].double().sum(0), ts[:, 1, :cout].double().sum(0)
    od = out.double()
    assert float((s - od.sum((0, 2, 3))).abs().max()) <= 1e-6 * float(od.abs().sum((0, 2, 3)).max())
    assert float(((q - (od * od).sum((0, 2, 3))) / (od * od).sum((0, 2, 3))).abs().max()) <= 1e-6
    bn_a, bn_b = nn.BatchNorm2d(cout).cuda(), nn.BatchNorm2d(cout).cuda()
    res = torch.randn(out.shape, generator=g).cuda()
    ya, (mean_a, inv_a, _, _) = ops.bn_train_forward(out, bn_a, res, True, tile_stats=ts)     # finalize + apply in one launch
    yb, (mean_b, inv_b, _, _) = ops.bn_train_forward(out, bn_b, res, True)                    # stats pass, finalize, apply
    assert rel_err(mean_a, mean_b) < 1e-6 and rel_err(inv_a, inv_b) < 1e-6 and rel_err(ya, yb) < 1e-5
    assert rel_err(bn_a.running_var, bn_b.running_var) < 1e-6 and rel_err(bn_a.running_mean, bn_b.running_mean) < 1e-6
    assert int(bn_a.num_batches_tracked) == 1 and int(bn_b.num_batches_tracked) == 1
    ref = torch.relu(torch.nn.functional.batch_norm(out.cpu(), None, None, None, None, True, 0.1, 1e-5) + res.cpu())
    assert rel_err(ya, ref) < 1e-5
    # and the backward pair: stage-1 partials + one dz kernel that also writes d gamma / d beta, against autograd on the CPU
    zc = out.cpu().clone().requires_grad_(True)
    gam = (torch.rand(cout, generator=g) + 0.5)
    dyc = torch.randn(out.shape, generator=g)
    gamr = gam.clone().requires_grad_(True)
    yc = torch.nn.functional.batch_norm(zc, None, None, gamr, torch.zeros(cout, requires_grad=True), True, 0.1, 1e-5)
    yc.backward(dyc)
    dz, dg, db = ops.bn_train_backward(dyc.cuda(), out, (mean_a, inv_a, float(N * OH * OW), None), gam.cuda())
    assert rel_err(dz, zc.grad) < 1e-4 and rel_err(dg, gamr.grad) < 1e-4 and rel_err(db, dyc.sum((0, 2, 3))) < 1e-5


def test_reserved_cus_shrink_the_persistent_grid_without_changing_results():
    """dasac_set_reserved_cus(n): the stream-K grid (and the streaming kernels' grid cap) is sized to 256 - n CUs so that RCCL's
    kernels, which the overlapped data-parallel wrapper runs beside the backward GEMMs, do not share CUs with workers that all
    carry the same matrix work.  The partition of the (tile, K-step) space changes with the worker count, the result only in
    the summation order of tiles cut by a range boundary."""
    from dasac_hip import ops
    lib = ops.L.load()
    g = torch.Generator().manual_seed(5)
    spec = ops.ConvSpec(256, 256, [(3, 3, 2, 2)], 1)
    x = torch.randn(8, 256, 97, 97, generator=g).cuda()
    w = (torch.randn(256, 256, 3, 3, generator=g) / 48.0).cuda()
    order = ops.gemm_order(spec, False)
    table, packed = ops.conv_table(spec, 97, 97, False, x.device, order), ops.conv_pack(spec, [w], False, None, order=order)
    prev = lib.dasac_set_reserved_cus(0)
    try:
        outs = {}
        for n in (0, 8, 16, 13):
            lib.dasac_set_reserved_cus(n)
            assert lib.dasac_reserved_cus() == (n + 7) // 8 * 8
            y = torch.full((8, 256, 97, 97), float("nan"), device="cuda")
            ops.conv_gemm(x, packed, table, y, (97, 97), 1, 256, spec.K, 1, None, None, None, False, schedule=2)
            outs[n] = y
        ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=2, dilation=2).float()
        for n, y in outs.items():
            assert rel_err(y, ref) < 5e-6, n
            assert float((y - outs[0]).abs().max()) <= 5e-6 * float(ref.abs().max()), n      # (K = 2304: another cut, another summation order)
        assert torch.equal(outs[16], outs[13])                       # 13 rounds up to 16: the same grid, the same bits
    finally:
        lib.dasac_set_reserved_cus(prev)


def test_tensors_between_2_and_4_gib_are_addressed_correctly():
    """Round 5 lifted the conv kernels' addressing window from 2 GiB to 4 GiB (unsigned 32-bit byte offsets; FCN-8s at 16 crops
    of 512x1024 -- cfg-5's fused student pass -- peaks at exactly 2 GiB per activation).  A 3x3 convolution 64 -> 64 over
    17 x 64 x 512 x 1024 (2.125 GiB in, 2.125 GiB out): forward and data gradient equal, bit for bit, the same call on the two
    halves of the batch (tiles never straddle an image here: 524288 pixels per image), the weight gradient their sum."""
    from dasac_hip import ops
    N_, C, H, W = 17, 64, 512, 1024
    g = torch.Generator(device="cuda").manual_seed(3)
    spec = ops.ConvSpec(C, C, [(3, 3, 1, 1)], 1)
    x = torch.randn(N_, C, H, W, device="cuda", generator=g)
    assert x.numel() * 4 > (1 << 31)
    w = torch.randn(C, C, 3, 3, device="cuda", generator=g) / 24.0
    shift = torch.randn(C, device="cuda", generator=g) * 0.1
    order = ops.gemm_order(spec, False)
    table, packed = ops.conv_table(spec, H, W, False, x.device, order), ops.conv_pack(spec, [w], False, None, order=order)

    def fwd(xx):
        y = torch.empty_like(xx)
        ops.conv_gemm(xx, packed, table, y, (H, W), 1, C, spec.K, 1, shift, None, None, True, schedule=1)
        return y

    y = fwd(x)
    lo, hi = slice(0, 9), slice(9, N_)
    assert torch.equal(y[lo], fwd(x[lo].contiguous())) and torch.equal(y[hi], fwd(x[hi].contiguous()))
    assert float(y[-1].abs().sum()) > 0 and bool(torch.isfinite(y[-1]).all())
    # spot check against ATen on the LAST image (the far end of the window)
    want = torch.relu(torch.nn.functional.conv2d(x[-1:].double(), w.double(), padding=1).float() + shift.view(1, -1, 1, 1))
    assert rel_err(y[-1:], want) < 2e-6
    # data gradient (transposed table / packing), masked by the forward's output, accumulated into a residual
    dz = torch.randn(N_, C, H, W, device="cuda", generator=g)
    res = torch.randn(1, C, H, W, device="cuda", generator=g).expand(N_, C, H, W).contiguous()
    dx = ops.conv_dgrad(spec, dz, [w], (H, W), res=res.clone(), mask=y)
    dx_lo = ops.conv_dgrad(spec, dz[lo].contiguous(), [w], (H, W), res=res[lo].clone(), mask=y[lo].contiguous())
    dx_hi = ops.conv_dgrad(spec, dz[hi].contiguous(), [w], (H, W), res=res[hi].clone(), mask=y[hi].contiguous())
    assert rel_err(dx[lo], dx_lo) < 1e-6 and rel_err(dx[hi], dx_hi) < 1e-6
    # weight gradient: the sum over the two halves
    dw = ops.conv_wgrad(spec, dz, x, [w])[0]
    dw2 = ops.conv_wgrad(spec, dz[lo].contiguous(), x[lo].contiguous(), [w])[0] + ops.conv_wgrad(spec, dz[hi].contiguous(), x[hi].contiguous(), [w])[0]
    assert rel_err(dw, dw2) < 1e-5
