"""SAC head kernels vs the CPU oracle and the reference-captured goldens, through the C ABI.
Float tolerance: 1e-3 relative to the tensor max (north_star); measured errors are ~1e-6."""
import pytest
import torch

from oracle import head_ref as H
from conftest import rel_err

pytestmark = pytest.mark.gpu
T = torch.from_numpy
TOL = 1e-5


def test_upsample_softmax_golden_g3(golden):
    from dasac_hip import ops
    g = golden("g3_bilinear")
    up, _, _ = ops.upsample_softmax(T(g["x"]).cuda(), (65, 97))
    assert rel_err(up, g["y"]) < TOL
    up, _, _ = ops.upsample_softmax(T(g["row"]).cuda(), (769, 9))
    assert rel_err(up, g["yrow"]) < TOL
    up, _, _ = ops.upsample_softmax(T(g["x2"]).cuda(), (8, 12))
    assert rel_err(up, g["y2"]) < TOL


def test_upsample_softmax_probs_and_sums():
    from dasac_hip import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 19, 9, 13, generator=g) * 3
    ign = torch.rand(3, 65, 97, generator=g) < 0.2
    up, probs, sums = ops.upsample_softmax(x.cuda(), (65, 97), ign.cuda(), want_probs=True, want_sums=True)
    ru = H.upsample_bilinear_ac(x, 65, 97)
    rp = torch.softmax(ru, 1)
    assert rel_err(up, ru) < TOL
    assert rel_err(sums, rp.double().sum((0, 2, 3))) < TOL
    assert rel_err(probs, rp * (~ign)[:, None]) < TOL
    assert float(probs.cpu()[ign[:, None].expand_as(rp)].abs().max()) == 0.0


@pytest.mark.parametrize("shape", [(2, 19, 9, 13, 65, 97), (1, 3, 4, 6, 8, 12), (2, 19, 5, 5, 33, 33), (1, 2, 1, 7, 1, 49)])
def test_upsample_backward_is_the_transpose(shape):
    from dasac_hip import ops
    B, C, h, w, Hh, W = shape
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, C, h, w, generator=g, requires_grad=True)
    gu = torch.randn(B, C, Hh, W, generator=g)
    H.upsample_bilinear_ac(x, Hh, W).backward(gu)
    gs = torch.tensor([0.37])
    d = ops.upsample_bwd(gu.cuda(), (h, w), gs.cuda())
    assert rel_err(d, 0.37 * x.grad) < TOL


def test_ce_losses_golden_g6(golden):
    from dasac_hip import ops
    g = golden("g6_losses")
    x, y, conf, chi = T(g["logits"]).cuda(), T(g["y"]).cuda(), T(g["conf"]).cuda(), T(g["chi"])
    fw = H.focal_weight(chi, 3).cuda()
    loss, dl, pc = ops.ce_loss(x, y, fw, conf, want_grad=True, want_per_class=True)
    assert rel_err(loss, g["conf_loss"]) < TOL and rel_err(dl, g["conf_grad"]) < TOL
    assert rel_err(pc, g["conf_per_class"]) < TOL
    loss, dl, pc = ops.ce_loss(x, y, fw, None, want_grad=True, want_per_class=True)
    assert rel_err(loss, g["plain_loss"]) < TOL and rel_err(dl, g["plain_grad"]) < TOL
    assert rel_err(pc, g["plain_per_class"]) < TOL
    loss, dl, _ = ops.ce_loss(x, y, None, None, want_grad=True)
    assert rel_err(loss, g["ce_loss"]) < TOL and rel_err(dl, g["ce_grad"]) < TOL


def test_ce_all_ignored_and_batch_of_one():
    from dasac_hip import ops
    x = torch.randn(1, 19, 5, 7)
    y = torch.full((1, 5, 7), 255, dtype=torch.int64)
    loss, dl, _ = ops.ce_loss(x.cuda(), y.cuda(), want_grad=True)
    assert float(loss) == 0.0 and float(dl.abs().max()) == 0.0


def test_warp_affine_golden_g4(golden):
    from dasac_hip import ops
    g = golden("g4_refine")
    out = ops.warp_affine(T(g["frames"]).cuda(), T(g["affine"]).cuda())
    assert rel_err(out, g["warp_frames"]) < TOL
    assert rel_err(out, g["avg_pool_frames_aligned"]) < TOL


def _refine_on_gpu(g, mode):
    from dasac_hip import ops
    frames, logits = T(g["frames"]).cuda(), T(g["logits"]).cuda()
    aff, inv, ign = T(g["affine"]).cuda(), T(g["affine_inv"]).cuda(), T(g["ignore"]).cuda()
    Tn = int(g["T"])
    chi = T(g["chi_in"]).clone().cuda()
    B, _, Hh, W = frames.shape
    _, probs, sums = ops.upsample_softmax(logits, (Hh, W), ign, want_up=False, want_probs=True, want_sums=True)
    ops.class_state(chi, sums, B, Hh * W, 1e-3, 0.99, True, 3.0)
    pooled, mask, aligned = ops.warp_pool(probs, aff, inv, Tn, mode)
    refined = ops.warp_back(pooled, mask, inv, Tn)
    return refined, chi, aligned


def test_refine_avg_pool_golden_g4(golden):
    g = golden("g4_refine")
    refined, chi, aligned = _refine_on_gpu(g, "avg_pool")
    assert rel_err(refined, g["avg_pool_refined"]) < 2e-5
    assert rel_err(chi, g["avg_pool_chi_out"]) < 1e-6
    assert rel_err(aligned, g["avg_pool_teacher_aligned"]) < TOL


@pytest.mark.parametrize("shape", [(4, 19, 21, 30), (2, 5, 22, 9), (2, 19, 1, 7), (6, 2, 33, 64), (2, 3, 9, 1), (2, 19, 1, 1)])
def test_warp_back_against_the_oracle(shape):
    """refined[b] = sample(pooled[b // T], theta_inv[b]) * sample(mask[b // T], theta_inv[b]) under random affines (odd and even
    heights, a single row, few classes, a single column -- grid_sample accepts any size)."""
    from dasac_hip import ops
    B, C, Hh, W = shape
    Tn = 2
    g = torch.Generator().manual_seed(sum(shape))
    pooled = torch.rand(B // Tn, C, Hh, W, generator=g)
    mask = (torch.rand(B // Tn, 1, Hh, W, generator=g) > 0.3).float()
    theta = torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]).repeat(B, 1, 1) + 0.25 * torch.randn(B, 2, 3, generator=g)
    got = ops.warp_back(pooled.cuda(), mask.cuda(), theta.cuda(), Tn)
    src = pooled.repeat_interleave(Tn, 0)
    ref = H.warp_affine(src, theta) * H.warp_affine(mask.repeat_interleave(Tn, 0), theta)
    assert got.shape == ref.shape and rel_err(got, ref) < 2e-5


def test_refine_minentropy_pool_golden_g4(golden):
    g = golden("g4_refine")
    refined, chi, _ = _refine_on_gpu(g, "minentropy_pool")
    ref = T(g["minentropy_pool_refined"])
    bad = ((refined.cpu() - ref).abs().amax(1) > 1e-4).float().mean()
    assert bad < 2e-3                       # argmin over views is a discontinuity
    assert rel_err(chi, g["minentropy_pool_chi_out"]) < 1e-6


def test_class_state_sequence_golden_g7(golden):
    from dasac_hip import ops
    g = golden("g7_state")
    chi = torch.zeros(19, device="cuda")
    for it in range(4):
        if it == 1:
            chi.fill_(1e-3)
        p = T(g["probs"][it]).cuda()
        sums = p.double().sum((0, 2, 3))
        disc, focal = ops.class_state(chi, sums, p.shape[0], p.shape[2] * p.shape[3], 1e-3, 0.99, True, 3.0)
        assert rel_err(chi, g["chi_seq"][it]) < 1e-6
        assert rel_err(disc, H.threshold_discount(chi.cpu(), 1e-3)) < 1e-6
        assert rel_err(focal, H.focal_weight(chi.cpu(), 3)) < 1e-6


def test_maxpool_and_relu_masked_backward():
    from dasac_hip import ops
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(3)
    # 3x3/2 ceil (the stem pool, deeplabv2.py:126) and 2x2/2 (VGG) take the four-outputs-per-thread kernel: widths that are / are
    # not multiples of 4 outputs, a window row and column hanging over the border (ceil mode), tiny planes; 3x3/1 and 5x5/3 the
    # one-output kernel; the backward of 3x3/2 pad 1 is the 2 x 4-pixels-per-thread kernel
    for (k, s, p, ceil, Hh, W) in ((3, 2, 1, True, 33, 41), (2, 2, 0, False, 16, 24), (3, 2, 1, True, 32, 32), (3, 2, 1, True, 7, 5),
                                   (3, 2, 1, True, 65, 130), (3, 2, 1, False, 10, 9), (3, 2, 1, True, 6, 4), (3, 2, 1, False, 2, 38), (2, 2, 0, False, 9, 11), (3, 1, 1, False, 12, 13), (5, 3, 2, True, 21, 23)):
        x = F.relu(torch.randn(2, 5, Hh, W, generator=g)).requires_grad_(True)
        y, idx = F.max_pool2d(x, k, s, p, ceil_mode=ceil, return_indices=True)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        yy, arg = ops.maxpool_fwd(x.detach().cuda(), k, s, p, ceil)
        assert torch.equal(yy.cpu(), y.detach()), (k, s, Hh, W)
        # the argmax byte: window position in bits 0-6 (first maximum wins, like ATen's indices), bit 7 = pooled value > 0
        code = (arg.cpu() & 0x7f).long()
        oh = torch.arange(y.shape[2]).view(1, 1, -1, 1)
        ow = torch.arange(y.shape[3]).view(1, 1, 1, -1)
        flat = (oh * s - p + code // k) * W + (ow * s - p + code % k)
        assert torch.equal(torch.gather(x.detach().flatten(2), 2, flat.flatten(2)), y.detach().flatten(2)), (k, s, Hh, W)
        assert torch.equal((arg.cpu() >> 7).bool(), y.detach() > 0)
        dx = ops.maxpool_bwd(dy.cuda(), yy, arg, (Hh, W), k, s, p, relu_mask=True)
        # oracle: maxpool backward followed by the ReLU backward of the producer (grad * (x>0))
        ref = x.grad * (x.detach() > 0)
        assert rel_err(dx, ref) < TOL, (k, s, Hh, W)
        dx0 = ops.maxpool_bwd(dy.cuda(), yy, arg, (Hh, W), k, s, p, relu_mask=False)
        assert rel_err(dx0, x.grad) < TOL, (k, s, Hh, W)


def test_bn_fold_channel_sums_ema():
    from dasac_hip import ops
    g = torch.Generator().manual_seed(4)
    gam, bet, mu, var = torch.rand(37, generator=g) + 0.5, torch.randn(37, generator=g), torch.randn(37, generator=g), torch.rand(37, generator=g) + 0.5
    sc, sh, inv = ops.bn_fold(gam.cuda(), bet.cuda(), mu.cuda(), var.cuda(), 1e-5)
    x = torch.randn(2, 37, 5, 6, generator=g)
    ref = torch.nn.functional.batch_norm(x, mu, var, gam, bet, False, 0.1, 1e-5)
    assert rel_err(x * sc.cpu().view(1, -1, 1, 1) + sh.cpu().view(1, -1, 1, 1), ref) < 1e-6
    assert rel_err(ops.channel_sums(x.cuda()), x.sum((0, 2, 3))) < 1e-6
    fast = [torch.randn(n, generator=g) for n in (5, 4097, 70000)]
    slow = [torch.randn(n, generator=g) for n in (5, 4097, 70000)]
    fd, sd = [t.cuda() for t in fast], [t.cuda() for t in slow]
    plan = ops.EmaPlan(fd, sd)
    for upd in (False, True, False):
        ref = H.momentum_update({"a.weight": slow[0], "b.bias": slow[1], "c.running_var": slow[2]},
                                {"a.weight": fast[0], "b.bias": fast[1], "c.running_var": fast[2]}, 0.99, upd)
        out = plan.run(0.99, upd)
        assert rel_err(out, ref) < 1e-6
        for a, b in zip(sd, slow):
            assert rel_err(a, b) < 1e-6


def test_full_size_head_properties():
    """cfg-3 size ([8,19,97,97] -> [8,19,769,769]): size-independent properties of the head kernels."""
    import math
    from dasac_hip import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    B, C, h, H = 8, 19, 97, 769
    logits = torch.randn(B, C, h, h, device="cuda", generator=g) * 3
    up, probs, sums = ops.upsample_softmax(logits, (H, H), want_up=True, want_probs=True, want_sums=True)
    # softmax rows sum to one; the class prior sums add up to the pixel count
    assert float((probs.sum(1) - 1).abs().max()) < 1e-5
    assert float(sums.sum()) == pytest.approx(B * H * H, rel=1e-6)
    # interpolation never leaves the range of its four taps, and reproduces the corners exactly (align_corners=True)
    assert float(up.max()) <= float(logits.max()) and float(up.min()) >= float(logits.min())
    assert torch.equal(up[:, :, 0, 0], logits[:, :, 0, 0]) and torch.equal(up[:, :, -1, -1], logits[:, :, -1, -1])
    # constant planes stay constant; uniform logits cost log(C) on every labelled pixel, 0 on ignored ones
    const = torch.arange(C, device="cuda", dtype=torch.float32).view(1, C, 1, 1).expand(B, C, h, h).contiguous()
    upc, _, _ = ops.upsample_softmax(const, (H, H))
    assert float((upc - torch.arange(C, device="cuda").view(1, C, 1, 1)).abs().max()) < 1e-5
    labels = torch.randint(0, C, (B, H, H), device="cuda", generator=g)
    labels[:, :16] = 255
    loss, _, _ = ops.ce_loss(torch.zeros(B, C, H, H, device="cuda"), labels)
    assert float(loss) == pytest.approx(math.log(C) * (H - 16) / H, rel=1e-5)
    # gradient of the mean CE sums to zero over the classes of every pixel
    _, dl, _ = ops.ce_loss(up, labels, want_grad=True)
    assert float(dl.sum(1).abs().max()) < 1e-9 + 1e-6 * float(dl.abs().max())
    # identity affines: warping is the identity up to the fp32 round-off of affine_grid's normalised coordinates
    # (sample positions land within ~1e-4 px of the pixel centres), pooling T identical views returns the view,
    # warp-back returns it again
    eye = torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]], device="cuda").repeat(B, 1, 1)
    assert float((ops.warp_affine(probs, eye) - probs).abs().max()) < 1e-3
    same = probs[:2].repeat_interleave(4, dim=0).contiguous()            # 2 groups x 4 identical views
    pooled, mask, aligned = ops.warp_pool(same, eye, eye, 4)
    assert float((aligned - same).abs().max()) < 1e-3 and float(mask.min()) == 1.0
    assert float((pooled - probs[:2]).abs().max()) < 1e-3
    back = ops.warp_back(pooled, mask, eye, 4)
    assert float((back - same).abs().max()) < 2e-3


# 19 classes take the one-wave-per-segment kernel (rows walked by one block, next row prefetched): row ends W % 4 = 0..3 (the
# rotated last quad), up-factors 1 / 2 / 8, several segments and row chunks; W < 4, B = 1 with confidences, 7 classes and an
# up-factor of 16 (35 taps) fall back to the block-per-row kernel
@pytest.mark.parametrize("shape,mode", [((2, 19, 9, 13, 65, 97), 1), ((3, 19, 5, 7, 33, 49), 0), ((2, 19, 97, 97, 769, 769), 1),
                                        ((1, 7, 4, 6, 8, 12), 1), ((2, 19, 64, 128, 512, 1024), 0), ((2, 19, 5, 6, 38, 46), 1),
                                        ((2, 19, 7, 9, 50, 71), 0), ((1, 19, 33, 33, 33, 35), 0), ((2, 19, 3, 4, 10, 3), 1),
                                        ((2, 19, 20, 30, 40, 60), 1), ((1, 19, 5, 5, 65, 65), 0), ((1, 19, 9, 9, 65, 65), 1)])
def test_ce_backward_straight_into_the_low_resolution_gradient(shape, mode):
    """dasac_ce_loss_bwd_low == dasac_ce_loss(dlogits) followed by dasac_upsample_bwd (the two-kernel path it replaces,
    itself pinned by goldens g3 / g6) -- same weights, same summation order: bit for bit; and autograd through the
    engine's loss functions takes the fused path whenever the loss sits on an upsampled tensor."""
    from dasac_hip import ops
    from dasac_hip import engine as E
    B, C, h, w, Hh, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    low = (torch.randn(B, C, h, w, generator=g) * 2).cuda()
    y = torch.randint(0, C, (B, Hh, W), generator=g)
    y[torch.rand(B, Hh, W, generator=g) < 0.3] = 255
    y = y.cuda()
    conf = torch.rand(B, 1, Hh, W, generator=g).cuda() if mode else None
    cw = torch.rand(C, generator=g).cuda()
    gs = torch.tensor([0.7], device="cuda")
    up, _, _ = ops.upsample_softmax(low, (Hh, W))
    _, dl, _ = ops.ce_loss(up, y, cw, conf, want_grad=True, gscale=gs)
    ref = ops.upsample_bwd(dl, (h, w))
    got = ops.ce_loss_bwd_low(up, y, (h, w), cw, conf, gscale=gs)
    assert torch.equal(got, ref)
    # through autograd: loss(upsample(low)) backpropagates into `low` without a full-resolution gradient tensor
    lo1 = low.clone().requires_grad_(True)
    loss = E.focal_ce(E.upsample_bilinear(lo1, (Hh, W)), y, cw, conf)
    assert type(loss.grad_fn).__name__ == "_CELossLowBackward"
    (0.7 * loss).sum().backward()
    assert rel_err(lo1.grad, ref) < 1e-6
    lo2 = low.clone().requires_grad_(True)
    up2 = E.upsample_bilinear(lo2, (Hh, W))
    loss2 = E._CELoss.apply(up2, y, cw, conf)                # the unfused path stays available
    (0.7 * loss2).sum().backward()
    assert rel_err(lo2.grad, ref) < 1e-6 and rel_err(loss2, loss) < 1e-7


@pytest.mark.parametrize("shape", [(2, 19, 9, 13, 65, 97), (1, 19, 4, 6, 8, 12), (2, 19, 5, 5, 6, 7), (1, 5, 3, 4, 10, 3), (2, 19, 33, 33, 33, 35)])
def test_upsample_softmax_four_pixels_per_thread_edges(shape):
    """Row tails (W % 4 != 0), up-factors below 4 (more than three low-res columns under four pixels), factor 1."""
    from dasac_hip import ops
    B, C, h, w, Hh, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, C, h, w, generator=g) * 3
    ign = torch.rand(B, Hh, W, generator=g) < 0.2
    up, probs, sums = ops.upsample_softmax(x.cuda(), (Hh, W), ign.cuda(), want_probs=True, want_sums=True)
    ref_up = H.upsample_bilinear_ac(x, Hh, W)
    ref_p = torch.softmax(ref_up, 1)
    assert rel_err(up, ref_up) < TOL
    assert rel_err(sums, ref_p.sum((0, 2, 3))) < 1e-5
    assert rel_err(probs, ref_p * (~ign)[:, None]) < TOL
    # the teacher's path (no upsampled logits wanted): same kernel, `up` not written -- identical probabilities and sums
    _, probs2, sums2 = ops.upsample_softmax(x.cuda(), (Hh, W), ign.cuda(), want_up=False, want_probs=True, want_sums=True)
    assert torch.equal(probs2, probs) and torch.equal(sums2, sums)


def test_label_pad_mask_class_sums_and_dropout_planes():
    """The three small kernels that took the last ATen arithmetic off the step: sac.py:337-338 label preparation,
    sac.py:108 class sums (float64) and Dropout2d's per-plane noise (fcn.py:52,56)."""
    import torch
    from dasac_hip import ops
    g = torch.Generator().manual_seed(4)
    y = torch.randint(-1, 19, (3, 37, 41), generator=g)
    y[0, :5] = -1
    y[1, 3:9, 7:20] = 255
    want_mask, want_y = (y == -1), y.clone()
    want_y[want_mask] = 255
    yd = y.cuda()
    mask = ops.label_pad_mask(yd)
    assert mask.dtype == torch.bool and torch.equal(mask.cpu(), want_mask) and torch.equal(yd.cpu(), want_y)
    ys = y.cuda().transpose(1, 2)                                # strided view: still rewritten in place
    mask = ops.label_pad_mask(ys)
    assert torch.equal(mask.cpu(), want_mask.transpose(1, 2)) and torch.equal(ys.cpu(), want_y.transpose(1, 2))

    probs = torch.rand(2, 19, 33, 49, generator=g)
    sums = ops.class_sums(probs.cuda())
    assert sums.dtype == torch.float64 and sums.shape == (19,)
    assert torch.allclose(sums.cpu(), probs.double().sum((0, 2, 3)), rtol=1e-12, atol=0)

    torch.cuda.manual_seed(11)
    a = ops.dropout_planes(64, 4096, 0.1, torch.device("cuda", 0))
    b = ops.dropout_planes(64, 4096, 0.1, torch.device("cuda", 0))
    vals = set(a.unique().tolist())
    assert vals == {0.0, float(torch.tensor(1.0 / 0.9, dtype=torch.float32))}
    assert abs(float((a == 0).float().mean()) - 0.1) < 5e-3 and not torch.equal(a, b)       # Bernoulli(0.9), a new draw per call
    rows = (a == 0).float().mean(1)
    assert float(rows.std()) < 0.02                                                          # no structure across planes
    # the stream is torch's CUDA generator: re-seeding and get/set_rng_state (checkpoint resume) replay the masks
    torch.cuda.manual_seed(11)
    assert torch.equal(ops.dropout_planes(64, 4096, 0.1, torch.device("cuda", 0)), a)
    state = torch.cuda.get_rng_state(0)
    c = ops.dropout_planes(64, 4096, 0.1, torch.device("cuda", 0))
    assert torch.equal(c, b)
    torch.cuda.set_rng_state(state, 0)
    assert torch.equal(ops.dropout_planes(64, 4096, 0.1, torch.device("cuda", 0)), c)
    torch.cuda.manual_seed(12)
    assert not torch.equal(ops.dropout_planes(64, 4096, 0.1, torch.device("cuda", 0)), a)


def test_loss_shortcut_to_low_resolution_is_dropped_when_logits_up_is_edited_or_watched():
    """engine._ce differentiates a loss on an upsampled tensor w.r.t. the low-resolution logits directly; that is only valid
    while logits_up still IS the upsampling and nobody observes its gradient."""
    import torch
    from dasac_hip import engine as E
    g = torch.Generator().manual_seed(2)
    low0 = torch.randn(2, 19, 9, 13, generator=g).cuda()
    y = torch.randint(0, 19, (2, 65, 97), generator=g).cuda()

    def grads(edit, watch):
        low = low0.clone().requires_grad_(True)
        up = E.upsample_bilinear(low, (65, 97))
        if edit:
            up.mul_(2.0)
        if watch:
            up.retain_grad()
        E.ce_mean_all_pixels(up, y).backward()
        return low.grad.clone(), up.grad

    g_plain, none = grads(False, False)
    assert none is None
    g_watch, up_grad = grads(False, True)
    assert up_grad is not None and float(up_grad.abs().sum()) > 0          # the observer sees the loss gradient
    assert torch.allclose(g_watch, g_plain, rtol=1e-5, atol=1e-9)
    g_edit, _ = grads(True, False)                                          # loss of 2*U(low): NOT the gradient of loss(U(low))
    want = torch.autograd.grad(torch.nn.functional.cross_entropy(2 * torch.nn.functional.interpolate(
        (lr := low0.clone().cpu().requires_grad_(True)), (65, 97), mode="bilinear", align_corners=True), y.cpu()), lr)[0]
    assert torch.allclose(g_edit.cpu(), want, rtol=1e-3, atol=1e-8)
    assert not torch.allclose(g_edit, g_plain, rtol=1e-2)
