"""End-to-end parity of the drop-in `models` package (HIP engine) against vectors captured from the
reference (tests/golden) and against the CPU oracle.  Tolerance: north_star's 1e-3 rel (of max)."""
from types import SimpleNamespace as NS

import pytest
import torch
import torch.nn as nn

from oracle import nets_ref as N
from oracle.step_ref import DEFAULT_CFG
from conftest import rel_err

pytestmark = pytest.mark.gpu
T = torch.from_numpy
CRIT = nn.CrossEntropyLoss(ignore_index=255, reduction="none")


def sampled(t, n=64):
    flat = t.detach().reshape(-1)
    m = min(n, flat.numel())
    idx = (torch.arange(m, dtype=torch.int64, device=flat.device) * (flat.numel() - 1)) // max(m - 1, 1)
    return flat[idx]


def model_cfg(**kw):
    d = dict(DEFAULT_CFG)
    d.update(INIT_MODEL="", OPT_NESTEROV=False)
    d.update(kw)
    return NS(**d)


def test_resnet101_eval_bn_golden_g2(golden):
    import models
    g = golden("g2_resnet101")
    net = models.DeepLabV2_ResNet101(num_classes=19, criterion=CRIT, freeze_bn=True)
    net.load_state_dict(N.resnet101_state(seed=2, randomize_bn=True, he_init=True, residual_gain=0.25, aspp_gain=0.2), strict=True)
    net.cuda().train()
    losses, outs = net(T(g["eval_x"]).cuda(), T(g["eval_y"]).cuda())
    losses["loss_ce"].mean().backward()
    assert rel_err(outs["logits"], g["eval_logits"]) < 1e-4
    assert rel_err(sampled(outs["logits_up"], 512), g["eval_logits_up_s"]) < 1e-4
    assert rel_err(losses["loss_ce"], g["eval_loss"]) < 1e-5
    named = dict(net.named_parameters())
    worst = []
    for k in [k[len("eval_g_"):] for k in g.files if k.startswith("eval_g_")]:
        gn, gm = float(g["eval_gn_" + k]), float(g["eval_gm_" + k])
        assert abs(float(named[k].grad.norm()) - gn) < 1e-3 * gn, k
        assert abs(float(named[k].grad.abs().max()) - gm) < 1e-3 * gm, k
        worst.append((float((sampled(named[k].grad).cpu() - T(g["eval_g_" + k])).abs().max()) / gm, k))
    worst.sort(reverse=True)
    print("g2 eval: sampled gradient error / tensor max:", worst[:3])
    assert worst[0][0] < 1e-3, worst[:3]          # north_star: 1e-3 of the tensor max


def _all_grads(seed, share_masks):
    """HIP gradients of all 320 parameters vs oracle autograd.  share_masks: the oracle's ReLUs take their on/off
    pattern from the HIP forward activations (MaskedRelu), so that the comparison is about the arithmetic and not
    about which side of zero a round-off-sized pre-activation landed on."""
    import models
    from dasac_hip import engine as E
    sd = N.resnet101_state(seed=seed, randomize_bn=True, he_init=True, residual_gain=0.25, aspp_gain=0.2)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(2, 3, 41, 57, generator=g)
    y = torch.randint(0, 19, (2, 41, 57), generator=g)
    y[:, :3] = 255
    net = models.DeepLabV2_ResNet101(num_classes=19, criterion=CRIT, freeze_bn=True)
    net.load_state_dict(sd, strict=True)
    net.cuda().train()
    l2, _ = net(x.cuda(), y.cuda())
    l2["loss_ce"].mean().backward()
    act, kw = None, {}
    if share_masks:
        eng = net._engine
        _, saved = eng.forward(x.cuda(), keep=True)
        masks = [(saved["acts"][op.dst] > 0).cpu() for op in eng.plan.ops if op.kind == "conv" and op.relu]
        act = N.MaskedRelu(masks)
        kw = dict(act=act)
    ref = {k: v.clone() for k, v in sd.items()}
    for k in N.trainable_keys(ref):
        ref[k].requires_grad_(True)
    losses, _ = N.segnet_forward("deeplabv2_resnet101", ref, x, y, **kw)
    losses["loss_ce"].sum().backward()
    assert rel_err(l2["loss_ce"], losses["loss_ce"]) < 1e-5
    errs = sorted(((rel_err(p.grad, ref[k].grad), k) for k, p in net.named_parameters()), reverse=True)
    return errs, act


@pytest.mark.parametrize("seed", [1, 5])
def test_resnet101_all_gradients_vs_oracle(seed):
    """Every one of the 320 parameter gradients against oracle autograd (frozen BN) at fp32 round-off level.
    The oracle's ReLUs reuse the HIP forward's on/off pattern; the two forwards disagree on it for at most a
    few units in ~5 million (pre-activations within ~1e-6 of zero)."""
    errs, act = _all_grads(seed, share_masks=True)
    assert len(errs) == 320
    assert act.disagree <= 1e-5 * act.total, (act.disagree, act.total)
    assert errs[0][0] < 2e-5, errs[:3]


class _RecordingRelu:
    """ReLU that keeps its pre-activations (call order = the engine's fused conv+ReLU op order)."""

    def __init__(self):
        self.z = []

    def __call__(self, z):
        self.z.append(z.detach())
        return torch.relu(z)


def _oracle_grads(sd, x, y, dtype, act=None, bn_train=False):
    ref = {k: (v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    for k in N.trainable_keys(ref):
        ref[k].requires_grad_(True)
    kw = {} if act is None else dict(act=act)
    losses, _ = N.segnet_forward("deeplabv2_resnet101", ref, x.to(dtype), y, bn_train=bn_train, **kw)
    losses["loss_ce"].sum().backward()
    return {k: ref[k].grad for k in N.trainable_keys(ref)}, float(losses["loss_ce"].detach())


@pytest.mark.parametrize("seed,bn_train", [(1, False), (2, False), (3, False), (5, False), (8, False), (2, True), (6, True)])
def test_resnet101_gradients_fp64_arbitration(seed, bn_train):
    """WITHOUT sharing ReLU patterns the HIP gradients and ATen-CPU fp32 autograd differ by up to ~1e-2 of a tensor's
    max on a few parameters.  A float64 run of the oracle arbitrates:
      1. the ReLU units on which the HIP forward and the fp64 forward disagree are a handful in ~5 million, and every
         one of them has an fp64 pre-activation within fp32 round-off of zero (|z| <= 1e-5 of its layer's max): which
         side of zero such a unit lands on is decided by the summation order of ANY fp32 implementation (ATen's own
         fp32 run has such units too -- just different ones);
      2. with the pattern fixed to what each fp32 implementation actually computed, both sit at fp32 round-off from
         the fp64 gradients, per parameter and relative to that parameter's max (north_star's metric), and the HIP
         path is no further from fp64 than twice ATen's fp32 (+1e-5).
    So the un-shared comparison measures borderline units, not arithmetic; every gradient assertion elsewhere in tests/
    that is looser than 1e-3 of the max cites this test.
    bn_train: the same with batch-statistics BN (baseline / AdaBN mode, 2 x 33 x 49 crops as in golden g2 'train'): a
    handful of samples per channel at stride 8 make the normalisation ill-conditioned, so BOTH fp32 implementations sit
    further from fp64 -- the bound is relative to ATen's own fp32 error, whatever that is."""
    import models
    sd = N.resnet101_state(seed=seed, randomize_bn=True, he_init=True, residual_gain=0.25, aspp_gain=0.2)
    g = torch.Generator().manual_seed(seed)
    hw = (33, 49) if bn_train else (41, 57)
    x = torch.randn(2, 3, *hw, generator=g)
    y = torch.randint(0, 19, (2,) + hw, generator=g)
    y[:, :3] = 255
    net = models.DeepLabV2_ResNet101(num_classes=19, criterion=CRIT, freeze_bn=not bn_train)
    net.load_state_dict(sd, strict=True)
    net.cuda().train()
    l_hip, _ = net(x.cuda(), y.cuda())
    l_hip["loss_ce"].mean().backward()
    hip = {k: p.grad.double().cpu() for k, p in net.named_parameters()}
    eng = net._engine
    _, saved = eng.forward(x.cuda(), keep=True)
    hip_masks = [(saved["acts"][op.dst] > 0).cpu() for op in eng.plan.ops if op.kind == "conv" and op.relu]

    # fp64, free-running: where do the patterns differ, and how close to zero are those units?
    rec64 = _RecordingRelu()
    g64, l64 = _oracle_grads(sd, x, y, torch.float64, rec64, bn_train)
    rec32 = _RecordingRelu()
    g32, _ = _oracle_grads(sd, x, y, torch.float32, rec32, bn_train)
    assert abs(float(l_hip["loss_ce"]) - l64) < 1e-5 * abs(l64)
    total = flips_hip = flips_aten = 0
    worst = 0.0
    for z64, m_hip, z32 in zip(rec64.z, hip_masks, rec32.z):
        total += z64.numel()
        d = (z64 > 0) != m_hip
        flips_hip += int(d.sum())
        flips_aten += int(((z64 > 0) != (z32 > 0)).sum())
        if d.any():
            worst = max(worst, float(z64[d].abs().max() / z64.abs().max()))
    print("fp64 arbitration%s: %d ReLU units, HIP flips %d (worst |z|/max %.2e), ATen-fp32 flips %d" % (
        " [train-mode BN]" if bn_train else "", total, flips_hip, worst, flips_aten))
    assert len(rec64.z) == len(hip_masks) and total > 2e6
    assert flips_hip <= (1e-4 if bn_train else 1e-5) * total and worst <= (1e-4 if bn_train else 1e-5)

    # each fp32 implementation against fp64 with ITS OWN pattern
    g64_hip, _ = _oracle_grads(sd, x, y, torch.float64, N.MaskedRelu(hip_masks), bn_train)
    g64_aten, _ = _oracle_grads(sd, x, y, torch.float64, N.MaskedRelu([z > 0 for z in rec32.z]), bn_train)
    rows = []
    for k in hip:
        e_hip = rel_err(hip[k], g64_hip[k])
        e_aten = rel_err(g32[k], g64_aten[k])
        rows.append((e_hip / (2 * e_aten + 1e-5), e_hip, e_aten, k))
    rows.sort(reverse=True)
    free = max(rel_err(hip[k], g64[k]) for k in hip)
    free_aten = max(rel_err(g32[k], g64[k]) for k in hip)
    print("fp64 arbitration, worst parameters (ratio, err HIP, err ATen-fp32):", rows[:3], "free-running max err: HIP %.2e ATen %.2e" % (free, free_aten))
    assert len(rows) == 320
    assert rows[0][0] <= 1.0, rows[:3]
    if not bn_train:
        assert max(r[1] for r in rows) < 2e-5          # measured 1.1e-6 (ATen fp32: 4e-7)
    # and the free-running comparison really is explained by the flips: un-shared error >> shared error only if flips exist
    # measured: frozen BN -- HIP 1e-6 .. 1.7e-2, ATen-fp32 7e-7 .. 1.7e-2 (whoever has flipped units is off, either side);
    # batch-statistics BN at 2 x 33 x 49 -- BOTH 0.09 .. 0.14 with 3-8 flipped units each (ill-conditioned statistics)
    assert free < (0.5 if bn_train else 5e-2) and (flips_hip > 0 or free < (1e-2 if bn_train else 1e-4)), (free, flips_hip)


def test_resnet101_gradients_with_borderline_relu():
    """The free-running comparison documented (bound explained by test_resnet101_gradients_fp64_arbitration): the typical
    parameter agrees to ~1e-3 of its max, single tensors upstream of a flipped unit move by up to ~1e-2; nothing blows up."""
    errs, _ = _all_grads(5, share_masks=False)
    med = errs[len(errs) // 2][0]
    assert med < 5e-3 and errs[0][0] < 5e-2, (errs[:3], med)


def test_vgg16_deeplab_cfg1_golden_g10(golden):
    import models
    g = golden("g10_vgg16_deeplab")
    net = models.DeepLabV2_VGG16(num_classes=19, criterion=CRIT, use_bn=True, freeze_bn=True)
    net.load_state_dict(N.deeplab_vgg16_state(seed=10, randomize_bn=True), strict=True)
    net.cuda().train()
    gen = torch.Generator().manual_seed(10)
    x = torch.randn(1, 3, 321, 321, generator=gen)
    y = torch.randint(0, 19, (1, 321, 321), generator=gen)
    losses, outs = net(x.cuda(), y.cuda())
    losses["loss_ce"].mean().backward()
    assert rel_err(outs["logits"], g["logits"]) < 1e-4
    assert rel_err(losses["loss_ce"], g["loss"]) < 1e-5
    named = dict(net.named_parameters())
    errs = (rel_err(sampled(named["features.0.weight"].grad), g["g_first"]), rel_err(sampled(named["features.42.weight"].grad), g["g_fc6"]),
            rel_err(named["classifier.conv2d_list.2.bias"].grad, g["g_cls_bias"]))
    print("g10 vgg16-deeplab gradient errors over tensor max:", errs)
    assert max(errs) < 1e-3                                  # north_star: grads within 1e-3 rel


def test_fcn8s_golden_g10(golden):
    import models
    g = golden("g10_fcn8s")
    net = models.VGG16_FCN8s(19, criterion=CRIT, use_bn=True, freeze_bn=True, drop_rate=0.0)
    net.load_state_dict(N.fcn8s_vgg16_state(seed=12, randomize_bn=True), strict=True)
    net.cuda().train()
    losses, outs = net(T(g["x"]).cuda(), T(g["y"]).cuda())
    assert set(outs) == {"logits_up"}
    losses["loss_ce"].mean().backward()
    assert rel_err(outs["logits_up"], g["logits_up"]) < 1e-4
    assert rel_err(losses["loss_ce"], g["loss"]) < 1e-5
    named = dict(net.named_parameters())
    errs = (rel_err(sampled(named["vgg_head.0.weight"].grad), g["g_head0"]), rel_err(named["score_pool3.weight"].grad.reshape(-1)[:64], g["g_sp3"]),
            rel_err(sampled(named["block1.0.weight"].grad), g["g_first"]))
    print("g10 fcn8s gradient errors over tensor max:", errs)
    assert max(errs) < 1e-3                                  # north_star: grads within 1e-3 rel


def test_fcn8s_dropout2d_with_injected_masks_vs_oracle():
    """fcn.py:52,56 Dropout2d(p=.1) in train mode: with the same per-(n, c) keep masks injected on both sides
    (`module.keep_mask` here, `drop_masks` in the oracle) forward and gradients must agree (K19)."""
    import models
    sd = N.fcn8s_vgg16_state(seed=12, randomize_bn=True)
    net = models.VGG16_FCN8s(19, criterion=CRIT, use_bn=True, freeze_bn=True, drop_rate=0.1)
    net.load_state_dict(sd, strict=True)
    net.cuda().train()
    drops = [m for m in net.vgg_head if isinstance(m, nn.Dropout2d)]
    assert len(drops) == 2 and all(m.p == 0.1 and m.training for m in drops)
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, 3, 64, 96, generator=g)
    y = torch.randint(0, 19, (2, 64, 96), generator=g)
    masks = [(torch.rand(2, 4096, generator=g) >= 0.1).float() / 0.9 for _ in drops]
    assert all(0 < float((m == 0).float().mean()) < 0.2 for m in masks)
    for m, k in zip(drops, masks):
        m.keep_mask = k.cuda()
    losses, outs = net(x.cuda(), y.cuda())
    losses["loss_ce"].mean().backward()
    ref = {k: v.clone() for k, v in sd.items()}
    for k in N.trainable_keys(ref):
        ref[k].requires_grad_(True)
    l_ref, o_ref = N.segnet_forward("fcn_vgg16_bn", ref, x, y, drop_masks=[k.view(2, 4096, 1, 1) for k in masks])
    l_ref["loss_ce"].sum().backward()
    assert rel_err(outs["logits_up"], o_ref["logits_up"]) < 1e-4
    assert rel_err(losses["loss_ce"], l_ref["loss_ce"]) < 1e-5
    named = dict(net.named_parameters())
    for k in ("vgg_head.0.weight", "vgg_head.1.weight", "vgg_head.4.weight", "vgg_head.4.bias", "vgg_head.8.weight", "block3.40.weight"):
        assert rel_err(named[k].grad, ref[k].grad) < 1e-3, k
    # without an injected mask the draw is the module's own: a different pattern, same expectation machinery
    for m in drops:
        m.keep_mask = None
    l2, _ = net(x.cuda(), y.cuda())
    assert float(l2["loss_ce"]) != pytest.approx(float(losses["loss_ce"]), rel=1e-6)
    net.eval()
    with torch.no_grad():
        e1, _ = net(x.cuda())
        e2, _ = net(x.cuda())
    assert torch.equal(e1, e2)


def test_sac_pool_functions_callable_and_engine_follows_replaced_parameters():
    """API surface of sac.py:49,218-269: `pool_func` is callable on its own; and an Engine captured before a parameter
    OBJECT was replaced (load_state_dict(assign=True) / m.weight = nn.Parameter(...)) must not keep running the old one."""
    import models
    from oracle import head_ref as H
    cfg = model_cfg()
    net = models.get_model(cfg, 0, num_classes=19, criterion=CRIT).cuda().train()
    g = torch.Generator().manual_seed(77)
    aligned = torch.softmax(torch.randn(4, 19, 17, 23, generator=g) * 2, 1) * torch.rand(4, 1, 17, 23, generator=g)
    aligned[1, :, :3] = 0.0
    pooled, mask = net.pool_func(aligned.cuda(), 2)
    p_ref, m_ref = H.avg_pool_views(aligned, 2)
    assert net.pool_func == net._avg_pool and rel_err(pooled, p_ref) < 1e-6 and torch.equal(mask.cpu(), m_ref)
    a2 = aligned.clone().cuda()
    pooled, mask = net._minentropy_pool(a2, 2)
    p_ref, m_ref = H.minentropy_pool_views(aligned.clone(), 2)
    assert rel_err(pooled, p_ref) < 1e-6 and torch.equal(mask.cpu(), m_ref) and pooled.data_ptr() == a2.data_ptr()
    with pytest.raises(RuntimeError):
        net._minentropy_pool(aligned[:3].cuda(), 2)
    # replaced parameter object
    x = torch.randn(1, 3, 33, 49, device="cuda")
    with torch.no_grad():
        l0, _ = net.backbone(x)
        conv = net.backbone.model.layer5.conv2d_list[0]
        conv.weight = nn.Parameter(conv.weight.detach() * 50.0)
        l1, _ = net.backbone(x)
    assert rel_err(l1, l0) > 1e-2


def test_two_sac_training_steps_golden_g8(golden):
    """train.py:266-298 twice on the HIP path vs the reference's own two steps."""
    import models
    import driver
    g = golden("g8_two_steps")
    cfg = model_cfg()
    net = models.get_model(cfg, 0, num_classes=19, criterion=CRIT)
    net.backbone.load_state_dict(N.resnet101_state(seed=8, randomize_bn=True, he_init=True, residual_gain=0.25, aspp_gain=0.2), strict=True)
    net.cuda().train()
    optim = driver.make_optimizer(net, cfg)
    aff, inv = T(g["affine"]).cuda(), T(g["affine_inv"]).cuda()
    for it in range(2):
        src = (T(g["it%d_xs" % it]).cuda(), T(g["it%d_ys" % it]).cuda())
        tgt = (T(g["it%d_f1" % it]).cuda(), T(g["it%d_gt" % it]).cuda(), T(g["it%d_f2" % it]).cuda(), aff, inv)
        ls, lt, outs = driver.sac_train_iteration(net, optim, src, tgt, int(g["T"]), it % 100 == 0, cfg.LR_TARGET)
        assert float(ls["loss_ce"]) == pytest.approx(float(g["it%d_src_loss" % it].item()), rel=1e-4)
        for k in ("loss_ce", "self_ce", "teacher_diff"):
            assert float(lt[k]) == pytest.approx(float(g["it%d_%s" % (it, k)].item()), rel=2e-3, abs=1e-6), (it, k)
        mism = (outs["teacher_labels"].to(torch.uint8).cpu() != T(g["it%d_labels" % it])).float().mean()
        assert mism < 1e-3, (it, float(mism))
        assert outs["teacher_labels"].dtype == torch.int64
        assert rel_err(net.running_conf, g["it%d_chi" % it]) < 1e-4
        st = net.backbone.state_dict()
        for k in [k[len("it0_p_"):] for k in g.files if k.startswith("it0_p_")]:
            assert rel_err(sampled(st[k]), g["it%d_p_%s" % (it, k)]) < 1e-4, (it, k)
        for key in ("logits_up", "logits", "teacher_init", "teacher_refined", "teacher_conf", "teacher_labels",
                    "running_conf", "teacher_aligned", "frames_aligned"):
            assert outs[key].is_contiguous(), key


def test_inference_paths_and_no_grad():
    import models
    cfg = model_cfg()
    net = models.get_model(cfg, 0, num_classes=19, criterion=CRIT).cuda().eval()
    x = torch.randn(1, 3, 33, 49, device="cuda")
    with torch.no_grad():
        logits, up = net(x, teacher=False)
        l2, up2 = net(x, teacher=True)
    assert logits.shape == (1, 19, 5, 7) and up.shape == (1, 19, 33, 49) and l2.shape == logits.shape


def test_three_sac_iterations_with_teacher_updates_vs_oracle():
    """train.py:266-298 three times with `update_teacher` on EVERY iteration (NET_MOMENTUM .5, 8x the reference LR so
    that student and teacher visibly move): EMA kernel -> teacher forward -> pseudo labels -> fused SGD must track the
    CPU oracle step by step (a teacher or student running on stale packed weights shows up from iteration 1 on)."""
    import models
    import driver
    from oracle import step_ref as S
    kw = dict(NET_MOMENTUM=0.5, LR=2e-3)
    cfg = model_cfg(**kw)
    sd = N.resnet101_state(seed=1, randomize_bn=True, he_init=True, residual_gain=0.25, aspp_gain=0.2)
    ref = S.SacOracle(sd, cfg=dict(S.DEFAULT_CFG, **kw))
    ref_opt = S.SgdOracle(ref)
    net = models.get_model(cfg, 0, num_classes=19, criterion=CRIT)
    net.backbone.load_state_dict(sd, strict=True)
    net.cuda().train()
    optim = driver.make_optimizer(net, cfg)
    to = lambda ts: tuple(t.cuda() for t in ts)
    seen = []
    for it in range(3):
        src, tgt = driver.synthetic_batches(2, 1, 2, (33, 49), "cpu", seed=20 + it)
        ls_r, lt_r, outs_r = S.sac_train_iteration(ref, ref_opt, src, tuple(t.clone() for t in tgt), 2, True)
        ls, lt, outs = driver.sac_train_iteration(net, optim, to(src), to(tgt), 2, True, cfg.LR_TARGET)
        assert float(ls["loss_ce"]) == pytest.approx(ls_r["loss_ce"], rel=2e-3), it
        assert float(lt["teacher_diff"]) == pytest.approx(lt_r["teacher_diff"], rel=2e-3, abs=1e-6), it
        assert float(lt["self_ce"]) == pytest.approx(lt_r["self_ce"], rel=2e-2, abs=1e-5), it
        mism = (outs["teacher_labels"].cpu() != outs_r["teacher_labels"]).float().mean()
        assert float(mism) < 5e-3, (it, float(mism))
        seen.append(float(lt["teacher_diff"]))
    assert seen[0] == 0.0 and seen[1] > 0.0 and seen[2] > 0.0
    w, w_ref = net.slow_net.state_dict()["model.layer4.2.conv3.weight"].cpu(), ref.teacher["model.layer4.2.conv3.weight"].detach()
    assert rel_err(w, w_ref) < 1e-3


@pytest.mark.parametrize("wrapped", [False, True])
def test_fused_iteration_equals_the_two_pass_iteration_and_the_oracle(wrapped):
    """driver.sac_train_iteration(fuse_passes=True) -- the student evaluated ONCE on [source; target] crops, one backward pass
    over loss_ce + LR_TARGET * self_ce (SAC.forward_fused) -- against the reference's two passes (train.py:266-298) on the same
    module, and against the CPU oracle's two passes: same losses, same label maps, same parameters after the step (the
    weight-gradient reductions run over 4 crops at once instead of 2 + 2: summation order only)."""
    import models
    import driver
    from oracle import step_ref as S
    from dasac_hip.parallel import OverlappedDataParallel
    kw = dict(NET_MOMENTUM=0.5, LR=2e-3)
    cfg = model_cfg(**kw)
    sd = N.resnet101_state(seed=4, randomize_bn=True, he_init=True, residual_gain=0.25, aspp_gain=0.2)
    ref = S.SacOracle(sd, cfg=dict(S.DEFAULT_CFG, **kw))
    ref_opt = S.SgdOracle(ref)
    nets, opts, steps = [], [], []
    for fused in (False, True):
        net = models.get_model(cfg, 0, num_classes=19, criterion=CRIT)
        net.backbone.load_state_dict(sd, strict=True)
        net.cuda().train()
        nets.append(net)
        opts.append(driver.make_optimizer(net, cfg))
        steps.append(OverlappedDataParallel(net, device_ids=[0]) if (wrapped and fused) else net)
    to = lambda ts: tuple(t.cuda() for t in ts)
    for it in range(3):
        src, tgt = driver.synthetic_batches(2, 1, 2, (33, 49), "cpu", seed=40 + it)
        ls_r, lt_r, outs_r = S.sac_train_iteration(ref, ref_opt, src, tuple(t.clone() for t in tgt), 2, it != 1)
        got = [driver.sac_train_iteration(steps[f], opts[f], to(src), to(tgt), 2, it != 1, cfg.LR_TARGET, fuse_passes=bool(f))
               for f in (0, 1)]
        (ls0, lt0, o0), (ls1, lt1, o1) = got
        assert set(lt1) == set(lt0) == {"loss_ce", "self_ce", "teacher_diff"} and set(o1) == set(o0)
        for key in o1:
            assert o1[key].is_contiguous() and o1[key].shape == o0[key].shape and o1[key].dtype == o0[key].dtype, key
        # the teacher side does not depend on how the student is batched: bit-equal on equal weights (iteration 0)
        if it == 0:
            assert torch.equal(o0["teacher_labels"], o1["teacher_labels"]) and torch.equal(o0["teacher_refined"], o1["teacher_refined"])
            assert float(ls0["loss_ce"]) == pytest.approx(float(ls1["loss_ce"]), rel=1e-5)
            assert float(lt0["self_ce"]) == pytest.approx(float(lt1["self_ce"]), rel=1e-5)
            assert float(lt0["loss_ce"]) == pytest.approx(float(lt1["loss_ce"]), rel=1e-5)
            # one optimiser step from equal weights: the two schedules differ by the summation order of the weight-gradient
            # reductions only (4 crops at once instead of 2 + 2) -- and both sit on the oracle's parameters
            sd0, sd1 = nets[0].state_dict(), nets[1].state_dict()
            for k in sd0:
                if sd0[k].is_floating_point():
                    assert rel_err(sd1[k], sd0[k]) < 1e-5, k
            for k in ("model.conv1.weight", "model.layer3.7.conv2.weight", "model.layer4.2.bn3.weight", "model.layer5.conv2d_list.2.weight"):
                assert rel_err(nets[1].backbone.state_dict()[k], ref.student[k].detach()) < 1e-4, k
        assert float(ls1["loss_ce"]) == pytest.approx(ls_r["loss_ce"], rel=2e-3), it
        assert float(lt1["loss_ce"]) == pytest.approx(lt_r["loss_ce"], rel=2e-3), it
        assert float(lt1["teacher_diff"]) == pytest.approx(lt_r["teacher_diff"], rel=2e-3, abs=1e-6), it
        assert float(lt1["self_ce"]) == pytest.approx(lt_r["self_ce"], rel=2e-2, abs=1e-5), it
        assert float((o1["teacher_labels"].cpu() != outs_r["teacher_labels"]).float().mean()) < 5e-3, it
    # three free-running optimiser steps at 8x the reference LR later the two schedules have drifted apart by what borderline
    # ReLUs / pseudo-labels do to any two fp32 summation orders (the oracle comparison of the existing three-iteration test
    # drifts the same way): a loose bound here, the tight one was taken right after the first step above
    sd0, sd1 = nets[0].state_dict(), nets[1].state_dict()
    for k in sd0:
        if sd0[k].is_floating_point():
            assert rel_err(sd1[k], sd0[k]) < 2e-2, k
    for k in ("model.conv1.weight", "model.layer3.7.conv2.weight", "model.layer4.2.bn3.weight", "model.layer5.conv2d_list.2.weight"):
        assert rel_err(nets[1].backbone.state_dict()[k], ref.student[k].detach()) < 2e-2, k
