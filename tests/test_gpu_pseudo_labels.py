"""K14 parity (bit-exact) against the oracle and the reference-captured golden, through the C ABI."""
import pytest
import torch

from oracle import head_ref as H

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _run(probs, ignore, upper, lower, disc):
    from dasac_hip import ops
    lab, conf, idx = ops.pseudo_labels(probs.cuda(), None if ignore is None else ignore.cuda(), upper, lower,
                                       None if disc is None else disc.cuda(), want_idx=True)
    return lab.cpu(), conf.cpu(), idx.cpu()


def test_golden_g5_bit_exact(golden):
    g = golden("g5_pseudo_labels")
    probs, ignore = T(g["probs"]), T(g["ignore"])
    for tag, disc in (("disc", T(g["discount"])), ("nodisc", None)):
        lab, conf, idx = _run(probs, ignore, float(g["upper"]), float(g["lower"]), disc)
        assert torch.equal(lab, T(g["labels_" + tag]))
        assert torch.equal(conf, T(g["conf_" + tag]))
        assert torch.equal(idx, T(g["idx_" + tag]))


@pytest.mark.parametrize("shape", [(1, 19, 1, 1), (3, 19, 17, 23), (2, 19, 65, 97), (5, 7, 33, 300), (2, 19, 769, 769)])
def test_random_bit_exact_vs_oracle(shape):
    g = torch.Generator().manual_seed(sum(shape))
    B, C, Hh, W = shape
    probs = torch.softmax(torch.randn(shape, generator=g) * 4, 1)
    probs[0, :, 0, 0] = 0                                       # fully masked pixel
    ignore = torch.rand(B, Hh, W, generator=g) < 0.1
    disc = torch.rand(C, generator=g)
    for d in (disc, None):
        for ign in (ignore, None):
            lab, conf, idx = _run(probs, ign, 0.75, 0.2, d)
            rl, rc, ri = H.pseudo_labels(probs, ign if ign is not None else torch.zeros(B, Hh, W, dtype=torch.bool), 0.75, 0.2, d)
            assert torch.equal(lab, rl) and torch.equal(conf, rc) and torch.equal(idx, ri)
    assert Hh * W == 1 or int((rl != 255).sum()) > 0


def test_full_size_properties():
    """cfg-3 size [8,19,769,769]: idempotence + label/argmax consistency + sort-free threshold property."""
    from dasac_hip import ops
    g = torch.Generator(device="cuda").manual_seed(1)
    probs = torch.softmax(torch.randn(8, 19, 769, 769, device="cuda", generator=g) * 3, 1)
    lab, conf, idx = ops.pseudo_labels(probs, None, 0.75, 0.2, None, want_idx=True)
    lab2, conf2, _ = ops.pseudo_labels(probs, None, 0.75, 0.2, None, want_idx=True)
    assert torch.equal(lab, lab2) and torch.equal(conf, conf2)
    m, k = probs.max(1, keepdim=True)
    assert torch.equal(conf, m) and torch.equal(idx, k)
    keep = lab != 255
    assert torch.equal(lab[keep], idx[:, 0][keep])
    # every kept pixel beats max(0.75*peak, 0.2) of its (image, class); every dropped one does not
    peak = torch.zeros(8, 19, device="cuda").scatter_reduce_(1, k.view(8, -1), m.view(8, -1), reduce="amax")
    thr = (peak * 0.75).clamp_min(0.2).gather(1, k.view(8, -1)).view_as(m)
    assert torch.equal(keep, (m > thr)[:, 0])


def test_errors_are_reported():
    from dasac_hip import ops, DasacError
    with pytest.raises(DasacError):
        ops.pseudo_labels(torch.rand(1, 19, 4, 4, device="cuda"), None, 0.75, 0.0)


def test_labels_bit_equal_through_the_module_threshold_path(golden):
    """sac.py:151-187 end to end on injected probabilities: the per-class discount 1 - exp(-chi/beta) comes from chi
    through the product's own path (a 19-float host round trip evaluated by the CPU ATen kernels the reference uses), so
    the thresholds and with them the int64 label map equal the reference's BIT FOR BIT -- golden g5 (captured from
    the reference) and fresh seeded cases against the oracle."""
    from types import SimpleNamespace as NS
    import torch.nn as nn
    import models
    from oracle import head_ref as H
    from oracle.step_ref import DEFAULT_CFG
    g = golden("g5_pseudo_labels")
    cfg = NS(**dict(DEFAULT_CFG, INIT_MODEL="", OPT_NESTEROV=False))
    net = models.get_model(cfg, 0, num_classes=19, criterion=nn.CrossEntropyLoss(ignore_index=255, reduction="none")).cuda().train()
    T = torch.from_numpy
    net.running_conf.copy_(T(g["chi"]))
    disc_here = net._threshold_discount().cpu()
    assert torch.equal(disc_here, H.threshold_discount(T(g["chi"]), cfg.THRESHOLD_BETA))       # same host, same ATen kernels
    assert torch.allclose(disc_here, T(g["discount"]), rtol=3e-7, atol=0)    # golden from another host: <= 1-2 ulp (Sleef ISA paths)
    for tag, disc in (("disc", True), ("nodisc", False)):
        lab, conf, idx = net._pseudo_labels_probs(T(g["probs"]).cuda(), T(g["ignore"]).cuda(), disc)
        assert torch.equal(lab.cpu(), T(g["labels_" + tag])) and torch.equal(idx.cpu(), T(g["idx_" + tag]))
        assert torch.equal(conf.cpu(), T(g["conf_" + tag]))
    gen = torch.Generator().manual_seed(123)
    for trial in range(4):
        chi = torch.rand(19, generator=gen) * (0.004 if trial % 2 else 0.2)
        probs = torch.softmax(torch.randn(2, 19, 97, 129, generator=gen) * (2 + trial), 1)
        ignore = torch.rand(2, 97, 129, generator=gen) < 0.05
        net.running_conf.copy_(chi)
        lab, conf, _ = net._pseudo_labels_probs(probs.cuda(), ignore.cuda(), True)
        ref_lab, ref_conf, _ = H.pseudo_labels(probs, ignore, cfg.RUN_CONF_UPPER, cfg.RUN_CONF_LOWER, H.threshold_discount(chi, cfg.THRESHOLD_BETA))
        assert torch.equal(lab.cpu(), ref_lab) and torch.equal(conf.cpu(), ref_conf), trial
        assert torch.equal(net._focal_weight(cfg.FOCAL_P).cpu(), H.focal_weight(chi, cfg.FOCAL_P))
        assert int((ref_lab != 255).sum()) > 0


def test_device_thresholds_stay_within_one_ulp_of_the_host_path_and_need_no_host_copy():
    """`SAC.device_thresholds = True` (round 6, opt-in): 1 - exp(-chi/beta) and (1 - chi)^p from the class-prior kernel itself
    (fp64 exp rounded once) instead of the 19-float host round trip.  Not bit-equal to the host's ATen by construction (Sleef's
    1-ULP expf for full vectors, libm for vector tails: which class takes which depends on the host's vector width), but
    within one unit in the last place of exp(.) over a dense sweep of chi -- and a whole refinement + labelling pass then
    issues no device-to-host copy."""
    from types import SimpleNamespace as NS
    import torch.nn as nn
    import models
    from dasac_hip import ops
    from oracle.step_ref import DEFAULT_CFG
    cfg = NS(**dict(DEFAULT_CFG, INIT_MODEL="", OPT_NESTEROV=False))
    beta = cfg.THRESHOLD_BETA
    gen = torch.Generator().manual_seed(9)
    worst = 0.0
    for scale in (2e-3, 2e-2, 1.0):                        # chi around beta (where the discount moves) and up to 1
        for _ in range(64):
            chi = (torch.rand(19, generator=gen) * scale).cuda()
            disc_d, fw_d = ops.class_state(chi.clone(), None, 1, 1, beta, cfg.STAT_MOMENTUM, False, cfg.FOCAL_P, want_disc=True, want_focal=True)
            disc_h, fw_h = ops.class_vectors(chi, beta, cfg.FOCAL_P)
            e_h = torch.exp(-chi.cpu() / beta)
            ulp = torch.maximum(e_h.abs() * 2.0 ** -23, torch.tensor(2.0 ** -149))
            worst = max(worst, float(((disc_d.cpu() - disc_h.cpu()).abs() / torch.maximum(ulp, torch.tensor(2.0 ** -24))).max()))
            assert torch.equal(fw_d.cpu(), fw_h.cpu())      # (1 - chi)^3 is two fp32 multiplications on both sides
    assert worst <= 1.0, worst
    # the module path: same labels up to threshold ties, and no D2H copy while refining + labelling
    net = models.get_model(cfg, 0, num_classes=19, criterion=nn.CrossEntropyLoss(ignore_index=255, reduction="none")).cuda().train()
    probs_logits = (torch.randn(4, 19, 13, 17, generator=gen) * 3).cuda()
    frames = torch.randn(4, 3, 97, 129, generator=gen).cuda()
    import driver
    theta, inv = driver.view_affines(driver.BENCH_VIEWS, 97, 129)
    ignore = (torch.rand(4, 97, 129, generator=gen) < 0.05).cuda()
    out = {}
    for dev_thr in (False, True):
        net.device_thresholds = dev_thr
        net.running_conf.fill_(0.05)
        refined, _ = net._refine(frames, probs_logits, 4, theta.cuda(), inv.cuda(), ignore, pool=True)
        disc, fw = net._class_vectors.finish(beta, cfg.FOCAL_P, True)
        lab, conf, _ = ops.pseudo_labels(refined, ignore, cfg.RUN_CONF_UPPER, cfg.RUN_CONF_LOWER, disc)
        out[dev_thr] = (lab.cpu(), conf.cpu(), disc.cpu())
        assert isinstance(net._class_vectors, ops.DeviceClassVectors if dev_thr else ops.HostClassVectors)
    assert float((out[True][0] != out[False][0]).float().mean()) < 1e-5 and int((out[True][0] != 255).sum()) > 0
    assert float((out[True][2] - out[False][2]).abs().max()) <= 2.0 ** -23
