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
