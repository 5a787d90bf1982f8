"""Pins the CPU oracle (oracle/) against vectors captured from the reference itself
(tests/golden/make_goldens.py, run in the build container).  CPU only."""
import numpy as np
import torch

from oracle import head_ref as H
from conftest import rel_err

T = torch.from_numpy


def test_g3_bilinear_align_corners(golden):
    g = golden("g3_bilinear")
    assert rel_err(H.upsample_bilinear_ac(T(g["x"]), 65, 97), g["y"]) < 2e-6
    assert rel_err(H.upsample_bilinear_ac(T(g["row"]), 769, 9), g["yrow"]) < 2e-6
    assert rel_err(H.upsample_bilinear_ac(T(g["x2"]), 8, 12), g["y2"]) < 2e-6


def test_g9_view_affines(golden):
    g = golden("g9_affine")
    params = [tuple(r) for r in g["params"]]
    th, inv = H.view_affines(params, int(g["crop"][0]), int(g["crop"][1]))
    assert torch.allclose(th, T(g["affine"]), atol=1e-7)
    assert torch.allclose(inv, T(g["affine_inv"]), atol=1e-6)


def test_g4_warps(golden):
    g = golden("g4_refine")
    frames, aff, inv = T(g["frames"]), T(g["affine"]), T(g["affine_inv"])
    assert rel_err(H.warp_affine(frames, aff), g["warp_frames"]) < 1e-5
    assert rel_err(H.warp_coverage(inv, frames.shape[2], frames.shape[3]), g["warp_ones_inv"]) < 1e-5


def _refine(g, kind):
    return H.refine(T(g["frames"]), T(g["logits"]), int(g["T"]), T(g["affine"]), T(g["affine_inv"]),
                    T(g["ignore"]), T(g["chi_in"]), beta=1e-3, stat_momentum=0.99, training=True,
                    pool=True, pool_kind=kind)


def test_g4_refine_avg_pool(golden):
    g = golden("g4_refine")
    refined, chi, diags = _refine(g, "avg_pool")
    assert rel_err(refined, g["avg_pool_refined"]) < 2e-5
    assert rel_err(chi, g["avg_pool_chi_out"]) < 1e-6
    assert rel_err(diags["teacher_aligned"], g["avg_pool_teacher_aligned"]) < 1e-5
    assert rel_err(diags["frames_aligned"], g["avg_pool_frames_aligned"]) < 1e-5


def test_g4_refine_minentropy_pool(golden):
    g = golden("g4_refine")
    refined, chi, _ = _refine(g, "minentropy_pool")
    ref = T(g["minentropy_pool_refined"])
    # argmin over views is a discontinuity: allow a handful of pixels to pick another view
    bad = ((refined - ref).abs().amax(1) > 1e-4).float().mean()
    assert bad < 2e-3
    assert rel_err(chi, g["minentropy_pool_chi_out"]) < 1e-6


def test_g5_pseudo_labels_bit_exact(golden):
    g = golden("g5_pseudo_labels")
    probs, ignore, chi = T(g["probs"]), T(g["ignore"]), T(g["chi"])
    disc = H.threshold_discount(chi, float(g["beta"]))
    assert torch.equal(disc, T(g["discount"]))
    for tag, d in (("disc", disc), ("nodisc", None)):
        lab, conf, idx = H.pseudo_labels(probs, ignore, float(g["upper"]), float(g["lower"]), d)
        assert torch.equal(lab, T(g["labels_" + tag]))
        assert torch.equal(conf, T(g["conf_" + tag]))
        assert torch.equal(idx, T(g["idx_" + tag]))
    lab = T(g["labels_nodisc"])
    assert lab[0, 2, 3] == 255 and lab[1, 5, 5] in (4, 255) and int(T(g["idx_nodisc"])[1, 0, 5, 5]) == 4
    assert lab[2, 0, 1] == 255                       # m == thr is NOT labelled (strict >)


def test_g6_losses(golden):
    g = golden("g6_losses")
    y, conf, chi = T(g["y"]), T(g["conf"]), T(g["chi"])
    for name, fn in (("conf", lambda x: H.focal_ce_conf(x, y, conf, chi, 3)),
                     ("plain", lambda x: H.focal_ce(x, y, chi, 3))):
        x = T(g["logits"]).clone().requires_grad_(True)
        loss, per_class = fn(x)
        (grad,) = torch.autograd.grad(loss, x)
        assert rel_err(loss, g[name + "_loss"]) < 1e-5
        assert rel_err(per_class, g[name + "_per_class"]) < 1e-5
        assert rel_err(grad, g[name + "_grad"]) < 1e-5
    x = T(g["logits"]).clone().requires_grad_(True)
    loss = H.ce_mean_all_pixels(x, y)
    (grad,) = torch.autograd.grad(loss.sum(), x)
    assert rel_err(loss, g["ce_loss"]) < 1e-6 and rel_err(grad, g["ce_grad"]) < 1e-5


def test_g7_running_conf_sequence(golden):
    g = golden("g7_state")
    chi = torch.zeros(19)
    for it in range(4):
        if it == 1:
            chi = torch.full((19,), 1e-3)
        chi = H.update_running_conf(chi, T(g["probs"][it]), 1e-3, 0.99)
        assert rel_err(chi, g["chi_seq"][it]) < 1e-6


def test_g7_momentum_sequence(golden):
    from oracle import nets_ref as N
    from oracle.step_ref import SacOracle
    g = golden("g7_state")
    m = SacOracle(N.resnet101_state(seed=11, randomize_bn=True))
    d = [m.momentum_update(True)]
    assert torch.equal(m.running_conf, torch.full((19,), 1e-3)) and float(m.slow_init[0]) == 1.0
    with torch.no_grad():
        # same perturbation as the generator: parameters() order == trainable key order
        for i, k in enumerate(N.trainable_keys(m.student)):
            m.student[k].add_(0.01 * ((i % 7) - 3))
        m.student["model.bn1.running_mean"].add_(0.5)
    d += [m.momentum_update(False), m.momentum_update(True), m.momentum_update(False)]
    assert rel_err(torch.cat(d), g["diffs"]) < 1e-5
    assert rel_err(m.teacher["model.conv1.weight"], g["slow_conv1"]) < 1e-6
    assert rel_err(m.teacher["model.bn1.running_mean"], g["slow_bn1_mean"]) < 1e-6
    assert rel_err(m.teacher["model.layer3.5.conv2.weight"][:4, :4], g["slow_l3_w"]) < 1e-6
