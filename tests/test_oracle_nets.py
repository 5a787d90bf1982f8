"""Pins oracle/nets_ref.py + oracle/step_ref.py against reference-captured vectors (CPU)."""
import numpy as np
import pytest
import torch

from oracle import nets_ref as N
from oracle.step_ref import SacOracle, SgdOracle, sac_train_iteration, view_slice_index, gather_index
from conftest import rel_err

T = torch.from_numpy


def sampled(t, n=64):
    flat = t.detach().reshape(-1)
    m = min(n, flat.numel())
    idx = (torch.arange(m, dtype=torch.int64) * (flat.numel() - 1)) // max(m - 1, 1)
    return flat[idx]


def _grad_keys(g, tag):
    return [k[len(tag) + 3:] for k in g.files if k.startswith(tag + "_g_")]


@pytest.mark.parametrize("tag,bn_train", [("eval", False), ("train", True)])
def test_g2_resnet101_logits_loss_grads(golden, tag, bn_train):
    g = golden("g2_resnet101")
    sd = N.resnet101_state(seed=2, randomize_bn=True, he_init=True, residual_gain=0.25, aspp_gain=0.2)
    for k in N.trainable_keys(sd):
        sd[k].requires_grad_(True)
    losses, outs = N.segnet_forward("deeplabv2_resnet101", sd, T(g[tag + "_x"]), T(g[tag + "_y"]), bn_train=bn_train)
    losses["loss_ce"].sum().backward()
    assert rel_err(outs["logits"], g[tag + "_logits"]) < 1e-4
    assert rel_err(sampled(outs["logits_up"], 512), g[tag + "_logits_up_s"]) < 1e-4
    assert rel_err(losses["loss_ce"], g[tag + "_loss"]) < 1e-5
    for k in _grad_keys(g, tag):
        assert rel_err(sd[k].grad.norm(), g[tag + "_gn_" + k]) < 1e-3, k
        assert float((sampled(sd[k].grad) - T(g[tag + "_g_" + k])).abs().max()) < 1e-3 * float(g[tag + "_gn_" + k]) + 1e-7, k
    if bn_train:
        assert rel_err(sd["model.bn1.running_mean"], g["train_rm_bn1"]) < 1e-5
        assert rel_err(sd["model.layer3.4.bn2.running_var"], g["train_rv_l3"]) < 1e-5
        assert int(sd["model.bn1.num_batches_tracked"]) == int(g["train_nbt"])


def test_g10_vgg16_deeplab_cfg1(golden):
    g = golden("g10_vgg16_deeplab")
    sd = N.deeplab_vgg16_state(seed=10, randomize_bn=True)
    assert sorted(sd.keys()) == list(g["keys"])
    for k in N.trainable_keys(sd):
        sd[k].requires_grad_(True)
    gen = torch.Generator().manual_seed(10)
    x = torch.randn(1, 3, 321, 321, generator=gen)
    y = torch.randint(0, 19, (1, 321, 321), generator=gen)
    losses, outs = N.segnet_forward("deeplabv2_vgg16_bn", sd, x, y)
    losses["loss_ce"].sum().backward()
    assert rel_err(outs["logits"], g["logits"]) < 1e-4
    assert rel_err(losses["loss_ce"], g["loss"]) < 1e-5
    assert rel_err(sampled(sd["features.0.weight"].grad), g["g_first"]) < 1e-4
    assert rel_err(sampled(sd["features.42.weight"].grad), g["g_fc6"]) < 1e-4
    assert rel_err(sd["classifier.conv2d_list.2.bias"].grad, g["g_cls_bias"]) < 1e-3


def test_g10_fcn8s(golden):
    g = golden("g10_fcn8s")
    sd = N.fcn8s_vgg16_state(seed=12, randomize_bn=True)
    assert sorted(sd.keys()) == list(g["keys"])
    for k in N.trainable_keys(sd):
        sd[k].requires_grad_(True)
    losses, outs = N.segnet_forward("fcn_vgg16_bn", sd, T(g["x"]), T(g["y"]))
    assert set(outs) == {"logits_up"}
    losses["loss_ce"].sum().backward()
    assert rel_err(outs["logits_up"], g["logits_up"]) < 1e-4
    assert rel_err(losses["loss_ce"], g["loss"]) < 1e-5
    assert rel_err(sampled(sd["vgg_head.0.weight"].grad), g["g_head0"]) < 1e-4
    assert rel_err(sd["score_pool3.weight"].grad.reshape(-1)[:64], g["g_sp3"]) < 1e-4
    assert rel_err(sampled(sd["block1.0.weight"].grad), g["g_first"]) < 1e-4


def test_keys_and_param_groups(golden):
    g = golden("keys_sac_resnet101")
    sd = N.resnet101_state(seed=0)
    names = ["running_conf", "slow_init"] + ["backbone." + k for k in sd] + ["slow_net." + k for k in sd]
    assert sorted(names) == list(g["names"])
    shapes = dict(zip(g["names"], g["shapes"]))
    for k, v in sd.items():
        assert "x".join(str(s) for s in v.shape) == shapes["backbone." + k], k
    m = SacOracle(sd)
    groups = m.param_groups()
    assert [len(x["keys"]) for x in groups] == [208, 104, 4, 4]
    for i, gr in enumerate(groups):
        assert ["backbone." + k for k in gr["keys"]] == list(g["group%d" % i])
        assert gr["lr"] == pytest.approx(float(g["group_lr"][i]) * m.cfg["LR"])
        assert gr["wd"] == pytest.approx(float(g["group_wd"][i]) * m.cfg["WEIGHT_DECAY"])


def test_g8_two_sac_training_steps(golden):
    g = golden("g8_two_steps")
    Tn = int(g["T"])
    m = SacOracle(N.resnet101_state(seed=8, randomize_bn=True, he_init=True, residual_gain=0.25, aspp_gain=0.2))
    opt = SgdOracle(m)
    aff, inv = T(g["affine"]), T(g["affine_inv"])
    for it in range(2):
        src = (T(g["it%d_xs" % it]), T(g["it%d_ys" % it]))
        tgt = (T(g["it%d_f1" % it]), T(g["it%d_gt" % it]).clone(), T(g["it%d_f2" % it]), aff, inv)
        ls, lt, outs = sac_train_iteration(m, opt, src, tgt, Tn, update_teacher=(it % 100 == 0))
        assert ls["loss_ce"] == pytest.approx(float(g["it%d_src_loss" % it].item()), rel=1e-4)
        for k in ("loss_ce", "self_ce", "teacher_diff"):
            assert lt[k] == pytest.approx(float(g["it%d_%s" % (it, k)].item()), rel=2e-3, abs=1e-6), (it, k)
        lab = outs["teacher_labels"].to(torch.uint8)
        mism = (lab != T(g["it%d_labels" % it])).float().mean()
        assert mism < 1e-3, (it, float(mism))          # float pipeline upstream: a few threshold flips allowed
        assert rel_err(m.running_conf, g["it%d_chi" % it]) < 1e-4
        for k in [k[len("it0_p_"):] for k in g.files if k.startswith("it0_p_")]:
            assert rel_err(sampled(m.student[k]), g["it%d_p_%s" % (it, k)]) < 1e-4, (it, k)
            assert rel_err(m.student[k].norm(), g["it%d_pn_%s" % (it, k)]) < 1e-5, (it, k)


def test_g11_view_sharding_index_tables():
    # train.py:199-209 comment table and sac.py:211-212 ("0,1,2,3 -> 0,0,2,2")
    assert view_slice_index(1, 0, 2, 4) is None
    assert view_slice_index(8, 3, 16, 4) is None
    assert [view_slice_index(4, r, 2, 4) for r in range(4)] == [(0, 0, 2), (0, 2, 4), (1, 0, 2), (1, 2, 4)]
    assert [view_slice_index(8, r, 2, 4) for r in range(8)] == [(r // 4, r % 4, r % 4 + 1) for r in range(8)]
    assert [gather_index(4, r, 2, 4) for r in range(4)] == [(0, 2), (0, 2), (2, 4), (2, 4)]
    assert [gather_index(8, r, 1, 4)[0] for r in range(8)] == [0, 0, 0, 0, 4, 4, 4, 4]
    assert gather_index(8, 0, 8, 4) is None


def test_aten_conv_bn_pool_equal_the_textbook_definitions():
    """The oracle delegates conv / BN / max-pool arithmetic to ATen (the reference's own provider).  Pin ATen itself
    against explicit numpy loops on a small dilated, strided, padded case so that nothing in the chain is taken on trust."""
    import numpy as np
    import torch.nn.functional as F
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 3, 7, 9)).astype(np.float64)
    w = rng.standard_normal((4, 3, 3, 3)).astype(np.float64)
    stride, pad, dil = 2, 2, 2
    OH = (7 + 2 * pad - dil * 2 - 1) // stride + 1
    OW = (9 + 2 * pad - dil * 2 - 1) // stride + 1
    ref = np.zeros((2, 4, OH, OW))
    for n in range(2):
        for co in range(4):
            for oh in range(OH):
                for ow in range(OW):
                    acc = 0.0
                    for ci in range(3):
                        for kh in range(3):
                            for kw in range(3):
                                ih, iw = oh * stride - pad + kh * dil, ow * stride - pad + kw * dil
                                if 0 <= ih < 7 and 0 <= iw < 9:
                                    acc += x[n, ci, ih, iw] * w[co, ci, kh, kw]
                    ref[n, co, oh, ow] = acc
    got = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), None, stride, pad, dil).numpy()
    assert np.abs(got - ref).max() < 1e-12
    # eval-mode BN and the ceil-mode 3x3/2 max-pool of the stem (deeplabv2.py:126)
    g, b, m, v = (rng.standard_normal(3) for _ in range(4))
    v = np.abs(v) + 0.5
    bn = F.batch_norm(torch.from_numpy(x), torch.from_numpy(m), torch.from_numpy(v), torch.from_numpy(g), torch.from_numpy(b), False, 0.0, 1e-5)
    want = (x - m[None, :, None, None]) / np.sqrt(v[None, :, None, None] + 1e-5) * g[None, :, None, None] + b[None, :, None, None]
    assert np.abs(bn.numpy() - want).max() < 1e-12
    mp = F.max_pool2d(torch.from_numpy(x), 3, 2, 1, ceil_mode=True).numpy()
    PH, PW = mp.shape[2:]
    for oh in range(PH):
        for ow in range(PW):
            win = x[:, :, max(oh * 2 - 1, 0):min(oh * 2 + 2, 7), max(ow * 2 - 1, 0):min(ow * 2 + 2, 9)]
            assert np.array_equal(mp[:, :, oh, ow], win.max(axis=(2, 3)))
