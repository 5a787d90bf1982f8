"""Train-mode (batch statistics) BatchNorm path = baseline / AdaBN mode (cfg-2): reference golden g2 'train'
case and the oracle's baseline iteration (train.py:274-289)."""
from types import SimpleNamespace as NS

import pytest
import torch
import torch.nn as nn

from oracle import nets_ref as N
from oracle.step_ref import DEFAULT_CFG, SacOracle, SgdOracle, baseline_train_iteration
from conftest import rel_err

pytestmark = pytest.mark.gpu
T = torch.from_numpy
CRIT = nn.CrossEntropyLoss(ignore_index=255, reduction="none")


def sampled(t, n=64):
    flat = t.detach().reshape(-1)
    m = min(n, flat.numel())
    idx = (torch.arange(m, dtype=torch.int64, device=flat.device) * (flat.numel() - 1)) // max(m - 1, 1)
    return flat[idx]


def test_resnet101_train_bn_golden_g2(golden):
    import models
    g = golden("g2_resnet101")
    net = models.DeepLabV2_ResNet101(num_classes=19, criterion=CRIT, freeze_bn=False)
    net.load_state_dict(N.resnet101_state(seed=2, randomize_bn=True, he_init=True, residual_gain=0.25, aspp_gain=0.2), strict=True)
    net.cuda().train()
    assert all(m.training for m in net.modules() if isinstance(m, nn.SyncBatchNorm))
    losses, outs = net(T(g["train_x"]).cuda(), T(g["train_y"]).cuda())
    losses["loss_ce"].mean().backward()
    assert rel_err(outs["logits"], g["train_logits"]) < 1e-4
    assert rel_err(losses["loss_ce"], g["train_loss"]) < 1e-5
    st = net.state_dict()
    assert rel_err(st["model.bn1.running_mean"], g["train_rm_bn1"]) < 1e-5
    assert rel_err(st["model.layer3.4.bn2.running_var"], g["train_rv_l3"]) < 1e-5
    assert int(st["model.bn1.num_batches_tracked"]) == int(g["train_nbt"])
    named = dict(net.named_parameters())
    errs = []
    for k in [k[len("train_g_"):] for k in g.files if k.startswith("train_g_")]:
        gn = float(g["train_gn_" + k])
        errs.append((abs(float(named[k].grad.norm()) - gn) / gn, k))
        assert float((sampled(named[k].grad).cpu() - T(g["train_g_" + k])).abs().max()) < 2e-2 * gn + 1e-7, k
    errs.sort(reverse=True)
    assert errs[len(errs) // 2][0] < 1e-3, errs[:4]        # a borderline ReLU may move single tensors


def test_baseline_adabn_iteration_vs_oracle():
    """Source fwd/bwd/SGD step + no-grad train-mode target forward (running-stat re-estimation)."""
    import models
    import driver
    cfg = NS(**dict(DEFAULT_CFG, INIT_MODEL="", OPT_NESTEROV=False, BASELINE=True))
    sd = N.resnet101_state(seed=4, randomize_bn=True, he_init=True, residual_gain=0.25, aspp_gain=0.2)
    g = torch.Generator().manual_seed(4)
    xs, ys = torch.randn(2, 3, 33, 41, generator=g), torch.randint(0, 19, (2, 33, 41), generator=g)
    xt = torch.randn(2, 3, 33, 41, generator=g)
    ref = SacOracle(sd, cfg=dict(BASELINE=True))
    l_ref = baseline_train_iteration(ref, SgdOracle(ref), (xs, ys), xt)
    net = models.get_model(cfg, 0, num_classes=19, criterion=CRIT)
    assert isinstance(net, models.SAC_Baseline)
    net.backbone.load_state_dict(sd, strict=True)
    net.cuda().train()
    optim = driver.make_optimizer(net, cfg)
    l = driver.baseline_train_iteration(net, optim, (xs.cuda(), ys.cuda()), xt.cuda())
    assert float(l["loss_ce"]) == pytest.approx(l_ref["loss_ce"], rel=1e-4)
    st = net.backbone.state_dict()
    for k in ("model.bn1.running_mean", "model.layer2.1.bn3.running_var", "model.layer4.2.bn1.running_mean"):
        assert rel_err(st[k], ref.student[k]) < 1e-4, k
    assert int(st["model.bn1.num_batches_tracked"]) == 2
    bad = sorted(((rel_err(st[k], ref.student[k].detach()), k) for k in N.trainable_keys(sd)), reverse=True)
    assert bad[len(bad) // 2][0] < 1e-5 and bad[0][0] < 1e-2, bad[:3]


def test_eval_forward_after_train_mode_steps_uses_the_new_running_stats():
    """AdaBN (train.py:281-289): train-mode forwards move the running statistics through the HIP kernels; the eval-mode
    fold (cached per BN layer) has to follow."""
    import models
    net = models.DeepLabV2_ResNet101(num_classes=19, criterion=CRIT, freeze_bn=False)
    net.load_state_dict(N.resnet101_state(seed=9, randomize_bn=True, he_init=True, residual_gain=0.25, aspp_gain=0.2), strict=True)
    net.cuda()
    x = torch.randn(2, 3, 33, 41, device="cuda")
    y = torch.zeros(2, 33, 41, dtype=torch.int64, device="cuda")
    with torch.no_grad():
        net.eval()
        e0, _ = net(x)
        net.train()
        for _ in range(2):
            net(x * 3 + 1, y)
        net.eval()
        e1, _ = net(x)
        fresh = models.DeepLabV2_ResNet101(num_classes=19, criterion=CRIT, freeze_bn=False)
        fresh.load_state_dict(net.state_dict(), strict=True)
        fresh.cuda().eval()
        e2, _ = fresh(x)
    assert rel_err(e1, e0) > 1e-3
    assert rel_err(e1, e2) < 1e-6
