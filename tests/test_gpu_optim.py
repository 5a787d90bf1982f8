"""Fused multi-tensor SGD (dasac_sgd_step) and the raw-pointer parameter updates' interaction with the engine caches.
Reference semantics: torch.optim.SGD(momentum, no nesterov) over four groups (base_trainer.py:63-66) and the teacher
EMA (models/sac.py:83-102).  Tolerance: 1e-6 rel (same op order as ATen, fp32)."""
import copy

import pytest
import torch
import torch.nn as nn

from oracle import nets_ref as N
from conftest import rel_err
from test_gpu_models import CRIT, model_cfg

pytestmark = pytest.mark.gpu


def test_fused_sgd_matches_torch_sgd():
    from dasac_hip.optim import FusedSGD
    g = torch.Generator().manual_seed(3)
    shapes = [(64, 3, 7, 7), (64,), (5000,), (19, 2048, 3, 3), (1,), (4097,)]
    pa = [nn.Parameter(torch.randn(s, generator=g).cuda()) for s in shapes]
    pb = [nn.Parameter(p.detach().clone()) for p in pa]
    groups = lambda ps: [{"params": ps[:2], "lr": 2.5e-4, "weight_decay": 5e-4}, {"params": ps[2:4], "lr": 5e-3, "weight_decay": 0.0},
                         {"params": ps[4:], "lr": 2.5e-3, "weight_decay": 5e-4}]
    oa, ob = FusedSGD(groups(pa), momentum=0.9), torch.optim.SGD(groups(pb), momentum=0.9)
    for it in range(4):
        for i, (a, b) in enumerate(zip(pa, pb)):
            if it == 0 and i == 5:
                continue                                   # a parameter that gets its first gradient one step later
            gr = torch.randn(a.shape, generator=g).cuda()
            a.grad, b.grad = gr.clone(), gr.clone()
        if it == 2:
            for o in (oa, ob):
                o.param_groups[1]["lr"] = 1e-3               # schedules poke param_groups
        v0 = pa[0]._version
        oa.step()
        ob.step()
        assert pa[0]._version > v0                           # engine caches key on this
        for a, b in zip(pa, pb):
            assert rel_err(a, b) < 1e-6
            if b in ob.state and "momentum_buffer" in ob.state[b]:
                assert rel_err(oa.state[a]["momentum_buffer"], ob.state[b]["momentum_buffer"]) < 1e-6
    sd = oa.state_dict()
    assert set(sd["state"][0]) == {"momentum_buffer"} and sd["param_groups"][0]["momentum"] == 0.9
    ob.load_state_dict(sd)                                   # layout-compatible with torch.optim.SGD


def test_stashed_gradients_sum_inside_the_update_bit_for_bit():
    """driver.sac_train_iteration sets the source-pass gradients aside and lets the update kernel add the target-pass
    ones (instead of 320 `add_` launches): same parameters, bit for bit, as accumulating into .grad first; a parameter
    that only one of the two passes touched is handled; zero_grad drops the stash."""
    from dasac_hip.optim import FusedSGD
    g = torch.Generator().manual_seed(5)
    shapes = [(64, 3, 7, 7), (64,), (5000,), (4097,)]
    pa = [nn.Parameter(torch.randn(s, generator=g).cuda()) for s in shapes]
    pb = [nn.Parameter(p.detach().clone()) for p in pa]
    groups = lambda ps: [{"params": ps[:2], "lr": 2.5e-4, "weight_decay": 5e-4}, {"params": ps[2:], "lr": 5e-3, "weight_decay": 0.0}]
    oa, ob = FusedSGD(groups(pa), momentum=0.9), FusedSGD(groups(pb), momentum=0.9)
    for it in range(3):
        g1 = [torch.randn(a.shape, generator=g).cuda() for a in pa]
        g2 = [torch.randn(a.shape, generator=g).cuda() for a in pa]
        oa.zero_grad()                                       # as driver.sac_train_iteration does before the source backward
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i != 3:
                a.grad = g1[i].clone()
        oa.stash_grads()
        assert all(a.grad is None for a in pa)
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i != 2:
                a.grad = g2[i].clone()                       # parameter 2: source pass only; parameter 3: target pass only
            b.grad = g1[i].clone() if i != 3 else None
            if i != 2:
                if b.grad is None:
                    b.grad = g2[i].clone()
                else:
                    b.grad += g2[i]                          # what AccumulateGrad does
        oa.step()
        ob.step()
        for a, b in zip(pa, pb):
            assert torch.equal(a, b) and torch.equal(oa.state[a]["momentum_buffer"], ob.state[b]["momentum_buffer"])
    pa[0].grad = torch.ones_like(pa[0])
    oa.stash_grads()
    oa.zero_grad()
    before = pa[0].detach().clone()
    oa.step()                                                # nothing stashed, no gradient: no update
    assert torch.equal(pa[0], before)


def test_training_with_fused_sgd_equals_torch_sgd():
    """Three SAC-free training steps of the RN101 model: the fused optimiser and torch.optim.SGD give the same loss
    curve, i.e. the packed-weight / BN-fold caches see every raw-pointer update."""
    import models
    import driver
    cfg = model_cfg()
    sd = N.resnet101_state(seed=4, randomize_bn=True, he_init=True, residual_gain=0.25, aspp_gain=0.2)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 3, 41, 49, generator=g).cuda()
    y = torch.randint(0, 19, (2, 41, 49), generator=g).cuda()
    curves = []
    for fused in (True, False):
        net = models.DeepLabV2_ResNet101(num_classes=19, criterion=CRIT, freeze_bn=True)
        net.load_state_dict(sd, strict=True)
        net.cuda().train()
        cfg2 = copy.copy(cfg)
        cfg2.LR = 1e-3                                       # 4x the reference LR: the updates matter within 3 steps
        opt = driver.make_optimizer(net, cfg2, fused=fused)
        losses = []
        for _ in range(3):
            l, _ = net(x, y)
            opt.zero_grad()
            l["loss_ce"].mean().backward()
            opt.step()
            losses.append(float(l["loss_ce"].mean()))
        curves.append(losses)
    assert curves[0][0] != curves[0][2]
    for a, b in zip(*curves):
        assert a == pytest.approx(b, rel=1e-5)


def test_teacher_forward_sees_the_ema_update():
    """sac.py:83-102 then a teacher forward: the EMA kernel writes the slow net in place, the teacher engine must
    re-pack its weights."""
    import models
    cfg = model_cfg(NET_MOMENTUM=0.5)
    net = models.get_model(cfg, 0, num_classes=19, criterion=CRIT)
    net.backbone.load_state_dict(N.resnet101_state(seed=6, randomize_bn=True, he_init=True, residual_gain=0.25, aspp_gain=0.2), strict=True)
    net.cuda().eval()
    x = torch.randn(1, 3, 33, 41, device="cuda")
    with torch.no_grad():
        net._momentum_update(True)                           # first call: teacher <- student
        t0, _ = net(x, teacher=True)
        for p in net.backbone.parameters():
            p.mul_(1.05)
        d = net._momentum_update(True)                       # EMA step through raw pointers
        assert float(d) > 0
        t1, _ = net(x, teacher=True)
        fresh = models.get_model(cfg, 0, num_classes=19, criterion=CRIT)
        fresh.backbone.load_state_dict(net.slow_net.state_dict(), strict=True)
        fresh.cuda().eval()
        t2, _ = fresh(x, teacher=False)
    assert rel_err(t1, t0) > 1e-3
    assert rel_err(t1, t2) < 1e-6
