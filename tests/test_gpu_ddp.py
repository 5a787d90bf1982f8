"""DistributedDataParallel over RCCL with one rank (all a 1-GPU box allows): the fused engine's single
autograd.Function must cooperate with DDP's reducer hooks, buffer broadcast and two backward passes per step
(train.py:104,133,232); results must equal the un-wrapped module."""
import os
from types import SimpleNamespace as NS

import pytest
import torch
import torch.distributed as dist
import torch.nn as nn

from oracle import nets_ref as N
from oracle.step_ref import DEFAULT_CFG

pytestmark = pytest.mark.gpu
CRIT = nn.CrossEntropyLoss(ignore_index=255, reduction="none")


def _run(wrap):
    import models
    import driver
    cfg = NS(**dict(DEFAULT_CFG, INIT_MODEL="", OPT_NESTEROV=False))
    net = models.get_model(cfg, 0, num_classes=19, criterion=CRIT)
    net.backbone.load_state_dict(N.resnet101_state(seed=3, randomize_bn=True, he_init=True, residual_gain=0.25, aspp_gain=0.2), strict=True)
    net.cuda().train()
    optim = driver.make_optimizer(net, cfg)
    step_net = nn.parallel.DistributedDataParallel(net, device_ids=[0]) if wrap else net
    src, tgt = driver.synthetic_batches(2, 1, 2, (33, 49), "cuda", seed=5)
    out = []
    for it in range(2):
        t = (tgt[0], tgt[1].clone(), tgt[2], tgt[3], tgt[4])
        ls, lt, _ = driver.sac_train_iteration(step_net, optim, src, t, 2, it == 0, cfg.LR_TARGET)
        out.append((float(ls["loss_ce"]), float(lt["self_ce"]), float(lt["teacher_diff"])))
    w = net.backbone.state_dict()["model.layer3.5.conv2.weight"].clone()
    return out, w


def test_ddp_single_rank_matches_plain_module():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    plain, w0 = _run(False)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        wrapped, w1 = _run(True)
    finally:
        dist.destroy_process_group()
    for a, b in zip(plain, wrapped):
        assert a == pytest.approx(b, rel=1e-5, abs=1e-7)
    assert torch.allclose(w0, w1, rtol=1e-5, atol=1e-8)
