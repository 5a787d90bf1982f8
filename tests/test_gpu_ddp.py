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


# ---------------------------------------------------------------------------------------------------------------
# two ranks (both on the one GPU of the box, gloo transport): gradient averaging over ranks through DDP's reducer
# on top of the fused engine, against a one-process emulation (per-rank gradients averaged by hand)
# ---------------------------------------------------------------------------------------------------------------
_PROBE = ("model.conv1.weight", "model.layer2.1.bn2.weight", "model.layer3.5.conv2.weight", "model.layer5.conv2d_list.1.bias")


def _build():
    import models
    cfg = NS(**dict(DEFAULT_CFG, INIT_MODEL="", OPT_NESTEROV=False))
    net = models.get_model(cfg, 0, num_classes=19, criterion=CRIT)
    net.backbone.load_state_dict(N.resnet101_state(seed=3, randomize_bn=True, he_init=True, residual_gain=0.25, aspp_gain=0.2), strict=True)
    net.cuda().train()
    return cfg, net


def _two_backward_passes(step_net, src, tgt, lr_target):
    ls, _ = step_net(*src)
    for p in step_net.parameters():
        p.grad = None
    ls["loss_ce"].mean().backward()
    lt, _ = step_net(tgt[0], tgt[1].clone(), tgt[2], tgt[3], tgt[4], use_teacher=True, update_teacher=True, T=2)
    (lr_target * lt["self_ce"].mean()).backward()
    return float(ls["loss_ce"]), float(lt["self_ce"])


def _ddp_rank(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "da-sac_amd"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from conftest import init_ranks
    dev = init_ranks(rank, world)
    import driver
    cfg, net = _build()
    optim = driver.make_optimizer(net, cfg)
    ddp = nn.parallel.DistributedDataParallel(net, device_ids=[dev])
    src, tgt = driver.synthetic_batches(2, 1, 2, (33, 49), "cuda", seed=50 + rank)
    losses = _two_backward_passes(ddp, src, tgt, cfg.LR_TARGET)
    optim.step()
    torch.cuda.synchronize()
    sd = net.backbone.state_dict()
    q.put((rank, losses, {k: sd[k].detach().cpu().numpy() for k in _PROBE}))      # by value (no shared-memory handles)
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_two_ranks_average_gradients_like_a_manual_mean():
    import socket
    import torch.multiprocessing as mp
    import driver
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=240) for _ in procs), key=lambda t: t[0])
    got = [(r, l, {k: torch.from_numpy(v) for k, v in w.items()}) for r, l, w in got]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # both ranks end up with the same parameters
    for k in _PROBE:
        assert torch.equal(got[0][2][k], got[1][2][k]), k
    # one-process emulation: same start, per-rank data, gradients averaged by hand, one optimiser step
    cfg, net = _build()
    start = {k: v.clone() for k, v in net.state_dict().items()}
    grads, losses = [], []
    for rank in range(2):
        net.load_state_dict(start)
        src, tgt = driver.synthetic_batches(2, 1, 2, (33, 49), "cuda", seed=50 + rank)
        losses.append(_two_backward_passes(net, src, tgt, cfg.LR_TARGET))
        grads.append([p.grad.clone() for p in net.parameters() if p.requires_grad])
    net.load_state_dict(start)
    optim = driver.make_optimizer(net, cfg)
    for p, g0, g1 in zip([p for p in net.parameters() if p.requires_grad], *grads):
        p.grad = (g0 + g1) / 2
    optim.step()
    sd = net.backbone.state_dict()
    for rank in range(2):
        assert got[rank][1] == pytest.approx(losses[rank], rel=1e-5)
    for k in _PROBE:
        ref, out = sd[k].cpu(), got[0][2][k]
        assert float((ref - out).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-9, k
        assert not torch.equal(out, start["backbone." + k].cpu()), k          # the step did move it


def test_ddp_buffer_broadcast_does_not_repack_frozen_networks():
    """DistributedDataParallel re-sends every buffer before each forward; for frozen BN that would bump the running
    statistics' version counters and make the engine re-fold / re-pack both networks on every forward (+17 ms per cfg-3
    step before SAC exempted them).  Two forwards under DDP: the second must pack nothing."""
    from dasac_hip import ops
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        cfg, net = _build()
        ddp = nn.parallel.DistributedDataParallel(net, device_ids=[0])
        import driver
        src, tgt = driver.synthetic_batches(2, 1, 2, (33, 49), "cuda", seed=3)
        calls = []
        real, real_all = ops.conv_pack, ops.refresh_network           # per-layer packs and the whole-network refresh
        ops.conv_pack = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        ops.refresh_network = lambda *a, **k: (calls.append(1), real_all(*a, **k))[1]
        try:
            with torch.no_grad():
                ddp(tgt[0], tgt[1].clone(), tgt[2], tgt[3], tgt[4], use_teacher=True, update_teacher=True, T=2)
                first = len(calls)
                ddp(tgt[0], tgt[1].clone(), tgt[2], tgt[3], tgt[4], use_teacher=True, update_teacher=False, T=2)
        finally:
            ops.conv_pack, ops.refresh_network = real, real_all
        assert first > 0 and len(calls) == first, (first, len(calls))
    finally:
        dist.destroy_process_group()
