"""world_size-2 gloo checks (CPU) of the multi-rank host logic: the view-gather of SAC._gather
(models/sac.py:198-216 in the reference) and the reference's batch-slicing index math (train.py:186-209)."""
import os
import socket
from types import SimpleNamespace as NS

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from oracle.step_ref import DEFAULT_CFG, gather_index, view_slice_index


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, T, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "da-sac_amd"))
    import models
    cfg = NS(**dict(DEFAULT_CFG, INIT_MODEL=""))
    torch.manual_seed(0)
    net = models.get_model(cfg, rank, num_classes=19, criterion=nn.CrossEntropyLoss(ignore_index=255, reduction="none"))
    assert net.world_size == world and net.rank == rank
    # every rank holds B = 1 view of a T = 2 group: each must end up with both views of ITS group
    mine = torch.full((1, 3, 2, 2), float(rank))
    got = net._gather(mine, T)
    lo, hi = gather_index(world, rank, 1, T)
    want = torch.cat([torch.full((1, 3, 2, 2), float(r)) for r in range(lo, hi)], 0)
    ok = torch.equal(got, want)
    # whole groups on a rank: no communication
    full = torch.randn(T, 3, 2, 2)
    ok = ok and net._gather(full, T) is full
    q.put((rank, bool(ok), list(got[:, 0, 0, 0].tolist())))
    dist.barrier()
    dist.destroy_process_group()


def test_view_gather_across_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 2, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, True, [0.0, 1.0]), (1, True, [0.0, 1.0])]


def test_rank_slicing_tables_match_reference_comments():
    # train.py:199-209 and sac.py:211-212 ("0,1,2,3 -> 0,0,2,2")
    assert [gather_index(4, r, 2, 4)[0] for r in range(4)] == [0, 0, 2, 2]
    assert [view_slice_index(8, r, 2, 4) for r in range(8)] == [(r // 4, r % 4, r % 4 + 1) for r in range(8)]
    with pytest.raises(AssertionError):
        view_slice_index(3, 0, 2, 4)


# ---------------------------------------------------------------------------------------------------------------
# G11: the reference's own Trainer._prep_batch / SAC._gather run unbound in `world` gloo processes
# (tests/golden/make_goldens.py: g11_index_tables) -> which loaded (rank, image, view) every rank ends up with
# ---------------------------------------------------------------------------------------------------------------
def _loaded(rank, world, N, L):
    Bl = max(1, N // world)                                   # datasets/__init__.py:66
    b, t = torch.arange(Bl).view(Bl, 1, 1), torch.arange(L).view(1, L, 1)
    return (rank * 1000 + b * 10 + t).expand(Bl, L, 2).contiguous().float()


def test_oracle_rank_tables_match_reference_golden_g11(golden):
    from oracle.step_ref import ThreadWorld
    g = golden("g11_index_tables")
    for world, N, L in g["cases"].tolist():
        tw = ThreadWorld(world)

        def rank_fn(r):
            got = tw.prep_batch(r, _loaded(r, world, N, L), N, L)
            mine = (r * 100 + torch.arange(got.shape[0])).float().view(-1, 1)
            return got[:, 0].long(), tw.gather_views(r)(mine, L)[:, 0].long()
        res = tw.run(rank_fn)
        tag = "_w%d_N%d_L%d" % (world, N, L)
        assert torch.equal(torch.stack([r[0] for r in res]), torch.from_numpy(g["prep" + tag])), tag
        assert torch.equal(torch.stack([r[1] for r in res]), torch.from_numpy(g["gather" + tag])), tag


def _g11_worker(rank, world, port, cases, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "da-sac_amd"))
    import driver
    import models
    cfg = NS(**dict(DEFAULT_CFG, INIT_MODEL=""))
    net = models.get_model(cfg, rank, num_classes=19, criterion=nn.CrossEntropyLoss(ignore_index=255, reduction="none"))
    out = {}
    for (w, N, L) in cases:
        for how in ("all_gather", "p2p"):
            got = driver.prep_batch(_loaded(rank, world, N, L), N, L, exchange=how)
            out[("prep", how, w, N, L)] = got[:, 0].long()
        mine = (rank * 100 + torch.arange(got.shape[0])).float().view(-1, 1)
        out[("gather", w, N, L)] = net._gather(mine, L)[:, 0].long()
    q.put((rank, {k: v.tolist() for k, v in out.items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_product_prep_batch_and_gather_match_reference_golden_g11(golden, world):
    g = golden("g11_index_tables")
    cases = [c for c in g["cases"].tolist() if c[0] == world]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_g11_worker, args=(r, world, port, cases, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for w, N, L in cases:
        tag = "_w%d_N%d_L%d" % (w, N, L)
        for how in ("all_gather", "p2p"):
            assert [res[r][("prep", how, w, N, L)] for r in range(world)] == g["prep" + tag].tolist(), (tag, how)
        assert [res[r][("gather", w, N, L)] for r in range(world)] == g["gather" + tag].tolist(), tag


def test_prep_batch_single_process_and_bad_world():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "da-sac_amd"))
    import driver
    t = torch.arange(2 * 4 * 3).view(2, 4, 3)
    assert torch.equal(driver.prep_batch(t, 2, 4, rank=0, world=1), t.flatten(0, 1))     # train.py:186-187
    with pytest.raises(AssertionError, match="Batch size does not fit world size"):
        driver.prep_batch(t, 2, 4, rank=0, world=3)


def test_gradient_sink_layout_covers_every_trainable_parameter_in_backward_order():
    """dasac_hip.parallel.GradSink: slices of the flat reduction buffer are disjoint, 256-byte aligned, ordered as the
    backward pass completes them (ASPP first, stem last) and cut into contiguous buckets (host logic only)."""
    from types import SimpleNamespace as NS
    import torch.nn as nn
    import models
    from dasac_hip import engine as E
    from dasac_hip.parallel import GradSink, _trainable_backbones
    from oracle.step_ref import DEFAULT_CFG
    cfg = NS(**dict(DEFAULT_CFG, INIT_MODEL="", OPT_NESTEROV=False))
    net = models.get_model(cfg, 0, num_classes=19, criterion=nn.CrossEntropyLoss(ignore_index=255, reduction="none"))
    net.train()
    assert _trainable_backbones(net) == [net.backbone]           # the teacher has no gradients, SAC itself no plan
    eng = E.Engine(net.backbone._plan())
    need = [p.requires_grad for p in eng.params]
    sink = GradSink(bucket_bytes=32 << 20)
    sink._build_layout(eng, need)
    spans = sorted((sink._offsets[j], eng.params[j].numel(), j) for j in sink._offsets)
    assert len(spans) == sum(need) == 320
    for (o, n, _), (o2, _, _) in zip(spans, spans[1:]):
        assert o % 64 == 0 and o + n <= o2
    assert spans[-1][0] + spans[-1][1] <= sink._total
    names = {id(p): n for n, p in net.backbone.named_parameters()}
    assert names[id(eng.params[spans[0][2]])].startswith("model.layer5.")      # first slice: the classifier
    assert names[id(eng.params[spans[-1][2]])].startswith(("model.conv1", "model.bn1"))
    # buckets tile [0, total) without gaps; the first one is small so that the first reduction starts early
    assert sink._buckets[0][0] == 0 and sink._buckets[-1][1] == sink._total
    for (_, hi, _), (lo, _, _) in zip(sink._buckets, sink._buckets[1:]):
        assert hi == lo
    sizes = [(hi - lo) * 4 for lo, hi, _ in sink._buckets]
    assert sizes[0] < 32 << 20 and max(sizes) < 48 << 20 and len(sizes) >= 5


def test_gradient_sink_clears_only_the_slices_nobody_wrote():
    """when a bucket still waits for a member at the end of a backward pass (an op that got no gradient),
    `finish` must clear THAT member's slice only -- the other members of the bucket were already written and handed to
    autograd as views.  Also: every bucket is reduced exactly once per pass, and the layout cache follows the engine OBJECT
    (a rebuilt engine with a recycled id() must not reuse another plan's offsets).  Host logic only."""
    from dasac_hip.parallel import GradSink

    class FakeOp:
        def __init__(self, pidx):
            self.pidx = pidx

    def fake_engine(shapes, groups):
        return NS(params=[torch.zeros(s) for s in shapes], plan=NS(ops=[FakeOp(g) for g in groups]))

    eng = fake_engine([(4, 3), (5,), (2, 2), (7,)], [[0, 1], [2], [3]])
    sink = GradSink(bucket_bytes=1 << 30)                     # everything in one bucket
    need = [True, True, True, True]
    g = torch.ones(1, 1, 2, 2)
    sink.begin(eng, need, g)
    assert len(sink._buckets) == 1
    views = {j: sink.alloc(j) for j in range(4)}
    for j in (3, 2, 0):                                       # parameter 1 never gets a gradient in this pass
        views[j].fill_(float(j + 1))
    sink.done([3])
    sink.done([2])
    sink.done([0])
    views[1].fill_(float("nan"))                              # whatever the allocator left there
    sink.finish()
    assert torch.equal(views[0], torch.full((4, 3), 1.0)) and torch.equal(views[2], torch.full((2, 2), 3.0))
    assert torch.equal(views[3], torch.full((7,), 4.0)) and torch.equal(views[1], torch.zeros(5))
    assert sink._launches == [1]
    # a second pass where everything is written: nothing is cleared, done() twice for one parameter counts once
    sink.begin(eng, need, g)
    views = {j: sink.alloc(j) for j in range(4)}
    for j in range(4):
        views[j].fill_(7.0)
    sink.done([3, 3])
    assert sink._pending[0] == 3
    sink.done([2, 0, 1])
    assert sink._launches == [1] and sink._pending[0] == -1
    sink.finish()
    assert all(torch.equal(views[j], torch.full_like(views[j], 7.0)) for j in range(4))
    # another engine object (different plan, possibly the same id() after a rebuild): the layout is rebuilt
    eng2 = fake_engine([(3,), (6, 2)], [[0], [1]])
    sink.begin(eng2, [True, True], g)
    assert set(sink._offsets) == {0, 1} and tuple(sink.alloc(1).shape) == (6, 2)
    sink.done([1, 0])
    sink.finish()
