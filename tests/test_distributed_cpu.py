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
