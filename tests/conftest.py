import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "da-sac_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Collection order: `pytest -m gpu -x` must reach the CONTRACT first -- the bit-exact kernels and the
# oracle / golden parity of the hot path -- and the multi-process and run-vs-run self-comparison files last, so that a flake
# in a wrapper test can never hide the parity evidence again.  Files not listed keep their alphabetical place in the middle.
_FIRST = ["test_gpu_pseudo_labels", "test_gpu_head", "test_gpu_views", "test_gpu_photometric", "test_gpu_conv", "test_gpu_models",
          "test_gpu_fullres", "test_gpu_bn_train", "test_gpu_optim", "test_driver_extras", "test_abi", "test_kernel_resources"]
_LAST = ["test_gpu_bf16x3", "test_gpu_no_aten_compute", "test_gpu_sharded", "test_gpu_ddp", "test_gpu_overlap_ddp",
         "test_gpu_zz_determinism"]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if name in _FIRST:
            return (0, _FIRST.index(name))
        if name in _LAST:
            return (2, _LAST.index(name))
        return (1, 0)
    items.sort(key=rank)          # stable: the order inside a file is kept


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


def rel_err(a, b):
    """max|a-b| / max|b|  -- the 'within 1e-3 rel' metric of BASELINE.json north_star."""
    import torch
    a = torch.as_tensor(a).detach().to(torch.float64).cpu()
    b = torch.as_tensor(b).detach().to(torch.float64).cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def init_ranks(rank, world):
    """Process group of a multi-rank GPU test.  On a box with at least `world` GPUs every rank takes its OWN device and the
    transport is RCCL ("nccl" on ROCm) -- the first multi-GPU box exercises the real collectives with no code change
    (VERDICT r3 item 10); on the usual 1-GPU box the ranks share device 0 and go through gloo (RCCL refuses two ranks on one
    device).  DASAC_TEST_BACKEND overrides.  Returns the device index of this rank."""
    import torch
    import torch.distributed as dist
    multi = torch.cuda.device_count() >= world
    backend = os.environ.get("DASAC_TEST_BACKEND", "nccl" if multi else "gloo")
    dev = rank if multi else 0
    torch.cuda.set_device(dev)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    kw = {"device_id": torch.device("cuda", dev)} if backend == "nccl" else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dev


def run_ranks(target, world, make_args, timeout=300, attempts=2):
    """Spawns `world` rank processes (`target(*make_args(rank, port, queue))`), collects one queue item per rank and returns them
    sorted by rank.  Bounded: a rank that dies or hangs costs `timeout` seconds, not the suite's budget.  A rank PROCESS failure (crash,
    non-zero exit, nothing on the queue in time) is retried once with fresh processes and a fresh port -- round 6 saw one
    `HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION` queue abort in an 8-processes-on-one-device run that two reruns did not reproduce
    (profiles/ROUND6.md).  Numerical assertions are the caller's and are never retried."""
    import queue as _queue
    import socket
    import sys
    import torch.multiprocessing as mp
    last = None
    for attempt in range(attempts):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=target, args=make_args(r, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = []
        try:
            for _ in procs:
                got.append(q.get(timeout=timeout))
        except _queue.Empty:
            last = "only {} of {} ranks reported within {} s".format(len(got), world, timeout)
        for p in procs:
            p.join(60 if len(got) == world else 5)
            if p.is_alive():
                p.kill()              # the exact processes this call started
                p.join(10)
        codes = [p.exitcode for p in procs]
        if len(got) == world and all(c == 0 for c in codes):
            return sorted(got, key=lambda t: t[0])
        last = "{}; exit codes {}".format(last or "ranks exited abnormally", codes)
        print("run_ranks: attempt {} failed ({}){}".format(attempt + 1, last, "; retrying" if attempt + 1 < attempts else ""), file=sys.stderr)
    raise AssertionError("rank processes failed {} times: {}".format(attempts, last))
