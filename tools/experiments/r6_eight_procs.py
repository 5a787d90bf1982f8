"""Round-6 repro: N processes on ONE device, each running RN101-SAC two-pass iterations at 33x49 with `reserved` CUs left free
(no process group at all).  Usage: python tools/experiments/r6_eight_procs.py <nproc> <reserved> [iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(rank, reserved, iters, q):
    for p in (ROOT, os.path.join(ROOT, "da-sac_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch
    import driver
    from dasac_hip import lib as L
    import test_gpu_overlap_ddp as T
    L.load().dasac_set_reserved_cus(reserved)
    cfg, net = T._build(seed=3)
    src, tgt = driver.synthetic_batches(2, 2, 4, (33, 49), "cuda", seed=50 + rank)
    t0 = time.time()
    for _ in range(iters):
        T._two_passes(net, src, tgt, cfg.LR_TARGET, T=4)
    torch.cuda.synchronize()
    q.put((rank, time.time() - t0))


if __name__ == "__main__":
    import torch.multiprocessing as mp
    n, reserved = int(sys.argv[1]), int(sys.argv[2])
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=main, args=(r, reserved, iters, q)) for r in range(n)]
    [p.start() for p in ps]
    res = []
    try:
        for _ in ps:
            res.append(q.get(timeout=240))
    except Exception as e:
        print("TIMEOUT / failure after", len(res), "ranks:", repr(e))
    for p in ps:
        p.join(5)
        if p.is_alive():
            p.kill()
    print("nproc", n, "reserved", reserved, "done", sorted(res), "exit codes", [p.exitcode for p in ps])
