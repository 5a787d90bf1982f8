"""Round-6 probe: N processes on ONE device run the SAME seeded FCN-8s + SAC iterations (no process group).  Every process must
report the same sequence of (loss_ce, self_ce, teacher_diff) bit for bit (the step is deterministic); differences mean the device
does not execute correctly under N-process time slicing.  Usage: python tools/experiments/r6_eight_procs_identical.py <nproc> [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(rank, iters, q):
    for p in (ROOT, os.path.join(ROOT, "da-sac_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    from types import SimpleNamespace as NS
    import torch, torch.nn as nn
    import driver, models
    import test_gpu_sharded as T
    arch = "fcn_vgg16_bn"
    make_sd, hw, _ = T.ARCHS[arch]
    cfg = NS(**dict(T._cfg(arch), INIT_MODEL="", OPT_NESTEROV=False))
    net = models.get_model(cfg, 0, num_classes=19, criterion=nn.CrossEntropyLoss(ignore_index=255, reduction="none"), drop_rate=0.0)
    net.backbone.load_state_dict(make_sd(), strict=True)
    net.cuda().train()
    optim = driver.make_optimizer(net, cfg)
    rec = []
    for it in range(iters):
        src, tgt = driver.synthetic_batches(2, 1, 4, hw, "cuda", seed=100 + it)
        ls, lt, _ = driver.sac_train_iteration(net, optim, src, tgt, 4, it == 0, cfg.LR_TARGET)
        rec.append((float(ls["loss_ce"]), float(lt["self_ce"]), float(lt["teacher_diff"])))
    q.put((rank, rec))


if __name__ == "__main__":
    import torch.multiprocessing as mp
    n = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=main, args=(r, iters, q)) for r in range(n)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=400) for _ in ps)
    [p.join(30) for p in ps]
    ref = res[0][1]
    bad = [(r, i, rec[i], ref[i]) for r, rec in res for i in range(iters) if rec[i] != ref[i]]
    print("nproc", n, "iters", iters, "reference", ref[-1], "mismatches", len(bad), bad[:6])
