"""Headline benchmark: train images/sec of the da-sac per-step hot path on MI355X.

Workload (BASELINE.json configs[2], "cfg-3"): ResNet-101 DeepLabv2 + SAC, per GPU 8 source crops +
2 groups x 4 views of target crops at 769x769, 19 classes, frozen BN, fp32.  One step =
source fwd/bwd -> target fwd (teacher fwd + fusion + pseudo labels) -> target bwd -> SGD step
(reference train.py:266-298).  images/sec follows the reference's counter (source images only,
train.py:314); N>1 is weak scaling under DistributedDataParallel over RCCL.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 769] [--no-cpu-baseline]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "da-sac_amd"))

# stdout carries exactly ONE line (the JSON).  Libraries print banners on file descriptor 1 behind Python's back (RCCL:
# "RCCL version : ..." at communicator teardown), so fd 1 itself is pointed at stderr for the whole run and the JSON
# line is written to the saved descriptor at the very end.
_JSON_FD = os.dup(1)
os.dup2(2, 1)

import torch
import torch.distributed as dist
import torch.nn as nn

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0     # same guide: dense bf16 matrix peak; a split-bf16 product costs three MFMAs


def model_cfg(arch="deeplabv2_resnet101", baseline=False):
    from types import SimpleNamespace as NS
    # core/config.py:130-159 + configs/deeplabv2_resnet101_train.yaml (fcn_vgg16_train.yaml: LR 5e-4, LR_TARGET 2)
    if arch == "fcn_vgg16_bn":
        return NS(ARCH=arch, INIT_MODEL="", BASELINE=baseline, LR=5e-4, LR_TARGET=2.0, WEIGHT_DECAY=5e-4,
                  MOMENTUM=0.9, OPT_NESTEROV=False, STAT_MOMENTUM=0.99, NET_MOMENTUM=0.99, NET_MOMENTUM_ITER=100,
                  CONF_DISCOUNT=True, CONF_POOL_ON=True, CONF_POOL="avg_pool", FOCAL_P=3, LOSS="focal_ce_conf",
                  RUN_CONF_UPPER=0.75, RUN_CONF_LOWER=0.2, THRESHOLD_BETA=1e-3)
    return NS(ARCH=arch, INIT_MODEL="", BASELINE=baseline, LR=2.5e-4, LR_TARGET=5.0, WEIGHT_DECAY=5e-4,
              MOMENTUM=0.9, OPT_NESTEROV=False, STAT_MOMENTUM=0.99, NET_MOMENTUM=0.99, NET_MOMENTUM_ITER=100,
              CONF_DISCOUNT=True, CONF_POOL_ON=True, CONF_POOL="avg_pool", FOCAL_P=3, LOSS="focal_ce_conf",
              RUN_CONF_UPPER=0.75, RUN_CONF_LOWER=0.2, THRESHOLD_BETA=1e-3)


def cpu_baseline(size):
    """The CPU oracle (a port of the reference's path, pinned to it by tests/) timed on this host's cores on a BOUNDED
    sample of the SAME workload at FULL resolution: 1/8 of a cfg-3 step -- one source crop and one target crop (L = 1)
    student forward + backward, one teacher forward, the SAC head and the SGD step, all at size x size (no pixel-ratio
    extrapolation).  images/sec = 1 source image per sample time, as the per-step metric counts source images only."""
    from oracle import nets_ref as N
    from oracle.step_ref import SacOracle, SgdOracle, sac_train_iteration
    import driver
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(avail, 32))           # ATen's CPU convs stop scaling (and oversubscribe badly) beyond this
    torch.set_num_threads(cores)
    m = SacOracle(N.resnet101_state(seed=0, randomize_bn=True, he_init=True, residual_gain=0.1, aspp_gain=6.0))
    opt = SgdOracle(m)
    src, tgt = driver.synthetic_batches(1, 1, 1, (size, size), "cpu", seed=1)
    m.running_conf.fill_(0.05)
    m.slow_init[0] = 1.0
    t0 = time.time()
    sac_train_iteration(m, opt, src, tgt, 1, update_teacher=False)
    dt = time.time() - t0
    return {"value": round(1.0 / dt, 5), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "1/8 of one cfg-3 step at full resolution (1 source + 1 target crop {0}x{0}: student fwd+bwd each, 1 teacher fwd, "
                      "SAC head, SGD) = {1:.1f} s on {2} threads; 1 source image per sample".format(size, dt, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=769)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--groups", type=int, default=2)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3"],
                    help="arithmetic of the forward/data-gradient GEMMs for the headline number: exact fp32 MFMA (default) or the "
                         "split-bf16 path (3 bf16 MFMAs per product, fp32 accumulate)")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra (untimed-by-the-contract) run in the other precision")
    ap.add_argument("--config", default="cfg3", choices=["cfg2", "cfg3", "cfg5"],
                    help="cfg3 (default, the headline): RN101+SAC 8+2x4 crops @769^2; cfg2: RN101 baseline/AdaBN step, 2 source + 2 "
                         "target crops @769^2, train-mode BN; cfg5: VGG16-FCN8s + SAC @512x1024 (per-GPU 8 + 2x4 crops)")
    args = ap.parse_args()
    arch, baseline, hw = "deeplabv2_resnet101", False, (args.size, args.size)
    if args.config == "cfg2":
        baseline, args.batch, args.groups, args.views = True, 2, 2, 1
    elif args.config == "cfg5":
        arch, hw = "fcn_vgg16_bn", (512, 1024)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node {}".format(args.gpus)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_ddp = world == 1 and os.environ.get("DASAC_BENCH_DDP") == "1"     # measure the DDP wrapper's own cost on one GPU
    if world > 1 or force_ddp:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if force_ddp:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        else:
            dist.init_process_group("nccl", device_id=dev)

    import models
    import driver
    from dasac_hip import ops

    cfg = model_cfg(arch, baseline)
    # stdout carries exactly ONE line (the JSON); the model constructors' progress prints go to stderr on rank 0
    sys.stdout = sys.stderr if rank == 0 else open(os.devnull, "w")
    net = models.get_model(cfg, local, num_classes=19, criterion=nn.CrossEntropyLoss(ignore_index=255, reduction="none"))
    driver.init_synthetic_weights(net, seed=0)
    net.cuda(local).train()
    if not baseline:
        net.running_conf.fill_(0.05)
    optim = driver.make_optimizer(net, cfg)
    step_net = nn.parallel.DistributedDataParallel(net, device_ids=[local]) if (world > 1 or force_ddp) else net
    src, tgt = driver.synthetic_batches(args.batch, args.groups, args.views, hw, dev, seed=rank)
    if arch != "deeplabv2_resnet101":
        driver.calibrate_classifier(net, src[0][:1])          # logits std ~3 whatever the backbone's feature scale
    src = (src[0], driver.self_consistent_labels(net, src[0]))

    def step(i):
        if baseline:      # train.py:274-289: source fwd/bwd/step + no-grad train-mode target forward (AdaBN)
            l = driver.baseline_train_iteration(step_net, optim, src, tgt[0])
            return None, l, None
        tgt_i = (tgt[0], tgt[1].clone(), tgt[2], tgt[3], tgt[4])      # forward rewrites -1 -> 255 in place
        return driver.sac_train_iteration(step_net, optim, src, tgt_i, args.views, update_teacher=(i == 0),
                                          lr_target=cfg.LR_TARGET)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(first, warmup):
        """`warmup` untimed steps, then exactly args.steps timed ones between fences; max over ranks."""
        for i in range(warmup):
            step(first + i)
        fence()
        ops.PROFILE.start()
        t0 = time.perf_counter()
        for i in range(args.steps):
            res = step(first + warmup + i)
        fence()
        dt_ = time.perf_counter() - t0
        prof_ = ops.PROFILE.stop()
        if world > 1:
            tmax = torch.tensor([dt_], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_ = float(tmax)
        return dt_, prof_, res

    def kernel_table(prof_):
        """GEMM kernels: achieved TFLOP/s; streaming kernels: achieved TB/s of ALGORITHMIC bytes (each operand once)."""
        out_ = {}
        for k, v in prof_.items():
            sec = max(v["seconds"], 1e-12)
            row = {"ms_per_step": round(v["seconds"] / args.steps * 1e3, 3), "launches_per_step": v["launches"] // args.steps}
            if v["flops"] > 0:
                row["tflops"] = round(v["flops"] / sec / 1e12, 2)
            if v.get("bytes", 0) > 0:
                row["algorithmic_TB_per_s"] = round(v["bytes"] / sec / 1e12, 3)
                row["algorithmic_MB_per_launch"] = round(v["bytes"] / max(v["launches"], 1) / 1e6, 1)
            out_[k] = row
        return out_

    ops.set_precision(args.precision)
    dt, prof, out = measure(0, args.warmup)
    alt = None
    if not args.no_alt and not baseline:
        # the same K steps once more in the other arithmetic (outside the contract's timed region; reported as "alt")
        other = "bf16x3" if args.precision == "fp32" else "fp32"
        ops.set_precision(other)
        try:
            dt2, prof2, _ = measure(args.warmup + args.steps, 1)
        except Exception as exc:      # the extra leg must never cost the contract's line
            dt2, prof2 = None, None
            alt = {"dtype": other, "error": repr(exc)[:200]}
        ops.set_precision(args.precision)
    if not args.no_alt and not baseline and dt2 is not None:
        alt = {"dtype": other, "value": round(world * args.batch * args.steps / dt2, 4), "unit": "images/sec",
               "ms_per_step": round(dt2 / args.steps * 1e3, 3), "kernels": kernel_table(prof2),
               "note": "same step with the forward/data-gradient GEMMs in {} arithmetic; weight gradient and everything else "
                       "unchanged".format("split-bf16 (3 bf16 MFMAs per product, fp32 accumulate; tests/test_gpu_bf16x3.py)"
                                          if other == "bf16x3" else "exact fp32 MFMA")}
    losses = {k: float(v.detach().mean()) for k, v in out[1].items()}
    labelled = float((out[2]["teacher_labels"] != 255).float().mean()) if out[2] is not None else 0.0

    if rank == 0:
        gemm = {k: v for k, v in prof.items() if k.startswith("conv_gemm")}
        dom_name = max(gemm, key=lambda k: gemm[k]["seconds"]) if gemm else "conv_gemm"
        dom = gemm.get(dom_name, {"flops": 0.0, "seconds": 1.0, "launches": 0})
        ach = dom["flops"] / max(dom["seconds"], 1e-12) / 1e12
        peak = PEAK_FP32_MFMA_TFLOPS if args.precision == "fp32" else PEAK_BF16_MFMA_TFLOPS / 3.0
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")     # PMC-measured HBM bytes per launch (separate rocprofv3 passes)
        if os.path.isfile(tfile):
            traffic = json.load(open(tfile)).get(dom_name)
        line = {
            "metric": "train images/sec (769x769, 19-cls, RN101 DeepLabv2, K=3)",
            "value": round(world * args.batch * args.steps / dt, 4), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "bf16x3 (fp32 operands split into bf16 head+tail, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": {"cfg3": "cfg-3: RN101-DeepLabv2 + SAC, per GPU {} source + {}x{} target crops @{}x{}, frozen BN, "
                                            "random-init weights",
                                    "cfg2": "cfg-2: RN101-DeepLabv2 baseline/AdaBN step, per GPU {} source crops fwd+bwd+SGD + {}x{} target "
                                            "crops no-grad train-mode fwd @{}x{}, batch-statistics BN, random-init weights",
                                    "cfg5": "cfg-5: VGG16-FCN8s + SAC, per GPU {} source + {}x{} target crops @{}x{}, frozen BN, Dropout2d "
                                            "p=0.1, random-init weights"}[args.config].format(args.batch, args.groups, args.views, hw[0], hw[1]),
                       "global_batch": world * args.batch, "crops_per_step": world * (args.batch + args.groups * args.views),
                       "parallelism": "dp{}".format(world)},
            "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                         "frac": round(ach / peak, 4), "traffic": traffic,
                         "kernel": "dasac::" + dom_name + " (forward + data-gradient implicit GEMM, {})".format(
                             "fp32 MFMA" if args.precision == "fp32" else "3 bf16 MFMAs per fp32 product: peak = bf16 peak / 3"),
                         "algorithmic_gflop_per_launch": round(dom["flops"] / max(dom["launches"], 1) / 1e9, 2),
                         "algorithmic_MB_per_launch": round(dom.get("bytes", 0.0) / max(dom["launches"], 1) / 1e6, 1),
                         "launches": dom["launches"], "avg_launch_ms": round(dom["seconds"] / max(dom["launches"], 1) * 1e3, 4)},
            "kernels": kernel_table(prof),
            "check": {"loss_ce": losses.get("loss_ce"), "self_ce": losses.get("self_ce"), "teacher_diff": losses.get("teacher_diff"),
                      "labelled_frac": round(labelled, 4)},
        }
        if alt is not None:
            line["alt"] = alt
        if world == 1 and not args.no_cpu_baseline and args.config == "cfg3":
            try:
                line["cpu_baseline"] = cpu_baseline(args.size)
            except Exception as exc:      # never lose the line to the reported baseline
                line["cpu_baseline"] = {"value": None, "unit": "images/sec", "cores": 0, "kind": "port", "sample": "failed: " + repr(exc)[:160]}
        os.write(_JSON_FD, (json.dumps(line) + "\n").encode())
    if world > 1 or force_ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
