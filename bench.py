"""Headline benchmark: train images/sec of the da-sac per-step hot path on MI355X.

Workload (BASELINE.json configs[2], "cfg-3"): ResNet-101 DeepLabv2 + SAC, per GPU 8 source crops +
2 groups x 4 views of target crops at 769x769, 19 classes, frozen BN, fp32.  One step =
source fwd/bwd -> target fwd (teacher fwd + fusion + pseudo labels) -> target bwd -> SGD step
(reference train.py:266-298).  images/sec follows the reference's counter (source images only,
train.py:314); N>1 is weak scaling, one process per GPU over RCCL, gradients all-reduced bucket by
bucket from inside the backward pass (dasac_hip.parallel.OverlappedDataParallel).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 769] [--no-cpu-baseline]

`--gpus N` with N > 1 launches its own ranks when it was not started by torch.distributed.run (the reference does the
same with mp.spawn, train.py:553-557): free-port rendezvous on 127.0.0.1, one process per GPU.

The timed region (exactly K steps between two fences) carries NO instrumentation; the per-kernel table and the roofline
figure come from a second, event-instrumented pass over a few more steps of the same workload.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "da-sac_amd"))

import torch
import torch.distributed as dist
import torch.nn as nn

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0     # same guide: dense bf16 matrix peak; a split-bf16 product costs three MFMAs

CRITERION = dict(ignore_index=255, reduction="none")


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


# ----------------------------------------------------------------------------------------------------------------
# CPU baseline + full-resolution parity: ONE bounded sample of the workload, run by the oracle (timed) and by the HIP path
# ----------------------------------------------------------------------------------------------------------------
# (round 5: EVERY trainable tensor of the student is compared at full resolution, gradients and updated parameters -- 320 tensors for
# ResNet-101; rounds 3-4 sampled nine)


def csrc_sha1():
    """Hash of the kernel sources (file names + contents of da-sac_amd/csrc): identifies what a PMC measurement was taken on."""
    import glob
    import hashlib
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "da-sac_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def _sample_inputs(size):
    import driver
    return driver.synthetic_batches(1, 1, 1, (size, size), "cpu", seed=1)


def cpu_sample(size):
    """The CPU oracle (a port of the reference's path, pinned to it by tests/) on a BOUNDED sample of the SAME workload
    at FULL resolution: 1/8 of a cfg-3 step -- one source crop and one target crop (L = 1): student forward + backward
    each, one teacher forward, the SAC head and the SGD step, all at size x size (no pixel-ratio extrapolation).
    Returns (seconds, cores, what the oracle computed) -- the outputs are what `parity_fullres` checks the HIP path against."""
    from oracle import nets_ref as N
    from oracle.step_ref import SacOracle, SgdOracle, sac_train_iteration
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(avail, 32))           # ATen's CPU convs stop scaling (and oversubscribe badly) beyond this
    torch.set_num_threads(cores)
    sd = N.resnet101_state(seed=0, randomize_bn=True, he_init=True, residual_gain=0.1, aspp_gain=6.0)
    m = SacOracle(sd)
    opt = SgdOracle(m)
    src, tgt = _sample_inputs(size)
    m.running_conf.fill_(0.05)
    m.slow_init[0] = 1.0
    t0 = time.time()
    ls, lt, outs = sac_train_iteration(m, opt, src, tuple(t.clone() for t in tgt), 1, update_teacher=False)
    dt = time.time() - t0
    ref = {"loss_ce": ls["loss_ce"], "self_ce": lt["self_ce"], "teacher_diff": lt["teacher_diff"],
           "labels": outs["teacher_labels"], "running_conf": m.running_conf.clone(),
           "grads": {k: v.grad.detach().clone() for k, v in m.student.items() if getattr(v, "grad", None) is not None},
           "params": {k: v.detach().clone() for k, v in m.student.items() if getattr(v, "grad", None) is not None}}
    return dt, cores, sd, ref


def cpu_baseline_entry(size, dt, cores):
    return {"value": round(1.0 / dt, 5), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "1/8 of one cfg-3 step at full resolution (1 source + 1 target crop {0}x{0}: student fwd+bwd each, 1 teacher fwd, "
                      "SAC head, SGD) = {1:.1f} s on {2} threads; 1 source image per sample".format(size, dt, cores)}


def hip_sample(size, sd, device, fuse=True):
    """The same sample through the HIP module (fresh model, the oracle's weights and inputs); fuse: the student schedule."""
    import models
    import driver
    cfg = model_cfg()
    net = models.get_model(cfg, device.index or 0, num_classes=19, criterion=nn.CrossEntropyLoss(**CRITERION))
    net.backbone.load_state_dict(sd, strict=True)
    net.slow_net.load_state_dict(sd, strict=True)
    net.to(device).train()
    net.running_conf.fill_(0.05)
    net.slow_init[0] = 1.0
    optim = driver.make_optimizer(net, cfg)
    src, tgt = _sample_inputs(size)
    to = lambda ts: tuple(t.to(device) for t in ts)
    # sum_grads_in_optimizer=False: .grad holds source + target gradients after the step, like the reference's
    ls, lt, outs = driver.sac_train_iteration(net, optim, to(src), to(tgt), 1, False, cfg.LR_TARGET, sum_grads_in_optimizer=False,
                                              fuse_passes=fuse)
    torch.cuda.synchronize(device)
    named = dict(net.backbone.named_parameters())
    return {"loss_ce": float(ls["loss_ce"]), "self_ce": float(lt["self_ce"]), "teacher_diff": float(lt["teacher_diff"]),
            "labels": outs["teacher_labels"].cpu(), "running_conf": net.running_conf.detach().cpu(),
            "grads": {k: v.grad.detach().cpu() for k, v in named.items() if v.grad is not None},
            "params": {k: v.detach().cpu() for k, v in named.items() if v.grad is not None}}


def compare_sample(ref, got):
    """Differences of the HIP path from the oracle on the full-resolution sample (the numbers the bench line reports and
    tests/test_gpu_fullres.py bounds)."""
    rel = lambda a, b: abs(float(a) - float(b)) / max(abs(float(b)), 1e-12)
    tmax = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
    keys = sorted(ref["grads"])
    assert keys == sorted(got["grads"]), "the oracle and the HIP path differentiate different parameter sets"
    g_err = {k: tmax(got["grads"][k], ref["grads"][k]) for k in keys}
    p_err = {k: tmax(got["params"][k], ref["params"][k]) for k in keys}
    g_worst, p_worst = max(keys, key=g_err.get), max(keys, key=p_err.get)
    return {"loss_ce_rel": rel(got["loss_ce"], ref["loss_ce"]), "self_ce_rel": rel(got["self_ce"], ref["self_ce"]),
            "label_mismatch_frac": float((got["labels"] != ref["labels"]).double().mean()),
            "labelled_frac": float((ref["labels"] != 255).double().mean()),
            "running_conf_max_abs": float((got["running_conf"] - ref["running_conf"]).abs().max()),
            "grad_max_err_over_tensor_max": g_err[g_worst], "grad_worst_tensor": g_worst,
            "grad_median_err_over_tensor_max": sorted(g_err.values())[len(keys) // 2],
            "param_max_err_over_tensor_max": p_err[p_worst], "param_worst_tensor": p_worst,
            "sampled_tensors": len(keys)}


def parity_fullres(size, device=None, sample=None, fuse=True):
    """(timing of the oracle, comparison dict) of the 1 source + 1 target crop sample at size x size."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    dt, cores, sd, ref = cpu_sample(size) if sample is None else sample
    cmp_ = compare_sample(ref, hip_sample(size, sd, device, fuse))
    cmp_["size"], cmp_["student_schedule"] = size, "fused" if fuse else "two passes"
    return (dt, cores), cmp_


def head_parity_fullres(size=769, groups=2, views=4, device=None, seed=17):
    """Full-resolution parity of the MULTI-VIEW head alone (no backbone, so the oracle needs seconds): synthetic teacher logits
    [groups*views, 19, (size-1)/8+1, ...], the four SURVEY 8d affines per group, padded rows -> `SAC._refine` (upsample +
    softmax + class prior, warp, L-view fusion, warp back) -> thresholds -> pseudo labels -> `_focal_ce_conf` (the [B,B,H,W]
    broadcast of sac.py:148 with B = groups*views) value and gradient w.r.t. the student's stride-8 logits, against
    oracle.head_ref on the same inputs (models/sac.py:134-149,238-313).  Returns (oracle seconds, comparison dict)."""
    import driver
    import models
    from dasac_hip import engine as E, ops
    from oracle import head_ref as R
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    cfg = model_cfg()
    B, h = groups * views, (size - 1) // 8 + 1
    g = torch.Generator().manual_seed(seed)
    (_, _), (_, gt, frames, theta, theta_inv) = driver.synthetic_batches(1, groups, views, (size, size), "cpu", seed=seed)
    # teacher logits of a group's views = one blobby scene per group seen through each view's inverse affine (what augmented
    # views of ONE image look like: the aligned views agree, the fused probabilities are confident and the thresholds fire)
    coarse = torch.randint(0, 19, (groups, (h + 7) // 8, (h + 7) // 8), generator=g)
    scene = torch.nn.functional.one_hot(coarse.repeat_interleave(8, 1).repeat_interleave(8, 2)[:, :h, :h], 19).permute(0, 3, 1, 2).float() * 8
    scene = (scene + torch.randn(groups, 19, h, h, generator=g)).repeat_interleave(views, 0)
    teacher_logits = R.warp_affine(scene, theta_inv) + 0.5 * torch.randn(B, 19, h, h, generator=g)
    student_logits = teacher_logits * 0.5 + torch.randn(B, 19, h, h, generator=g)
    ignore = gt == -1
    chi0 = torch.full((19,), 0.05)
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 32)))
    # ---- oracle
    t0 = time.time()
    refined_r, chi_r, diags_r = R.refine(frames, teacher_logits, views, theta, theta_inv, ignore, chi0.clone(), beta=cfg.THRESHOLD_BETA,
                                         stat_momentum=cfg.STAT_MOMENTUM, training=True, pool=True, pool_kind=cfg.CONF_POOL)
    labels_r, conf_r, _ = R.pseudo_labels(refined_r, ignore, cfg.RUN_CONF_UPPER, cfg.RUN_CONF_LOWER, R.threshold_discount(chi_r, cfg.THRESHOLD_BETA))
    sl_r = student_logits.clone().requires_grad_(True)
    loss_r, _ = R.focal_ce_conf(R.upsample_bilinear_ac(sl_r, size, size), labels_r, conf_r, chi_r, cfg.FOCAL_P)
    loss_r.backward()
    dt = time.time() - t0
    # ---- HIP module (head methods only; the two backbones are never run)
    net = models.get_model(cfg, device.index or 0, num_classes=19, criterion=nn.CrossEntropyLoss(**CRITERION))
    net.to(device).train()
    net.running_conf.copy_(chi0)
    net.slow_init[0] = 1.0
    dev = lambda t: t.to(device)
    refined, diags = net._refine(dev(frames), dev(teacher_logits), views, dev(theta), dev(theta_inv), dev(ignore), pool=True)
    disc, fw = net._class_vectors.finish(cfg.THRESHOLD_BETA, cfg.FOCAL_P, cfg.CONF_DISCOUNT)
    labels, conf, _ = ops.pseudo_labels(refined, dev(ignore), cfg.RUN_CONF_UPPER, cfg.RUN_CONF_LOWER, disc)
    chi_hip = net.running_conf.detach().cpu().clone()
    # the integer contract: on EQUAL probabilities (the oracle's, uploaded) and the module's own threshold path
    net.running_conf.copy_(chi_r)
    disc_eq, fw_eq = ops.class_vectors(net.running_conf, cfg.THRESHOLD_BETA, cfg.FOCAL_P)
    labels_eq, conf_eq, _ = ops.pseudo_labels(dev(refined_r), dev(ignore), cfg.RUN_CONF_UPPER, cfg.RUN_CONF_LOWER, disc_eq)
    # the loss on equal labels / confidences (the oracle's), differentiated down to the stride-8 logits
    sl = dev(student_logits).requires_grad_(True)
    loss = E.focal_ce(E.upsample_bilinear(sl, (size, size)), dev(labels_r), fw_eq, dev(conf_r))
    loss.backward()
    # and end to end on the HIP side's own labels
    with torch.no_grad():
        loss_e2e = E.focal_ce(E.upsample_bilinear(dev(student_logits), (size, size)), labels, fw, conf)
    torch.cuda.synchronize(device)
    tmax = lambda a, b: float((a.double().cpu() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
    cmp_ = {"size": size, "crops": B, "views_per_group": views,
            "refined_max_abs": float((refined.cpu() - refined_r).abs().max()),
            "teacher_aligned_max_abs": float((diags["teacher_aligned"].cpu() - diags_r["teacher_aligned"]).abs().max()),
            "frames_aligned_rel": tmax(diags["frames_aligned"], diags_r["frames_aligned"]),
            "running_conf_max_abs": float((chi_hip - chi_r).abs().max()),
            "labels_equal_on_equal_probs": bool(torch.equal(labels_eq.cpu(), labels_r)),
            "conf_equal_on_equal_probs": bool(torch.equal(conf_eq.cpu(), conf_r)),
            "labelled_frac": float((labels_r != 255).double().mean()),
            "label_mismatch_frac_end_to_end": float((labels.cpu() != labels_r).double().mean()),
            "self_ce_rel": abs(float(loss) - float(loss_r)) / max(abs(float(loss_r)), 1e-12),
            "self_ce_rel_end_to_end": abs(float(loss_e2e) - float(loss_r)) / max(abs(float(loss_r)), 1e-12),
            "dlogits_max_err_over_tensor_max": tmax(sl.grad, sl_r.grad)}
    return dt, cmp_


def pillow_views_ms(hw, n_views, seed=0):
    """What the reference's DataLoader worker does per target image after the base crop (dataloader_target.py:281-306), timed
    with Pillow itself on ONE host thread: per view hflip (p = .5) + zoom window crop -> resize back (BILINEAR image, NEAREST
    label and mask; tf_target.py:141-239), GaussianBlur(radius ~ U(.1, 2)), colour jitter (p = .5: brightness, contrast,
    saturation, hue through HSV) and greyscale (p = .2) on the student's copy (tf_target.py:331-390), ToTensor + Normalize of both
    copies.  A CPU comparator for `ms_per_step_with_device_views` (not a parity oracle: the bit-exact one is oracle/views_ref.py)."""
    import random
    import numpy as np
    from PIL import Image, ImageEnhance, ImageFilter
    H, W = hw
    rng = random.Random(seed)
    gen = np.random.RandomState(seed)
    image = Image.fromarray(gen.randint(0, 256, (H, W, 3)).astype(np.uint8))
    label = Image.fromarray(gen.randint(0, 19, (H, W)).astype(np.uint8))
    mask = Image.fromarray((gen.rand(H, W) < 0.02).astype(np.uint8))
    mean, std = np.array([0.485, 0.456, 0.406], np.float32), np.array([0.229, 0.224, 0.225], np.float32)
    t0 = time.perf_counter()
    for k in range(n_views):
        im, lb, mk = image, label, mask
        if rng.random() > 0.5:
            im, lb, mk = (t.transpose(Image.FLIP_LEFT_RIGHT) for t in (im, lb, mk))
        if k > 0:
            sc = rng.uniform(0.5, 1.0)
            h, w = int(sc * H), int(sc * W)
            ii, jj = rng.randint(0, H - h), rng.randint(0, W - w)
            box = (jj, ii, jj + w, ii + h)
            im = im.crop(box).resize((W, H), Image.BILINEAR)
            lb = lb.crop(box).resize((W, H), Image.NEAREST)
            mk = mk.crop(box).resize((W, H), Image.NEAREST)
        im1 = im.filter(ImageFilter.GaussianBlur(rng.uniform(0.1, 2.0)))
        if rng.random() < 0.5:
            im1 = ImageEnhance.Brightness(im1).enhance(rng.uniform(0.6, 1.4))
            im1 = ImageEnhance.Contrast(im1).enhance(rng.uniform(0.6, 1.4))
            im1 = ImageEnhance.Color(im1).enhance(rng.uniform(0.6, 1.4))
            hsv = np.array(im1.convert("HSV"))
            hsv[..., 0] += np.uint8(int(rng.uniform(-0.1, 0.1) * 255))
            im1 = Image.fromarray(hsv, "HSV").convert("RGB")
        if rng.random() < 0.2:
            im1 = im1.convert("L").convert("RGB")
        m = np.asarray(mk) != 0
        for src_ in (im1, im):
            f = (np.asarray(src_, np.float32) / 255.0 - mean) / std
            f[m] = 0.0
            np.ascontiguousarray(f.transpose(2, 0, 1))
        g_ = np.asarray(lb).astype(np.int64)
        g_[m] = -1
    return (time.perf_counter() - t0) * 1e3


def time_other_config(config, dev, steps=5, warmup=1):
    """cfg-2 / cfg-5 of BASELINE.json in the driver-timed process: `steps` un-instrumented steps between
    fences -> ms_per_step; one more step with HIP events around the GEMM launches -> algorithmic conv TFLOP per step, so that
    tflops = that / the UN-instrumented step time and frac = tflops / the fp32 matrix peak (whole step, not a kernel)."""
    import driver
    import models
    from dasac_hip import ops
    if config == "cfg2":
        arch, baseline, hw, batch, groups, views = "deeplabv2_resnet101", True, (769, 769), 2, 2, 1
    else:
        arch, baseline, hw, batch, groups, views = "fcn_vgg16_bn", False, (512, 1024), 8, 2, 4
    cfg = model_cfg(arch, baseline)
    net = models.get_model(cfg, dev.index, num_classes=19, criterion=nn.CrossEntropyLoss(**CRITERION))
    driver.init_synthetic_weights(net, seed=0)
    net.cuda(dev.index).train()
    if not baseline:
        net.running_conf.fill_(0.05)
    optim = driver.make_optimizer(net, cfg)
    src, tgt = driver.synthetic_batches(batch, groups, views, hw, dev, seed=0)
    if arch != "deeplabv2_resnet101":
        driver.calibrate_classifier(net, src[0][:1])
    src = (src[0], driver.self_consistent_labels(net, src[0]))

    fuse = {"on": False}

    def step(i):
        if baseline:
            return driver.baseline_train_iteration(net, optim, src, tgt[0])
        t = (tgt[0], tgt[1].clone(), tgt[2], tgt[3], tgt[4])
        return driver.sac_train_iteration(net, optim, src, t, views, update_teacher=(i == 0), lr_target=cfg.LR_TARGET, fuse_passes=fuse["on"])

    def timed(first):
        for i in range(warmup):
            step(first + i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(first + warmup + i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    ms = timed(0)                               # the module API as train.py calls it (two student passes in SAC mode)
    ms_fused = None
    if not baseline:
        fuse["on"] = True
        ms_fused = round(timed(warmup + steps), 3)
        fuse["on"] = False
    ops.PROFILE.start()
    step(2 * (warmup + steps))
    prof = ops.PROFILE.stop()
    tflop = sum(v["flops"] for v in prof.values()) / 1e12
    return {"ms_per_step": round(ms, 3), "ms_per_step_fused_schedule": ms_fused,
            "images_per_sec": round(batch / ms * 1e3, 3), "conv_tflop_per_step": round(tflop, 3),
            "tflops": round(tflop / ms * 1e3, 2), "frac_of_fp32_matrix_peak": round(tflop / ms * 1e3 / PEAK_FP32_MFMA_TFLOPS, 4),
            "workload": "{} source + {}x{} target crops @{}x{}, {}".format(batch, groups, views, hw[0], hw[1],
                                                                          "baseline/AdaBN, batch-statistics BN" if baseline else arch + " + SAC"),
            "steps": steps, "warmup": warmup}


# ----------------------------------------------------------------------------------------------------------------
def self_launch(args, json_fd):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: start the N ranks ourselves, one per GPU, free-port
    rendezvous on 127.0.0.1 (train.py:553-557 does the same with mp.spawn).  The children inherit the real stdout, on which
    rank 0 writes the one JSON line."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), DASAC_BENCH_SELF_LAUNCHED="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node={}".format(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env, stdout=json_fd, stderr=2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=769)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--groups", type=int, default=2)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle sample (and with it parity_fullres)")
    ap.add_argument("--no-kernel-table", action="store_true", help="skip the second, event-instrumented pass")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short cfg-2 / cfg-5 runs appended to the N = 1 line")
    ap.add_argument("--profile-steps", type=int, default=3, help="steps of the instrumented pass (kernels / roofline)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3"],
                    help="arithmetic of the forward/data-gradient GEMMs: exact fp32 MFMA (default, the reference's arithmetic) or the "
                         "opt-in split-bf16 path (3 bf16 MFMAs per product, fp32 accumulate)")
    ap.add_argument("--fused", action="store_true",
                    help="time the FUSED student schedule as the headline (SAC.forward_fused: source + target crops in one student pass and "
                         "one backward -- the same gradient sum, DESIGN 5).  Default since round 6: the module API exactly as the reference's "
                         "train.py calls it (net(image, masks) + backward, net(frames1, ..., use_teacher=True) + backward, train.py:128-133,"
                         "219-233); the fused schedule is then reported next to it as `value_fused_schedule`")
    ap.add_argument("--two-pass", action="store_true", help="(default since round 6; kept for old command lines)")
    ap.add_argument("--alt", action="store_true", help="also run the steps once in the other precision (reported as \"alt\", no credit)")
    ap.add_argument("--config", default="cfg3", choices=["cfg2", "cfg3", "cfg5"],
                    help="cfg3 (default, the headline): RN101+SAC 8+2x4 crops @769^2; cfg2: RN101 baseline/AdaBN step, 2 source + 2 "
                         "target crops @769^2, train-mode BN; cfg5: VGG16-FCN8s + SAC @512x1024 (per-GPU 8 + 2x4 crops)")
    args = ap.parse_args()

    # stdout carries exactly ONE line (the JSON).  Libraries print banners on file descriptor 1 behind Python's back (RCCL:
    # "RCCL version : ..." at communicator teardown), so fd 1 itself is pointed at stderr for the whole run and the JSON
    # line is written to the saved descriptor at the very end.
    json_fd = os.dup(1)
    os.dup2(2, 1)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, json_fd))

    arch, baseline, hw = "deeplabv2_resnet101", False, (args.size, args.size)
    if args.config == "cfg2":
        baseline, args.batch, args.groups, args.views = True, 2, 2, 1
    elif args.config == "cfg5":
        arch, hw = "fcn_vgg16_bn", (512, 1024)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "WORLD_SIZE={} but --gpus {}".format(world, args.gpus)
    # DASAC_BENCH_RANKS_PER_GPU=r: r ranks share a device (a 1-GPU box can exercise the N > 1 path; RCCL refuses two ranks
    # on one device, so the transport is gloo then).  Default: one rank per GPU over RCCL ("nccl" on ROCm).
    per_gpu = max(1, int(os.environ.get("DASAC_BENCH_RANKS_PER_GPU", "1")))
    backend = os.environ.get("DASAC_BENCH_BACKEND", "nccl" if per_gpu == 1 else "gloo")
    dev_index = local // per_gpu
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    force_ddp = world == 1 and os.environ.get("DASAC_BENCH_DDP") == "1"     # measure the wrapper's own cost on one GPU
    if world > 1 or force_ddp:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if force_ddp:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            dist.init_process_group(backend, rank=0, world_size=1, **({"device_id": dev} if backend == "nccl" else {}))
        else:
            dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    import models
    import driver
    from dasac_hip import ops
    from dasac_hip import lib as L_
    from dasac_hip.parallel import OverlappedDataParallel

    cfg = model_cfg(arch, baseline)
    # stdout carries exactly ONE line (the JSON); the model constructors' progress prints go to stderr on rank 0
    sys.stdout = sys.stderr if rank == 0 else open(os.devnull, "w")
    net = models.get_model(cfg, dev_index, num_classes=19, criterion=nn.CrossEntropyLoss(**CRITERION))
    driver.init_synthetic_weights(net, seed=0)
    net.cuda(dev_index).train()
    if not baseline:
        net.running_conf.fill_(0.05)
    optim = driver.make_optimizer(net, cfg)
    wrapper = "none"
    step_net = net
    if world > 1 or force_ddp:
        wrapper = os.environ.get("DASAC_BENCH_WRAPPER", "overlapped")
        if wrapper == "ddp":          # the reference's own wrapper (train.py:104): reduction after the backward pass, bucket copies
            step_net = nn.parallel.DistributedDataParallel(net, device_ids=[dev_index])
        else:
            step_net = OverlappedDataParallel(net, device_ids=[dev_index], reduce_single_rank=force_ddp)
    src, tgt = driver.synthetic_batches(args.batch, args.groups, args.views, hw, dev, seed=rank)
    if arch != "deeplabv2_resnet101":
        driver.calibrate_classifier(net, src[0][:1])          # logits std ~3 whatever the backbone's feature scale
    src = (src[0], driver.self_consistent_labels(net, src[0]))

    schedule = {"fuse": bool(args.fused) and not args.two_pass}

    def step(i):
        if baseline:      # train.py:274-289: source fwd/bwd/step + no-grad train-mode target forward (AdaBN)
            l = driver.baseline_train_iteration(step_net, optim, src, tgt[0])
            return None, l, None
        tgt_i = (tgt[0], tgt[1].clone(), tgt[2], tgt[3], tgt[4])      # forward rewrites -1 -> 255 in place
        return driver.sac_train_iteration(step_net, optim, src, tgt_i, args.views, update_teacher=(i == 0),
                                          lr_target=cfg.LR_TARGET, fuse_passes=schedule["fuse"])

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(first, warmup, steps, instrumented):
        """`warmup` untimed steps, then exactly `steps` timed ones between fences; max over ranks."""
        for i in range(warmup):
            step(first + i)
        fence()
        if instrumented:
            ops.PROFILE.start()
        timed_wrapper = step_net if hasattr(step_net, "time_exposed_reduction") else None
        if timed_wrapper is not None:
            timed_wrapper.time_exposed_reduction(True)
        t0 = time.perf_counter()
        for i in range(steps):
            res = step(first + warmup + i)
        fence()
        dt_ = time.perf_counter() - t0
        prof_ = ops.PROFILE.stop() if instrumented else None
        # per rank: its own wall time over the K steps and the part of it its launch stream spent WAITING for gradient reductions
        # at the end of the backward passes (the exposed, un-overlapped share) -- the first multi-GPU run explains its own curve
        exposed = timed_wrapper.exposed_reduction_ms() if timed_wrapper is not None else 0.0
        if timed_wrapper is not None:
            timed_wrapper.time_exposed_reduction(False)
        per_rank[:] = [{"rank": rank, "ms_per_step": round(dt_ / steps * 1e3, 3), "exposed_reduce_ms_per_step": round(exposed / steps, 3)}]
        if world > 1:
            gathered = [None] * world
            dist.all_gather_object(gathered, per_rank[0])
            per_rank[:] = gathered
            tmax = torch.tensor([dt_], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_ = float(tmax)
        return dt_, prof_, res

    def kernel_table(prof_, steps):
        """GEMM kernels: achieved TFLOP/s; streaming kernels: achieved TB/s of ALGORITHMIC bytes (each operand once)."""
        out_ = {}
        for k, v in prof_.items():
            sec = max(v["seconds"], 1e-12)
            row = {"ms_per_step": round(v["seconds"] / steps * 1e3, 3), "launches_per_step": v["launches"] // steps}
            if v["flops"] > 0:
                row["tflops"] = round(v["flops"] / sec / 1e12, 2)
            if v.get("bytes", 0) > 0:
                row["algorithmic_TB_per_s"] = round(v["bytes"] / sec / 1e12, 3)
                row["algorithmic_MB_per_launch"] = round(v["bytes"] / max(v["launches"], 1) / 1e6, 1)
            out_[k] = row
        return out_

    def measure_with_device_views(first, steps):
        """The same K steps with SURVEY 8f next-1 INSIDE the fences: per step and target image ONE u8 crop (+ label, padding mask)
        crosses PCIe from pinned memory, `views.TargetViews.make` emits the L views on the device (flip / zoom / resize / blur /
        jitter / greyscale / normalise / mask: dasac_make_views + dasac_view_photometric, the reference's Pillow pipeline of
        dataloader_target.py:281-306) and `driver.prep_batch` slices them for this rank (train.py:157-209).  Returns
        (ms per step, ms per step of the upload + view kernels alone by an event pair on the launch stream)."""
        import views as V
        tv = V.TargetViews(hw, args.views, seed=0, blur=(.1, 2.), jitter=0.4, jitter_p=0.5, grey_p=0.2)
        mean = torch.tensor(V.MEAN).view(3, 1, 1)
        std = torch.tensor(V.STD).view(3, 1, 1)
        host = []
        for gi in range(args.groups):       # u8 crops whose normalised form looks like the synthetic frames the other legs use
            f = tgt[2][gi * args.views].detach().cpu()
            img = ((f * std + mean) * 255.0).round().clamp(0, 255).to(torch.uint8).contiguous().pin_memory()
            lab = tgt[1][gi * args.views].detach().cpu().clamp(0, 18).to(torch.uint8).contiguous().pin_memory()
            msk = torch.zeros(hw, dtype=torch.uint8)
            msk[:3] = 1
            host.append((img, lab, msk.pin_memory()))
        evs = []

        def step_v(i):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            parts = [tv.make(*(t.to(dev, non_blocking=True) for t in h)) for h in host]
            loaded = tuple(torch.stack([p[j] for p in parts], 0) for j in range(5))             # [N, L, ...] like the loader's batch
            tgt_i = tuple(driver.prep_batch(t, args.groups, args.views) for t in loaded)
            b.record()
            evs.append((a, b))
            return driver.sac_train_iteration(step_net, optim, src, tgt_i, args.views, update_teacher=False,
                                              lr_target=cfg.LR_TARGET, fuse_passes=schedule["fuse"])

        step_v(first)
        fence()
        evs.clear()
        t0 = time.perf_counter()
        for i in range(steps):
            step_v(first + 1 + i)
        fence()
        dt_ = time.perf_counter() - t0
        views_ms = sum(a.elapsed_time(b) for a, b in evs) / steps
        return dt_ / steps * 1e3, views_ms

    ops.set_precision(args.precision)
    per_rank = []
    # 1) the headline: K steps, nothing but the step itself inside the timed region
    dt, _, out = measure(0, args.warmup, args.steps, False)
    per_rank_headline = list(per_rank)
    done = args.warmup + args.steps
    # 2) the same workload once more with a HIP event pair around every instrumented launch (kernel table / roofline)
    prof, dt_prof, psteps = {}, None, max(1, args.profile_steps)
    if not args.no_kernel_table:
        dt_prof, prof, _ = measure(done, 0, psteps, True)
        done += psteps
    # 3) the other student schedule (two passes <-> one fused pass), a few steps, for the record
    other_sched = None
    if not baseline and wrapper != "ddp" and not args.no_kernel_table:
        schedule["fuse"] = not schedule["fuse"]
        try:
            dt3, _, _ = measure(done, 1, args.steps, False)
            other_sched = round(dt3 / args.steps * 1e3, 3)
        except Exception as exc:
            other_sched = repr(exc)[:200]
        schedule["fuse"] = not schedule["fuse"]
        done += 1 + args.steps
    # 3b) the headline schedule once more with the target views generated ON THE DEVICE inside the timed region (N = 1)
    with_views = None
    if world == 1 and not baseline and args.config == "cfg3" and not args.no_kernel_table:
        try:
            ms_v, ms_views_only = measure_with_device_views(done, args.steps)
            with_views = {"ms_per_step": round(ms_v, 3), "views_ms_per_step": round(ms_views_only, 3),
                          "per_step": "{} u8 crops {}x{} (+ label, mask) H2D from pinned memory, {} views each on the device (zoom / flip / "
                                      "blur / jitter p=.5 / greyscale p=.2), prep_batch".format(args.groups, hw[0], hw[1], args.views)}
        except Exception as exc:
            with_views = {"error": repr(exc)[:200]}
        done += 1 + args.steps
    # 3c) what the one host round trip of a target step costs: the same steps with thresholds / focal weights from the device
    #     (SAC.device_thresholds: within 1 ULP of the host's ATen values, not bit-equal to them -- the default stays the host path)
    dev_thr_ms = None
    if world == 1 and not baseline and not args.no_kernel_table and hasattr(net, "device_thresholds"):
        net.device_thresholds = True
        try:
            dt4, _, _ = measure(done, 1, args.steps, False)
            dev_thr_ms = round(dt4 / args.steps * 1e3, 3)
        except Exception as exc:
            dev_thr_ms = repr(exc)[:200]
        net.device_thresholds = False
        done += 1 + args.steps
    fused_now = bool(schedule["fuse"]) and not baseline and wrapper != "ddp" and \
        net.backbone._batch_fits(args.batch + args.groups * args.views, hw[0], hw[1])     # else the driver runs the two passes
    alt = None
    if args.alt and not baseline:
        # the same K steps once more in the other arithmetic (outside the contract's timed region; reported as "alt")
        other = "bf16x3" if args.precision == "fp32" else "fp32"
        ops.set_precision(other)
        try:
            dt2, _, _ = measure(done, 1, args.steps, False)
            alt = {"dtype": other, "value": round(world * args.batch * args.steps / dt2, 4), "unit": "images/sec",
                   "ms_per_step": round(dt2 / args.steps * 1e3, 3),
                   "note": "opt-in arithmetic, NOT the reference's; carries no roofline or throughput claim (DESIGN 5b)"}
        except Exception as exc:      # the extra leg must never cost the contract's line
            alt = {"dtype": other, "error": repr(exc)[:200]}
        ops.set_precision(args.precision)
    losses = {k: float(v.detach().mean()) for k, v in out[1].items()}
    labelled = float((out[2]["teacher_labels"] != 255).float().mean()) if out[2] is not None else 0.0

    if rank == 0:
        gemm = {k: v for k, v in prof.items() if k.startswith("conv_gemm")}
        dom_name = max(gemm, key=lambda k: gemm[k]["seconds"]) if gemm else "conv_gemm<tile-per-block>"
        dom = gemm.get(dom_name, {"flops": 0.0, "seconds": 1.0, "launches": 0})
        ach = dom["flops"] / max(dom["seconds"], 1e-12) / 1e12
        peak = PEAK_FP32_MFMA_TFLOPS if args.precision == "fp32" else PEAK_BF16_MFMA_TFLOPS / 3.0
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")     # PMC-measured HBM bytes per launch (separate rocprofv3 passes)
        if os.path.isfile(tfile):
            blob = json.load(open(tfile))
            # the counters were taken on a particular version of the kernels: tools/hbm_traffic.py stamps the file with the hash of
            # csrc/, and a stale file is reported as such instead of silently describing kernels that no longer exist
            if blob.get("csrc_sha1") == csrc_sha1():
                traffic = blob.get(dom_name)
            else:
                traffic = {"stale": "profiles/traffic.json was measured on csrc {} but this tree is {}: re-run tools/profile_round.sh".format(
                    str(blob.get("csrc_sha1"))[:12], csrc_sha1()[:12])}
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
                       "parallelism": "dp{}".format(world),
                       "student_schedule": ("fused: source + target crops in ONE student pass and ONE backward over loss_ce + LR_TARGET*self_ce "
                                            "(same weights, frozen BN, teacher independent: the gradient sum of train.py:266-298's two passes)"
                                            if fused_now else "two passes: the module API as train.py calls it (train.py:128-133,219-233)"),
                       "distributed": {"world_size": dist.get_world_size() if dist.is_initialized() else 1,
                                       "backend": dist.get_backend() if dist.is_initialized() else None, "wrapper": wrapper,
                                       "ranks_per_gpu": per_gpu, "self_launched": os.environ.get("DASAC_BENCH_SELF_LAUNCHED") == "1",
                                       "reserved_cus": L_.load().dasac_reserved_cus(), "per_rank": per_rank_headline},
                       # deviations from SURVEY 8d's synthetic recipe (same arithmetic work; the reference's SGD hyper-parameters
                       # diverge within a few steps on N(0,0.01) weights with random labels):
                       "synthetic_recipe": "weights He-normal (not N(0,.01)), BN gamma~U(.5,1.5)*{1,.1 closing a residual branch,.3 shortcut}, "
                                           "beta/mean~N(0,.1), var~U(.5,1.5), classifier x6 (not x50); source labels = the initial net's own "
                                           "argmax with a 16-px ignore border (not randint); crops N(0,1); views identity|zoom.7+shift+flip|"
                                           "zoom.5+shift|flip; running_conf pre-seeded 0.05; teacher update on step 0 only"},
            "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                         "frac": round(ach / peak, 4), "traffic": traffic,
                         "kernel": "dasac::" + dom_name + " (forward + data-gradient implicit GEMM, {})".format(
                             "fp32 MFMA" if args.precision == "fp32" else "3 bf16 MFMAs per fp32 product: peak = bf16 peak / 3"),
                         "algorithmic_gflop_per_launch": round(dom["flops"] / max(dom["launches"], 1) / 1e9, 2),
                         "algorithmic_MB_per_launch": round(dom.get("bytes", 0.0) / max(dom["launches"], 1) / 1e6, 1),
                         "launches": dom["launches"], "avg_launch_ms": round(dom["seconds"] / max(dom["launches"], 1) * 1e3, 4),
                         "measured_in": "second pass of {} steps with a HIP event pair around each launch (not the headline's timed region)".format(psteps)},
            "ms_per_step_instrumented": None if dt_prof is None else round(dt_prof / psteps * 1e3, 3),
            "ms_per_step_other_schedule": {("two_pass" if fused_now else "fused"): other_sched},
            # `value` is the module API exactly as train.py calls it unless --fused was given; both figures are always in the line:
            # value_train_py_api = net(image, masks) + backward, then net(frames1, ..., use_teacher=True) + backward (train.py:128-133,
            # 219-233); value_fused_schedule = SAC.forward_fused, one student pass + one backward over the same gradient sum
            "value_train_py_api": (round(world * args.batch / other_sched * 1e3, 4) if isinstance(other_sched, float) else None)
            if fused_now else round(world * args.batch * args.steps / dt, 4),
            "value_fused_schedule": round(world * args.batch * args.steps / dt, 4) if fused_now
            else (round(world * args.batch / other_sched * 1e3, 4) if isinstance(other_sched, float) else None),
            "ms_per_step_device_thresholds": dev_thr_ms,
            "ms_per_step_with_device_views": None if with_views is None else with_views.get("ms_per_step"),
            "device_views": with_views,
            "kernels": kernel_table(prof, psteps),
            "check": {"loss_ce": losses.get("loss_ce"), "self_ce": losses.get("self_ce"), "teacher_diff": losses.get("teacher_diff"),
                      "labelled_frac": round(labelled, 4)},
        }
        if alt is not None:
            line["alt"] = alt
        if world == 1 and not args.no_cpu_baseline and args.config == "cfg3":
            # free the benchmark model first: the parity sample builds its own
            del step_net, net, optim, out
            torch.cuda.empty_cache()
            try:
                sample = cpu_sample(args.size)
                line["cpu_baseline"] = cpu_baseline_entry(args.size, sample[0], sample[1])
            except Exception as exc:      # never lose the line to the reported baseline
                sample = None
                line["cpu_baseline"] = {"value": None, "unit": "images/sec", "cores": 0, "kind": "port", "sample": "failed: " + repr(exc)[:160]}
            if sample is not None:
                try:
                    line["parity_fullres"] = parity_fullres(args.size, dev, sample, fuse=fused_now)[1]
                except Exception as exc:
                    line["parity_fullres"] = {"error": repr(exc)[:200]}
            del sample
            torch.cuda.empty_cache()
            try:      # the multi-view head at full resolution with 8 crops / 4 views per group (head only: seconds of oracle time)
                hdt, hcmp = head_parity_fullres(args.size, args.groups, args.views, dev)
                line["parity_fullres_head"] = dict(hcmp, oracle_seconds=round(hdt, 1))
            except Exception as exc:
                line["parity_fullres_head"] = {"error": repr(exc)[:200]}
            try:      # the reference's Pillow view pipeline for the same images, one host thread (the loader bottleneck next-1 removes)
                one = pillow_views_ms(hw, args.views)
                line["cpu_views_pillow"] = {"ms_per_target_image": round(one, 1), "ms_per_step": round(one * args.groups, 1), "threads": 1,
                                            "what": "Pillow {}: per view flip / crop+resize (image BILINEAR, label+mask NEAREST) / GaussianBlur / "
                                                    "jitter / greyscale / ToTensor+Normalize x2, {} views of one {}x{} crop".format(
                                                        __import__("PIL").__version__, args.views, hw[0], hw[1])}
            except Exception as exc:
                line["cpu_views_pillow"] = {"error": repr(exc)[:160]}
            if not args.no_other_configs:
                line["other_configs"] = {}
                for name in ("cfg2", "cfg5"):
                    torch.cuda.empty_cache()
                    try:
                        line["other_configs"][name] = time_other_config(name, dev)
                    except Exception as exc:
                        line["other_configs"][name] = {"error": repr(exc)[:200]}
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if world > 1 or force_ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
