"""Full-resolution end-to-end parity.  cfg-3 (769x769 crops): the sample bench.py times the CPU oracle on -- 1 source + 1
target crop, student forward/backward each, teacher forward, SAC head, SGD step -- runs through the HIP module too and the
two are compared (losses, label map, class prior, sampled gradients and updated parameters), under both student schedules;
cfg-2 (baseline / AdaBN iteration with batch-statistics BN at 769x769, gradients arbitrated by a float64 oracle run) and cfg-5
(VGG16-FCN8s + SAC at 512x1024) likewise at their own resolutions; and `bench.py --gpus 2` launches its own ranks (two on the one
GPU of the box) and prints the contract's JSON line.  The CPU oracle needs 10-60 s per case on the box's host cores."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


_SAMPLE = {}


@pytest.mark.parametrize("fuse", [False, True])
def test_cfg3_resolution_sample_matches_the_oracle(fuse):
    """fuse: the student's source and target passes as one (what bench.py times) or one after the other (train.py:266-298)."""
    sys.path.insert(0, ROOT)
    import bench
    if "s" not in _SAMPLE:
        _SAMPLE["s"] = bench.cpu_sample(769)                  # the oracle's 14 s once for both schedules
    (_, cores), cmp_ = bench.parity_fullres(769, sample=_SAMPLE["s"], fuse=fuse)
    print("parity_fullres:", cmp_, "cores", cores)
    assert cmp_["labelled_frac"] > 0.05                       # the thresholds fire: the label comparison is not vacuous
    assert cmp_["loss_ce_rel"] <= 1e-4
    # bounds = at most 5x what five rounds of boxes measured (self_ce 6.5e-5, 2 of 591 361 labels = 3.4e-6, gradients 2.6e-4 of the
    # tensor max): a regression of one order of magnitude fails here
    assert cmp_["self_ce_rel"] <= 5e-4
    assert cmp_["label_mismatch_frac"] <= 2e-5                # fp32 probabilities differ at 1e-6: borderline pixels only
    assert cmp_["running_conf_max_abs"] <= 1e-6
    # north_star: logits / grads within 1e-3 rel (of the tensor max); free-running gradients carry the borderline-ReLU effect
    # that the fp64 arbitration of test_gpu_models.py explains -- at 591k pixels per plane one flipped unit weighs far less
    assert cmp_["grad_max_err_over_tensor_max"] <= 1e-3
    assert cmp_["param_max_err_over_tensor_max"] <= 1e-5


def test_cfg3_resolution_sample_is_bit_reproducible():
    """Round 4: no floating-point atomics are left on the path, the stream-K ranges and the weight-gradient splits are functions
    of the shape alone -- two runs of the full-resolution sample (769 x 769: student forward / backward, teacher, SAC head, SGD
    step) from the same state end in the SAME bits: losses, label map, class prior, gradients, parameters."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    if "s" not in _SAMPLE:
        _SAMPLE["s"] = bench.cpu_sample(769)
    sd = _SAMPLE["s"][2]
    dev = torch.device("cuda", 0)
    a, b = bench.hip_sample(769, sd, dev, fuse=True), bench.hip_sample(769, sd, dev, fuse=True)
    assert a["loss_ce"] == b["loss_ce"] and a["self_ce"] == b["self_ce"] and a["teacher_diff"] == b["teacher_diff"]
    assert torch.equal(a["labels"], b["labels"]) and torch.equal(a["running_conf"], b["running_conf"])
    for k in a["grads"]:
        assert torch.equal(a["grads"][k], b["grads"][k]), k
        assert torch.equal(a["params"][k], b["params"][k]), k


def test_full_resolution_head_parity_eight_crops_four_views():
    """VERDICT r3 missing 4: the multi-view head at cfg-3's size with B = 8 crops and L = 4 views per group -- the T = 4 fusion
    of models/sac.py:238-269,289-311 and the [B,B,H,W] broadcast of :148 -- against oracle.head_ref at 769 x 769 (head only, so
    the oracle takes seconds): refined probabilities and the warped diagnostics, class prior, the label map with torch.equal on
    equal probabilities through the module's threshold path, loss value and d loss / d (stride-8 logits)."""
    sys.path.insert(0, ROOT)
    import bench
    dt, c = bench.head_parity_fullres(769, groups=2, views=4)
    print("head_parity_fullres:", c, "oracle seconds", round(dt, 1))
    assert c["crops"] == 8 and c["views_per_group"] == 4 and c["labelled_frac"] > 0.3
    assert c["refined_max_abs"] <= 1e-5 and c["teacher_aligned_max_abs"] <= 1e-5 and c["frames_aligned_rel"] <= 1e-5
    assert c["running_conf_max_abs"] <= 1e-7
    assert c["labels_equal_on_equal_probs"] and c["conf_equal_on_equal_probs"]          # the bit-exact contract
    assert c["label_mismatch_frac_end_to_end"] < 1e-4                                   # probabilities differ at 1e-6: borderline pixels
    assert c["self_ce_rel"] <= 1e-5 and c["self_ce_rel_end_to_end"] <= 1e-3
    assert c["dlogits_max_err_over_tensor_max"] <= 1e-4


@pytest.mark.parametrize("ranks,size", [(2, 129), (8, 65)])
def test_bench_launches_its_own_ranks(ranks, size):
    """`python bench.py --gpus N` without torch.distributed.run (what the driver runs for the scaling curve): N ranks, here
    all on the single device of the box (DASAC_BENCH_RANKS_PER_GPU=N -> gloo transport), tiny crops; exactly one JSON line
    on stdout with n_gpus = N, the process group's world size and one `per_rank` entry per rank (N = 8: the driver's own
    8-GPU command)."""
    env = dict(os.environ, DASAC_BENCH_RANKS_PER_GPU=str(ranks))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--size", str(size),
           "--batch", "2", "--groups", "1", "--views", "2", "--profile-steps", "1"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == ranks and line["value"] > 0 and line["steps"] == 2
    d = line["config"]["distributed"]
    assert d["world_size"] == ranks and d["self_launched"] and d["wrapper"] == "overlapped" and d["ranks_per_gpu"] == ranks
    assert sorted(e["rank"] for e in d["per_rank"]) == list(range(ranks))
    assert all(e["ms_per_step"] > 0 for e in d["per_rank"])
    assert line["config"]["global_batch"] == 2 * ranks
    assert "roofline" in line and line["kernels"]


# -------------------------------------------------------------------------------------------------------------------------
# cfg-2 and cfg-5 at THEIR resolutions (VERDICT r2 "configs exercised only in reduced form"): a bounded sample of each through
# the HIP module and the CPU oracle, weights generated once (driver.init_synthetic_weights) and handed to both sides.
# -------------------------------------------------------------------------------------------------------------------------
def _probe(sd, keys):
    return {k: sd[k].detach().float().cpu().clone() for k in keys}


def _tmax(a, b):
    import torch
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def test_cfg2_resolution_baseline_step_matches_the_oracle():
    """cfg-2: ResNet-101 DeepLabv2 baseline / AdaBN iteration (train.py:274-289) at 769x769 with batch-statistics BN: 2 source
    crops forward + backward + SGD step, then the no-grad train-mode forward of 2 target crops that only moves the BN running
    statistics.  Loss, updated parameters and the running statistics of first / middle / last BN layers against the oracle."""
    import torch
    import torch.nn as nn
    sys.path.insert(0, ROOT)
    import bench
    import driver
    import models
    from oracle.step_ref import SacOracle, SgdOracle, baseline_train_iteration, DEFAULT_CFG
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 32)))
    cfg = bench.model_cfg("deeplabv2_resnet101", baseline=True)
    net = models.get_model(cfg, 0, num_classes=19, criterion=nn.CrossEntropyLoss(ignore_index=255, reduction="none"))
    driver.init_synthetic_weights(net, seed=2)
    sd = {k: v.detach().clone() for k, v in net.backbone.state_dict().items()}
    net.cuda().train()
    assert any(m.training for m in net.backbone.modules() if isinstance(m, (nn.BatchNorm2d, nn.SyncBatchNorm)))
    optim = driver.make_optimizer(net, cfg)
    src, tgt = driver.synthetic_batches(2, 2, 1, (769, 769), "cpu", seed=4)
    ref = SacOracle(sd, cfg=dict(DEFAULT_CFG, BASELINE=True))
    l_ref = baseline_train_iteration(ref, SgdOracle(ref), src, tgt[0])
    l_hip = driver.baseline_train_iteration(net, optim, tuple(t.cuda() for t in src), tgt[0].cuda())
    torch.cuda.synchronize()
    got = net.backbone.state_dict()
    assert abs(float(l_hip["loss_ce"]) - l_ref["loss_ce"]) <= 1e-4 * abs(l_ref["loss_ce"])
    worst = {}
    for k in ("model.conv1.weight", "model.layer1.0.conv1.weight", "model.layer2.3.bn2.weight", "model.layer3.10.conv2.weight",
              "model.layer4.2.bn3.bias", "model.layer5.conv2d_list.0.weight", "model.layer5.conv2d_list.3.bias"):
        worst[k] = _tmax(got[k].cpu(), ref.student[k].detach())
    for k in ("model.bn1.running_mean", "model.bn1.running_var", "model.layer3.22.bn3.running_mean", "model.layer3.22.bn3.running_var",
              "model.layer4.2.bn2.running_var"):
        worst[k] = _tmax(got[k].cpu(), ref.student[k].detach())
    named = dict(net.backbone.named_parameters())
    gworst = {k: _tmax(named[k].grad.cpu(), ref.student[k].grad) for k in ("model.conv1.weight", "model.layer3.10.conv2.weight",
                                                                           "model.layer4.2.bn3.weight", "model.layer5.conv2d_list.0.weight")}
    print("cfg-2 @769: loss", float(l_hip["loss_ce"]), l_ref["loss_ce"], worst, gworst)
    assert max(worst.values()) <= 1e-4, worst
    # Gradients below a batch-statistics BN: the two fp32 implementations differ by ~1e-2 of the tensor max in the early layers
    # (the classifier, above every BN backward, agrees to 3e-6).  Who is right is decided by a float64 run of the oracle, as in
    # tests/test_gpu_models.py::test_resnet101_gradients_fp64_arbitration: the HIP path (float64 BN sums, fp32 everything else)
    # must be no further from the fp64 gradients than ATen's own fp32 is.
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    ref64 = SacOracle(sd64, cfg=dict(DEFAULT_CFG, BASELINE=True))
    baseline_train_iteration(ref64, SgdOracle(ref64), (src[0].double(), src[1]), tgt[0].double())
    e_hip = {k: _tmax(named[k].grad.cpu(), ref64.student[k].grad) for k in gworst}
    e_aten = {k: _tmax(ref.student[k].grad, ref64.student[k].grad) for k in gworst}
    print("cfg-2 @769 gradients vs fp64: HIP", e_hip, "ATen fp32", e_aten)
    for k in gworst:
        assert e_hip[k] <= 2.0 * e_aten[k] + 1e-4, (k, e_hip[k], e_aten[k])
    assert int(got["model.bn1.num_batches_tracked"]) == int(ref.student["model.bn1.num_batches_tracked"]) == 2


@pytest.mark.parametrize("fuse", [False, True])
def test_cfg5_resolution_sample_matches_the_oracle(fuse):
    """cfg-5: VGG16-FCN8s + SAC at 512x1024 (1 source + 1 target crop, L = 1; Dropout2d p forced to 0 as SURVEY 8d prescribes for
    parity runs): losses, label map, class prior and parameters after the SGD step against the oracle."""
    import torch
    import torch.nn as nn
    sys.path.insert(0, ROOT)
    import bench
    import driver
    import models
    from oracle.step_ref import SacOracle, SgdOracle, sac_train_iteration, DEFAULT_CFG
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 32)))
    cfg = bench.model_cfg("fcn_vgg16_bn")
    net = models.get_model(cfg, 0, num_classes=19, criterion=nn.CrossEntropyLoss(ignore_index=255, reduction="none"))
    driver.init_synthetic_weights(net, seed=3)
    for m in net.modules():
        if isinstance(m, nn.Dropout2d):
            m.p = 0.0
    net.cuda().train()
    src, tgt = driver.synthetic_batches(1, 1, 1, (512, 1024), "cpu", seed=6)
    driver.calibrate_classifier(net, src[0].cuda())
    sd = {k: v.detach().cpu().clone() for k, v in net.backbone.state_dict().items()}
    net.slow_net.load_state_dict(sd)
    net.running_conf.fill_(0.05)
    net.slow_init[0] = 1.0
    optim = driver.make_optimizer(net, cfg)
    ref = SacOracle(sd, cfg=dict(DEFAULT_CFG, ARCH="fcn_vgg16_bn", LR=cfg.LR, LR_TARGET=cfg.LR_TARGET))
    ref.running_conf.fill_(0.05)
    ref.slow_init[0] = 1.0
    ls_r, lt_r, o_r = sac_train_iteration(ref, SgdOracle(ref), src, tuple(t.clone() for t in tgt), 1, update_teacher=False)
    to = lambda ts: tuple(t.cuda() for t in ts)
    ls, lt, o = driver.sac_train_iteration(net, optim, to(src), to(tgt), 1, False, cfg.LR_TARGET, fuse_passes=fuse)
    torch.cuda.synchronize()
    labelled = float((o_r["teacher_labels"] != 255).float().mean())
    mism = float((o["teacher_labels"].cpu() != o_r["teacher_labels"]).float().mean())
    got = net.backbone.state_dict()
    worst = {k: _tmax(got[k].cpu(), ref.student[k].detach())
             for k in ("block1.0.weight", "block2.27.weight", "block3.40.weight", "vgg_head.0.weight", "vgg_head.1.weight", "vgg_head.4.weight",
                       "vgg_head.8.weight", "vgg_head.8.bias", "score_pool4.weight", "score_pool3.bias")}
    print("cfg-5 @512x1024 fuse", fuse, "loss_ce", float(ls["loss_ce"]), ls_r["loss_ce"], "self_ce", float(lt["self_ce"]), lt_r["self_ce"],
          "labelled", labelled, "mismatch", mism, worst)
    assert labelled > 0.01
    assert abs(float(ls["loss_ce"]) - ls_r["loss_ce"]) <= 1e-4 * abs(ls_r["loss_ce"])
    assert abs(float(lt["self_ce"]) - lt_r["self_ce"]) <= 5e-3 * abs(lt_r["self_ce"]) + 1e-7
    assert mism < 1e-3
    assert float((net.running_conf.cpu() - ref.running_conf).abs().max()) <= 1e-6
    assert max(worst.values()) <= 1e-4, worst


@pytest.mark.gpu
def test_cfg5_full_batch_fused_student_pass_equals_the_two_passes():
    """cfg-5 at its FULL per-GPU size (VGG16-FCN8s + SAC, 8 source + 2 x 4 target crops at 512x1024): since round 5 the conv
    kernels address up to 4 GiB per tensor, so the student's two passes run as ONE 16-crop pass whose first activations are exactly
    2 GiB (models/fcn.py:136-149, train.py:266-298).  One iteration from the same state through both schedules (Dropout2d p = 0):
    same losses, same label map, parameters after the SGD step within 1e-4 of their max (run-vs-run across schedules: the
    weight-gradient reductions sum in another order)."""
    import copy
    import torch
    import torch.nn as nn
    sys.path.insert(0, ROOT)
    import bench
    import driver
    import models
    cfg = bench.model_cfg("fcn_vgg16_bn")
    dev = torch.device("cuda", 0)
    net = models.get_model(cfg, 0, num_classes=19, criterion=nn.CrossEntropyLoss(ignore_index=255, reduction="none"))
    driver.init_synthetic_weights(net, seed=0)
    net.cuda(0).train()
    for m in net.modules():
        if isinstance(m, nn.Dropout2d):
            m.p = 0.0
    net.running_conf.fill_(0.05)
    src, tgt = driver.synthetic_batches(8, 2, 4, (512, 1024), dev, seed=0)
    driver.calibrate_classifier(net, src[0][:1])
    src = (src[0], driver.self_consistent_labels(net, src[0]))
    assert net.backbone._batch_fits(16, 512, 1024), "the fused pass must fit the addressing window"
    state = copy.deepcopy(net.state_dict())
    results = {}
    for fuse in (True, False):
        net.load_state_dict(state)
        optim = driver.make_optimizer(net, cfg)
        tgt_i = (tgt[0], tgt[1].clone(), tgt[2], tgt[3], tgt[4])
        ls, lt, outs = driver.sac_train_iteration(net, optim, src, tgt_i, 4, update_teacher=True, lr_target=cfg.LR_TARGET, fuse_passes=fuse)
        torch.cuda.synchronize()
        results[fuse] = (float(ls["loss_ce"]), float(lt["self_ce"]), outs["teacher_labels"].clone(),
                         {k: v.detach().clone() for k, v in net.backbone.state_dict().items() if v.dtype.is_floating_point})
    (la, sa, laba, pa), (lb, sb, labb, pb) = results[True], results[False]
    assert abs(la - lb) <= 1e-5 * abs(lb) and abs(sa - sb) <= 1e-4 * max(abs(sb), 1e-6), (la, lb, sa, sb)
    assert torch.equal(laba, labb)                      # the teacher does not depend on the student schedule
    worst = max(float((pa[k] - pb[k]).abs().max() / pb[k].abs().max().clamp_min(1e-30)) for k in pb)
    assert worst <= 1e-4, worst


def test_split_k_tail_schedule_inside_the_network_equals_one_block_per_tile():
    """Round 6: at cfg-3's per-pass size (8 crops of 769 x 769) every long-K layer of ResNet-101 runs as ONE launch of tile-per-block
    rounds + split-K tail pieces.  The whole student pass (forward, loss, backward) under the library's schedule against the same pass
    with every GEMM forced to one block per tile: the same arithmetic up to the summation order of the cut tiles -- logits and loss to
    1e-5 / 1e-6, gradients to 1e-4 of the tensor max for the median tensor (single tensors upstream of a ReLU unit whose pre-activation
    is within rounding of zero move more: the effect tests/test_gpu_models.py arbitrates in float64)."""
    import torch
    import torch.nn as nn
    sys.path.insert(0, ROOT)
    import bench
    import driver
    import models
    from dasac_hip import ops
    cfg = bench.model_cfg()
    net = models.get_model(cfg, 0, num_classes=19, criterion=nn.CrossEntropyLoss(ignore_index=255, reduction="none"))
    driver.init_synthetic_weights(net, seed=0)
    net.cuda().train()
    (x, y), _ = driver.synthetic_batches(8, 1, 1, (769, 769), "cuda", seed=0)
    real = ops.conv_gemm

    def run(force_tile_per_block):
        if force_tile_per_block:
            ops.conv_gemm = lambda *a, **k: real(*a, **dict(k, schedule=1))
        try:
            ops.PROFILE.start()
            losses, outs = net.backbone(x, y)
            for p in net.backbone.parameters():
                p.grad = None
            losses["loss_ce"].mean().backward()
            prof = ops.PROFILE.stop()
        finally:
            ops.conv_gemm = real
        torch.cuda.synchronize()
        return (outs["logits"].detach().clone(), float(losses["loss_ce"]),
                {n: p.grad.detach().clone() for n, p in net.backbone.named_parameters() if p.grad is not None}, prof)

    lg_a, loss_a, g_a, prof_a = run(False)
    lg_b, loss_b, g_b, prof_b = run(True)
    assert prof_a["conv_gemm<tile+tail>"]["launches"] >= 100 and "conv_gemm<tile+tail>" not in prof_b
    assert _tmax(lg_a, lg_b) < 1e-5 and abs(loss_a - loss_b) <= 1e-6 * abs(loss_b)
    errs = sorted(_tmax(g_a[k], g_b[k]) for k in g_a)
    print("tail vs tile-per-block, 320 gradients: median %.2e, 95%% %.2e, max %.2e" % (errs[len(errs) // 2], errs[int(len(errs) * 0.95)], errs[-1]))
    assert len(errs) == 320 and errs[len(errs) // 2] < 5e-5 and errs[-1] < 1e-3         # measured: median 8.3e-6, max 7.9e-5
