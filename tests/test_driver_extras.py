"""Checkpoint layout (CPU) and the validation-count kernel (GPU) -- SURVEY.md 8(f) next-2 / next-3."""
import os
import tempfile
from types import SimpleNamespace as NS

import pytest
import torch
import torch.nn as nn

from oracle.step_ref import DEFAULT_CFG

CRIT = nn.CrossEntropyLoss(ignore_index=255, reduction="none")


def test_checkpoint_roundtrip_and_baseline_into_sac():
    import models
    import driver
    cfg_b = NS(**dict(DEFAULT_CFG, INIT_MODEL="", BASELINE=True))
    cfg_s = NS(**dict(DEFAULT_CFG, INIT_MODEL="", BASELINE=False))
    base = models.get_model(cfg_b, 0, num_classes=19, criterion=CRIT)
    with torch.no_grad():
        base.backbone.model.conv1.weight.fill_(0.123)
    opt = torch.optim.SGD(base.parameter_groups(1e-3, 5e-4), momentum=0.9)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "epoch003_score0.412.pth")          # utils/checkpoints.py:63 naming
        driver.save_checkpoint(path, base, opt, 0.412, 3)
        blob = torch.load(path)
        assert set(blob) == {"model", "opt", "score", "epoch"} and all(k.startswith("module.backbone.") for k in blob["model"])
        sac = models.get_model(cfg_s, 0, num_classes=19, criterion=CRIT)
        epoch, score, missing, unexpected = driver.load_checkpoint(path, sac)
        assert (epoch, score) == (3, 0.412) and not unexpected
        assert all(k.startswith(("slow_net.", "running_conf", "slow_init")) for k in missing)
        assert float(sac.backbone.model.conv1.weight[0, 0, 0, 0]) == pytest.approx(0.123)
        assert float(sac.slow_init[0]) == 0.0


@pytest.mark.gpu
def test_iou_counts_kernel_matches_reference_definition():
    from dasac_hip import ops
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(3, 19, 37, 41, generator=g)
    gt = torch.randint(0, 19, (3, 37, 41), generator=g)
    gt[torch.rand(3, 37, 41, generator=g) < 0.2] = 255
    counts = ops.iou_counts(logits.cuda(), gt.cuda())
    counts = ops.iou_counts(logits.cuda(), gt.cuda(), counts).cpu()          # accumulation over batches
    pred = logits.argmax(1)
    valid = gt != 255
    for c in range(19):                                                      # utils/metrics.py:25-45
        tp = int(((pred == c) & (gt == c) & valid).sum())
        fp = int(((pred == c) & (gt != c) & valid).sum())
        fn = int(((pred != c) & (gt == c) & valid).sum())
        assert counts[:, c].tolist() == [2 * tp, 2 * fp, 2 * fn]


def test_summarise_iou_matches_reference_golden_g14(golden):
    """utils/metrics.py:40-53 on the reference's own accumulated counts (CPU: host arithmetic only)."""
    import driver
    g = golden("g14_jaccard")
    j, p, r = driver.summarise_iou(torch.from_numpy(g["counts2"]))
    assert torch.equal(j, torch.from_numpy(g["jaccards"])) and torch.equal(p, torch.from_numpy(g["precision"]))
    assert torch.equal(r, torch.from_numpy(g["recall"]))
    assert float(j[17]) == 0.0 and float(j[18]) == 0.0            # never occurring / predicted but never true


@pytest.mark.gpu
def test_iou_counts_kernel_matches_reference_golden_g14(golden):
    """The reference's `Jaccard.add_sample` (utils/metrics.py:19-38) run over three batches -- ignore pixels, a fully ignored
    image, gt = -1 pixels, an absent class, a never-true class -- against the one-pass count kernel: the integer counts
    after EVERY batch and the summary are equal."""
    import driver
    from dasac_hip import ops
    g = golden("g14_jaccard")
    counts = None
    for b in range(3):
        counts = ops.iou_counts(torch.from_numpy(g["logits%d" % b]).cuda(), torch.from_numpy(g["gt%d" % b]).cuda(), counts)
        assert torch.equal(counts.cpu(), torch.from_numpy(g["counts%d" % b])), b
    j, p, r = driver.summarise_iou(counts)
    assert torch.equal(j, torch.from_numpy(g["jaccards"])) and torch.equal(p, torch.from_numpy(g["precision"]))
    assert torch.equal(r, torch.from_numpy(g["recall"]))


@pytest.mark.gpu
def test_fused_inference_labels_match_oracle():
    """infer_val.py:160-163 + argmax + convert_to_cs as one kernel: label maps equal the oracle's except where the
    top-2 probabilities are within fp32 noise of each other."""
    import driver
    from oracle import head_ref as R
    from dasac_hip import ops
    g = torch.Generator().manual_seed(11)
    logits = torch.randn(2, 19, 23, 31, generator=g) * 3
    lut = torch.tensor(driver.CITYSCAPES_TRAIN_TO_ID, dtype=torch.uint8)
    lab, conf = ops.infer_labels(logits.cuda(), (177, 241), lut.cuda(), want_conf=True)
    ref_lab, ref_conf, gap = R.infer_labels(logits, 177, 241, lut)
    assert lab.dtype == torch.uint8 and lab.shape == (2, 177, 241)
    diff = lab.cpu() != ref_lab
    assert int(diff.sum()) == int((diff & (gap < 1e-5)).sum())      # only exact near-ties may differ
    assert float(diff.float().mean()) < 1e-4
    assert float((conf.cpu() - ref_conf).abs().max()) < 1e-5
    lab2, _ = ops.infer_labels(logits.cuda(), (177, 241))
    assert torch.equal(lut.cuda()[lab2.long()], lab)
    # a 5-class head goes through the generic instantiation
    l5 = torch.randn(1, 5, 9, 7, generator=g)
    lab5, _ = ops.infer_labels(l5.cuda(), (33, 29))
    r5, _, gap5 = R.infer_labels(l5, 33, 29)
    d5 = lab5.cpu() != r5
    assert int(d5.sum()) == int((d5 & (gap5 < 1e-5)).sum())


@pytest.mark.gpu
def test_infer_label_maps_through_the_model():
    import driver
    import models
    from types import SimpleNamespace as NS
    from oracle.step_ref import DEFAULT_CFG
    import torch.nn as nn
    d = dict(DEFAULT_CFG)
    d.update(INIT_MODEL="", OPT_NESTEROV=False)
    net = models.get_model(NS(**d), 0, num_classes=19, criterion=nn.CrossEntropyLoss(ignore_index=255, reduction="none")).cuda().eval()
    x = torch.randn(1, 3, 65, 81, device="cuda")
    lab, _ = driver.infer_label_maps(net, x, lut=driver.CITYSCAPES_TRAIN_TO_ID)
    with torch.no_grad():
        _, up = net(x, teacher=False)
    want = torch.tensor(driver.CITYSCAPES_TRAIN_TO_ID, device="cuda")[up.softmax(1).argmax(1)]
    assert float((lab.long() != want).float().mean()) < 1e-3


def test_make_optimizer_follows_get_optim():
    """base_trainer.py:47-73: SGD (fused or torch's), Adam with betas=(BETA1, .999), any other torch.optim name with lr only,
    NotImplementedError for names torch.optim does not have; the four parameter groups keep their own lr / weight decay."""
    import models
    import driver
    from dasac_hip.optim import FusedSGD
    cfg = NS(**dict(DEFAULT_CFG, INIT_MODEL="", OPT_NESTEROV=False))
    net = models.get_model(cfg, 0, num_classes=19, criterion=CRIT)
    net.train()
    sgd = driver.make_optimizer(net, cfg)
    assert isinstance(sgd, FusedSGD) and [g["lr"] for g in sgd.param_groups] == pytest.approx([2.5e-4, 5e-4, 2.5e-3, 5e-3])
    assert [g["weight_decay"] for g in sgd.param_groups] == [5e-4, 0.0, 5e-4, 0.0] and sgd.param_groups[0]["momentum"] == 0.9
    nes = driver.make_optimizer(net, NS(**dict(vars(cfg), OPT_NESTEROV=True)))
    assert type(nes) is torch.optim.SGD and nes.param_groups[0]["nesterov"]
    adam = driver.make_optimizer(net, NS(**dict(vars(cfg), OPT="Adam", BETA1=0.5)))
    assert type(adam) is torch.optim.Adam and adam.param_groups[0]["betas"] == (0.5, 0.999)
    assert [g["lr"] for g in adam.param_groups] == pytest.approx([2.5e-4, 5e-4, 2.5e-3, 5e-3])
    rms = driver.make_optimizer(net, NS(**dict(vars(cfg), OPT="RMSprop")))
    assert type(rms) is torch.optim.RMSprop
    with pytest.raises(NotImplementedError):
        driver.make_optimizer(net, NS(**dict(vars(cfg), OPT="NoSuchOptimiser")))


def test_train_epoch_policy(monkeypatch):
    """train.py:266-298: the teacher is refreshed on every NET_MOMENTUM_ITER-th iteration of an epoch, TARGET_ONLY is passed
    through, baseline mode takes the AdaBN path."""
    import driver
    calls = []
    monkeypatch.setattr(driver, "sac_train_iteration",
                        lambda net, optim, s, t, L, upd, lr_t, target_only=False: calls.append(("sac", s, t, L, upd, lr_t, target_only)))
    monkeypatch.setattr(driver, "baseline_train_iteration", lambda net, optim, s, t: calls.append(("base", s, t)))
    cfg = NS(**dict(DEFAULT_CFG, NET_MOMENTUM_ITER=3))
    src, tgt = [("s%d" % i,) for i in range(7)], [("t%d" % i,) for i in range(7)]
    seen = []
    n = driver.train_epoch(None, None, src, tgt, cfg, 4, target_only=True, on_iteration=lambda i, out: seen.append(i))
    assert n == 7 and seen == list(range(7))
    assert [c[4] for c in calls] == [True, False, False, True, False, False, True]
    assert all(c[0] == "sac" and c[3] == 4 and c[5] == 5.0 and c[6] is True for c in calls)
    calls.clear()
    driver.train_epoch(None, None, src[:2], tgt[:2], NS(**dict(DEFAULT_CFG, BASELINE=True)), 1)
    assert calls == [("base", ("s0",), "t0"), ("base", ("s1",), "t1")]


def _g15():
    import numpy as np
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g15_inference.npz"))


def _g15_case(g, name):
    import numpy as np
    logits = torch.from_numpy(g["logits_" + name])
    H, W = (int(v) for v in g["size_" + name])
    n = logits.shape[0] * H * W
    tie = torch.from_numpy(np.unpackbits(g["tie_" + name])[:n].astype(bool)).view(logits.shape[0], H, W)
    return logits, H, W, torch.from_numpy(g["pred_" + name]), torch.from_numpy(g["pred_cs_" + name]), torch.from_numpy(g["conf_" + name].astype("float32")), tie


def test_inference_oracle_and_label_table_match_the_reference_golden_g15():
    """Golden g15 = the reference's own inference path (infer_val.py:160-166, :78-88 and ITS `convert_to_cs` over
    `tools.category.labels`, captured by tests/golden/make_goldens.py): pins the hard-coded train id -> label id table of
    driver.py and oracle.head_ref.infer_labels, which the GPU test of the fused kernel is compared with."""
    import driver
    from oracle import head_ref as R
    g = _g15()
    assert tuple(int(v) for v in g["lut"]) == tuple(driver.CITYSCAPES_TRAIN_TO_ID)
    lut = torch.tensor(driver.CITYSCAPES_TRAIN_TO_ID, dtype=torch.uint8)
    for name in ("a", "b"):
        logits, H, W, pred, pred_cs, conf, tie = _g15_case(g, name)
        lab, c, gap = R.infer_labels(logits, H, W)
        assert torch.equal(lab[~tie], pred[~tie]) and float((lab != pred).float().mean()) < 1e-4
        lab_cs, _, _ = R.infer_labels(logits, H, W, lut)
        assert torch.equal(lab_cs[~tie], pred_cs[~tie])
        assert float((c - conf).abs().max()) < 1e-3          # (the golden stores the winning probability as fp16)


@pytest.mark.gpu
def test_fused_inference_kernel_matches_the_reference_golden_g15():
    """dasac_infer_labels (upsample + softmax + argmax + id table in one kernel) against what the reference's inference path
    wrote for the same logits (golden g15): identical label maps, both train ids and Cityscapes ids, outside exact near-ties."""
    import driver
    from dasac_hip import ops
    g = _g15()
    lut = torch.tensor(driver.CITYSCAPES_TRAIN_TO_ID, dtype=torch.uint8).cuda()
    for name in ("a", "b"):
        logits, H, W, pred, pred_cs, conf, tie = _g15_case(g, name)
        lab, c = ops.infer_labels(logits.cuda(), (H, W), want_conf=True)
        assert torch.equal(lab.cpu()[~tie], pred[~tie]) and float((lab.cpu() != pred).float().mean()) < 1e-4
        lab_cs, _ = ops.infer_labels(logits.cuda(), (H, W), lut)
        assert torch.equal(lab_cs.cpu()[~tie], pred_cs[~tie])
        assert float((c.cpu() - conf).abs().max()) < 1e-3
