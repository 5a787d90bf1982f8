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
