"""Host-side behaviour of the drop-in `models` package that needs no GPU: state-dict layout, optimiser
groups, BN-freeze semantics of train(), refusal to compute on CPU (there is no CPU fallback)."""
from types import SimpleNamespace as NS

import pytest
import torch
import torch.nn as nn

from oracle.step_ref import DEFAULT_CFG

CRIT = nn.CrossEntropyLoss(ignore_index=255, reduction="none")


def _cfg(**kw):
    d = dict(DEFAULT_CFG)
    d.update(INIT_MODEL="")
    d.update(kw)
    return NS(**d)


def test_sac_state_dict_and_groups_match_reference(golden):
    import models
    g = golden("keys_sac_resnet101")
    net = models.get_model(_cfg(), 0, num_classes=19, criterion=CRIT)
    sd = net.state_dict()
    assert sorted(sd.keys()) == list(g["names"])
    shapes = dict(zip(g["names"], g["shapes"]))
    for k, v in sd.items():
        assert "x".join(str(s) for s in v.shape) == shapes[k], k
    groups = net.parameter_groups(1.0, 1.0)
    names = {id(p): n for n, p in net.named_parameters()}
    for i, gr in enumerate(groups):
        assert [names[id(p)] for p in gr["params"]] == list(g["group%d" % i])
        assert gr["lr"] == float(g["group_lr"][i]) and gr["weight_decay"] == float(g["group_wd"][i])
    assert all(not p.requires_grad for p in net.slow_net.parameters())
    assert float(net.slow_init[0]) == 0.0 and float(net.running_conf.abs().sum()) == 0.0


def test_vgg_and_fcn_keys(golden):
    import models
    net = models.get_model(_cfg(ARCH="deeplabv2_vgg16_bn"), 0, num_classes=19, criterion=CRIT)
    assert sorted(net.backbone.state_dict().keys()) == list(golden("g10_vgg16_deeplab")["keys"])
    net = models.get_model(_cfg(ARCH="fcn_vgg16_bn"), 0, num_classes=19, criterion=CRIT)
    assert sorted(net.backbone.state_dict().keys()) == list(golden("g10_fcn8s")["keys"])
    assert len(net.state_dict()) == 224


def test_train_keeps_frozen_bn_in_eval_and_baseline_trains_bn():
    import models
    net = models.get_model(_cfg(), 0, num_classes=19, criterion=CRIT)
    net.train()
    bns = [m for m in net.backbone.modules() if isinstance(m, nn.SyncBatchNorm)]
    assert len(bns) == 104 and all(not m.training for m in bns) and net.backbone.training
    assert all(m.weight.requires_grad for m in bns)       # gamma/beta stay trainable (basenet.py:112-131)
    base = models.get_model(_cfg(BASELINE=True), 0, num_classes=19, criterion=CRIT)
    base.train()
    assert isinstance(base, models.SAC_Baseline) and not isinstance(base, models.SAC)
    assert all(m.training for m in base.backbone.modules() if isinstance(m, nn.SyncBatchNorm))


def test_no_cpu_fallback_and_criterion_check():
    import models
    from dasac_hip import DasacError
    net = models.DeepLabV2_ResNet101(num_classes=19, criterion=CRIT, freeze_bn=True).eval()
    with pytest.raises(DasacError):
        net(torch.randn(1, 3, 33, 33))
    with pytest.raises(ValueError):
        models.DeepLabV2_ResNet101(num_classes=19, criterion=nn.CrossEntropyLoss())


def test_product_never_imports_the_oracle():
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "da-sac_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("oracle/", ""), os.path.join(dirpath, f)


def test_sac_exempts_exactly_the_frozen_bn_buffers_from_ddp_broadcast():
    """models/sac.py (ours): DDP must keep broadcasting `running_conf` / `slow_init` from rank 0 but not the
    running statistics of frozen BN layers, whose re-send would only invalidate the engine's packed-weight caches."""
    import torch.nn as nn
    import models
    from types import SimpleNamespace as NS
    from oracle.step_ref import DEFAULT_CFG
    net = models.get_model(NS(**dict(DEFAULT_CFG, INIT_MODEL="")), 0, num_classes=19,
                           criterion=nn.CrossEntropyLoss(ignore_index=255, reduction="none"))
    ign, bufs = set(net._ddp_params_and_buffers_to_ignore), dict(net.named_buffers())
    assert ign <= set(bufs) and set(bufs) - ign == {"running_conf", "slow_init"}
    assert all(k.split(".")[-1] in ("running_mean", "running_var", "num_batches_tracked") for k in ign)
    base = models.get_model(NS(**dict(DEFAULT_CFG, INIT_MODEL="", BASELINE=True)), 0, num_classes=19,
                            criterion=nn.CrossEntropyLoss(ignore_index=255, reduction="none"))
    assert not hasattr(base, "_ddp_params_and_buffers_to_ignore")      # baseline mode trains its BN: everything stays broadcast
