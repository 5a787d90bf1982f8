"""Drop-in `models` package: `from models import get_model` resolves here when `da-sac_amd/` is
on sys.path ahead of the reference checkout (/root/reference/models/__init__.py:14-41)."""
import os
from functools import partial

from .deeplabv2 import DeepLabV2_ResNet101, DeepLabV2_VGG16
from .fcn import VGG16_FCN8s
from .sac import SAC, SAC_Baseline

ARCHS = {
    "deeplabv2_resnet101": DeepLabV2_ResNet101,
    "deeplabv2_vgg16_bn": partial(DeepLabV2_VGG16, use_bn=True),
    "fcn_vgg16_bn": partial(VGG16_FCN8s, use_bn=True),
}


def get_model(cfg, rank, *args, **kwargs):
    """cfg = cfg.MODEL.  Baseline (AdaBN) -> SAC_Baseline(student); otherwise SAC(student, momentum copy)
    with every BatchNorm frozen (`freeze_bn = not cfg.BASELINE`)."""
    if len(cfg.INIT_MODEL) > 0 and os.path.isfile(cfg.INIT_MODEL):
        kwargs["pretrained"] = cfg.INIT_MODEL
    else:
        print("Backbone model not found: {}".format(cfg.INIT_MODEL))
    kwargs["freeze_bn"] = not cfg.BASELINE
    make = ARCHS[cfg.ARCH.lower()]
    student = make(*args, **kwargs)
    if cfg.BASELINE:
        return SAC_Baseline(cfg, student, rank, **kwargs)
    return SAC(cfg, student, make(*args, **kwargs), rank, **kwargs)
