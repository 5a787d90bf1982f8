"""Generator-side helper (runs only in the build container, never on the GPU box).

Imports the *reference* implementation from /root/reference on CPU so that the
golden-vector generator scripts next to this file can capture its outputs.
`torchvision` is absent in this image; the reference only needs
`torchvision.models.vgg16` / `vgg16_bn` as a layer list (models/deeplabv2.py:238,243,
models/fcn.py:23,32), so a minimal stand-in for that third-party layer list
(torchvision cfg "D") is registered in sys.modules first.

Nothing in here is shipped or imported by the product, the tests or bench.py.
"""
import sys
import types

import torch.nn as nn

REFERENCE_ROOT = "/root/reference"

_VGG_D = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]


class _VGG(nn.Module):
    def __init__(self, bn):
        super().__init__()
        seq, cin = [], 3
        for v in _VGG_D:
            if v == "M":
                seq.append(nn.MaxPool2d(2, 2))
                continue
            seq.append(nn.Conv2d(cin, v, 3, padding=1))
            if bn:
                seq.append(nn.BatchNorm2d(v))
            seq.append(nn.ReLU(inplace=True))
            cin = v
        self.features = nn.Sequential(*seq)
        self.classifier = nn.Sequential(*[nn.Identity() for _ in range(7)])


def install_torchvision_stub():
    if "torchvision" in sys.modules:
        return
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvm.vgg16 = lambda **kw: _VGG(False)
    tvm.vgg16_bn = lambda **kw: _VGG(True)
    tv.models = tvm
    tvt = types.ModuleType("torchvision.transforms")
    tvt.__path__ = []
    tvf = types.ModuleType("torchvision.transforms.functional")
    tvt.functional = tvf
    sys.modules["torchvision.transforms.functional"] = tvf
    tv.transforms = tvt
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tvm
    sys.modules["torchvision.transforms"] = tvt


def import_reference():
    """Returns the reference's `models` package and its global cfg."""
    install_torchvision_stub()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import models as ref_models  # noqa: the reference package
    from core.config import cfg as ref_cfg
    return ref_models, ref_cfg
