"""FCN-8s on VGG-16 -- drop-in for the reference's `models/fcn.py` (/root/reference/models/fcn.py:10-149)."""
import torch
import torch.nn as nn

from dasac_hip.engine import Plan
from dasac_hip import engine as E
from .basenet import BaseNet, check_criterion
from .deeplabv2 import _vgg16_features, plan_sequential, BatchNorm


class VGG16_FCN8s(BaseNet):
    _returns_logits = False         # fcn.py:149 returns only {"logits_up"}

    def __init__(self, num_classes, criterion=None, pretrained=None, use_bn=False, freeze_bn=False, drop_rate=0.1):
        super().__init__()
        check_criterion(criterion)
        self.criterion = criterion
        feats = nn.Sequential(*_vgg16_features(use_bn))
        # pool3 / pool4 close block1 / block2; slices of a Sequential keep the child names (fcn.py:26-36)
        cut = (24, 34) if use_bn else (17, 24)
        self.block1, self.block2, self.block3 = feats[:cut[0]], feats[cut[0]:cut[1]], feats[cut[1]:]
        if pretrained is not None:
            print("VGG16-FCN8s: Loading snapshot: ", pretrained)
            holder = nn.Module()
            holder.features = feats
            holder.load_state_dict(torch.load(pretrained, map_location="cpu"), strict=False)
        else:
            print("VGG16-FCN8s: Initialising from scratch")
        head = [nn.Conv2d(512, 4096, 7, padding=3)]
        head += ([BatchNorm(4096)] if use_bn else []) + [nn.ReLU(inplace=True), nn.Dropout2d(p=drop_rate), nn.Conv2d(4096, 4096, 1)]
        head += ([BatchNorm(4096)] if use_bn else []) + [nn.ReLU(inplace=True), nn.Dropout2d(p=drop_rate), nn.Conv2d(4096, num_classes, 1)]
        self.vgg_head = nn.Sequential(*head)
        if freeze_bn:
            self._freeze_bn(self)
        self._from_scratch(self.vgg_head)
        self.score_pool4 = nn.Conv2d(512, num_classes, 1)
        self.score_pool4.weight.data.normal_(0, 0.01)
        self._from_scratch(self.score_pool4)
        self.score_pool3 = nn.Conv2d(256, num_classes, 1)
        self.score_pool3.weight.data.normal_(0, 0.01)
        self._from_scratch(self.score_pool3)

    def lr_mult(self):
        return 1., 10.

    def lr_mult_bias(self):
        return 2., 20.

    @staticmethod
    def up_x2(x):
        return E.upsample_bilinear(x, (2 * x.size(2), 2 * x.size(3)))

    def _plan(self):
        """fcn.py:111-134: scores at 1/32, fused upward with the pool4 (1/16) and pool3 (1/8) scores."""
        P = Plan()
        p3 = plan_sequential(P, 0, list(self.block1))
        p4 = plan_sequential(P, p3, list(self.block2))
        p5 = plan_sequential(P, p4, list(self.block3))
        s = plan_sequential(P, p5, list(self.vgg_head))
        s = P.up2_add(s, P.conv(p4, self.score_pool4))
        s = P.up2_add(s, P.conv(p3, self.score_pool3))
        return P.finish(s)

    def forward(self, x, y=None):
        """Training call returns ({"loss_ce"}, {"logits_up"}) only (fcn.py:149)."""
        return self._segment(x, y, with_logits=False)
