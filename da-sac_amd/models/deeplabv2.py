"""DeepLabv2 (dilated ResNet-101 / VGG-16 + ASPP) -- drop-in for the reference's
`models/deeplabv2.py` (/root/reference/models/deeplabv2.py): same class names, constructor
arguments, `forward(im, y=None)` contract and state-dict keys; executed by the fused HIP engine.
"""
import torch
import torch.nn as nn

from dasac_hip.engine import Plan
from .basenet import BaseNet, check_criterion

BatchNorm = nn.SyncBatchNorm
ASPP_RATES = (6, 12, 18, 24)


class Bottleneck(nn.Module):
    """1x1(stride) -> 3x3(dilated) -> 1x1(x4) with BN after each conv (deeplabv2.py:54-99).  Container
    only: `plan()` emits three fused conv+BN+ReLU ops (the last one adds the shortcut)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, stride=stride, bias=False)
        self.bn1 = BatchNorm(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = BatchNorm(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = BatchNorm(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def plan(self, P, x):
        a = P.conv(x, self.conv1, self.bn1, relu=True)
        b = P.conv(a, self.conv2, self.bn2, relu=True)
        shortcut = x if self.downsample is None else P.conv(x, self.downsample[0], self.downsample[1])
        return P.conv(b, self.conv3, self.bn3, relu=True, res=shortcut)


class Classifier_Module(nn.Module):
    """ASPP head: sum of dilated 3x3 classifiers (deeplabv2.py:101-116) -- one fused contraction."""

    def __init__(self, fan_in, dilation_series, padding_series, num_classes):
        super().__init__()
        self.conv2d_list = nn.ModuleList(
            nn.Conv2d(fan_in, num_classes, 3, padding=p, dilation=d, bias=True) for d, p in zip(dilation_series, padding_series))
        for m in self.conv2d_list:
            m.weight.data.normal_(0, 0.01)

    def plan(self, P, x):
        return P.conv_sum(x, self.conv2d_list)


class ResNet(nn.Module):
    """Dilated ResNet, output stride 8 (deeplabv2.py:118-171): stem 7x7/2 + ceil-mode max-pool, stages
    with (stride, dilation) = (1,1),(2,1),(1,2),(1,4), ASPP classifier as `layer5`."""

    def __init__(self, block, layers, num_classes):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1, ceil_mode=True)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], dilation=2)
        self.layer4 = self._make_layer(block, 512, layers[3], dilation=4)
        self.layer5 = Classifier_Module(512 * block.expansion, ASPP_RATES, ASPP_RATES, num_classes)
        for m in self.modules():       # deeplabv2.py:135-141
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0, 0.01)
            elif isinstance(m, BatchNorm):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1):
        out_planes = planes * block.expansion
        shortcut = None
        if stride != 1 or self.inplanes != out_planes or dilation in (2, 4):
            shortcut = nn.Sequential(nn.Conv2d(self.inplanes, out_planes, 1, stride=stride, bias=False), BatchNorm(out_planes))
        seq = [block(self.inplanes, planes, stride, dilation=dilation, downsample=shortcut)]
        self.inplanes = out_planes
        seq += [block(self.inplanes, planes, dilation=dilation) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def plan(self, P, x):
        y = P.conv(x, self.conv1, self.bn1, relu=True)
        m = self.maxpool
        y = P.maxpool(y, m.kernel_size, m.stride, m.padding, m.ceil_mode)
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in stage:
                y = blk.plan(P, y)
        return self.layer5.plan(P, y)


class DeepLabV2_ResNet101(BaseNet):

    def __init__(self, num_classes=20, criterion=nn.CrossEntropyLoss(ignore_index=255, reduction="none"),
                 pretrained=None, freeze_bn=False):
        super().__init__()
        check_criterion(criterion)
        self.model = ResNet(Bottleneck, [3, 4, 23, 3], num_classes)
        if pretrained is not None:
            self._init_weights(pretrained)
        else:
            print("ResNet-101: Starting training from scratch")
        if freeze_bn:
            print("DeepLabv2/ResNet-101: Fixing BN")
            self._freeze_bn(self)
        self._from_scratch(self.model.layer5)
        self.criterion = criterion

    def _init_weights(self, path_to_weights):
        print("Loading weights from: ", path_to_weights)
        self.model.load_state_dict(torch.load(path_to_weights, map_location="cpu"), strict=False)

    def lr_mult(self):
        return 1., 10.

    def lr_mult_bias(self):
        return 2., 20.

    def _plan(self):
        P = Plan()
        return P.finish(self.model.plan(P, 0))

    def forward(self, im, y=None):
        """(logits, logits_up) when y is None, else ({"loss_ce"}, {"logits_up", "logits"}) (deeplabv2.py:213-227)."""
        return self._segment(im, y)


def _vgg16_features(batch_norm):
    """torchvision's VGG-16 `features` layer list (configuration "D"): the reference takes it from
    `torchvision.models.vgg16[_bn]()` (deeplabv2.py:238,243; fcn.py:23,32); torchvision is not a
    dependency here, the list is rebuilt with identical indices."""
    cfg = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")
    layers, cin = [], 3
    for v in cfg:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            continue
        layers.append(nn.Conv2d(cin, v, 3, padding=1))
        if batch_norm:
            layers.append(BatchNorm(v))
        layers.append(nn.ReLU(inplace=True))
        cin = v
    return layers


def plan_sequential(P, x, layers):
    """Emits fused ops for a conv[-BN][-ReLU] / max-pool / Dropout2d layer list."""
    i, n = 0, len(layers)
    while i < n:
        m = layers[i]
        if isinstance(m, nn.Conv2d):
            bn = layers[i + 1] if i + 1 < n and isinstance(layers[i + 1], BaseNet._batchnorm) else None
            j = i + (2 if bn is not None else 1)
            relu = j < n and isinstance(layers[j], nn.ReLU)
            x = P.conv(x, m, bn, relu=relu)
            i = j + (1 if relu else 0)
        elif isinstance(m, nn.MaxPool2d):
            x = P.maxpool(x, m.kernel_size, m.stride, m.padding, m.ceil_mode)
            i += 1
        elif isinstance(m, nn.Dropout2d):
            x = P.dropout2d(x, m)
            i += 1
        else:
            raise TypeError("no fused op for layer {}".format(m))
    return x


class DeepLabV2_VGG16(BaseNet):

    def __init__(self, num_classes, criterion=None, pretrained=None, use_bn=False, freeze_bn=False):
        super().__init__()
        check_criterion(criterion)
        self.criterion = criterion
        feats = _vgg16_features(use_bn)
        # conv5_x dilated by 2, pool4/pool5 removed (deeplabv2.py:236-260)
        dilate, drop = ((34, 37, 40), (33, 43)) if use_bn else ((24, 26, 28), (23, 30))
        if pretrained is not None:
            print("VGG16: Loading snapshot: ", pretrained)
            holder = nn.Module()
            holder.features = nn.Sequential(*feats)
            holder.load_state_dict(torch.load(pretrained, map_location="cpu"), strict=False)
        for i in dilate:
            feats[i].dilation, feats[i].padding = (2, 2), (2, 2)
        feats = [f for i, f in enumerate(feats) if i not in drop]
        fc6 = nn.Conv2d(512, 1024, 3, padding=4, dilation=4)
        fc7 = nn.Conv2d(1024, 1024, 3, padding=4, dilation=4)
        self.features = nn.Sequential(*(feats + [fc6, nn.ReLU(inplace=True), fc7, nn.ReLU(inplace=True)]))
        self.classifier = Classifier_Module(1024, ASPP_RATES, ASPP_RATES, num_classes)
        if freeze_bn:
            print("DeepLabv2/VGG-16: Fixing BN")
            self._freeze_bn(self)
        for new in (self.classifier, fc6, fc7):
            self._from_scratch(new)

    def lr_mult(self):
        return 1., 10.

    def lr_mult_bias(self):
        return 2., 20.

    def _plan(self):
        P = Plan()
        x = plan_sequential(P, 0, list(self.features))
        return P.finish(self.classifier.plan(P, x))

    def forward(self, im, y=None):
        return self._segment(im, y)
