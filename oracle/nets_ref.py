"""Oracle (CPU, test infrastructure) for the segmentation networks, written as pure
functions over a flat state dict that uses the reference's checkpoint keys.

Follows /root/reference/models/deeplabv2.py (:54-99 Bottleneck, :101-116 ASPP sum,
:118-171 dilated ResNet, :173-227 DeepLabV2_ResNet101, :229-312 DeepLabV2_VGG16) and
/root/reference/models/fcn.py (:10-149 VGG16_FCN8s).  conv/BN/pool arithmetic is ATen CPU.
Autograd on the returned tensors gives the oracle gradients.
"""
import torch
import torch.nn.functional as F

from .head_ref import upsample_bilinear_ac, ce_mean_all_pixels

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

RESNET101_STAGES = ((64, 3, 1, 1), (128, 4, 2, 1), (256, 23, 1, 2), (512, 3, 1, 4))  # planes, blocks, stride, dilation
ASPP_RATES = (6, 12, 18, 24)
VGG_D = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")


# ---------------------------------------------------------------------------- state dicts
def _conv_w(gen, cout, cin, k, std=0.01):
    return torch.empty(cout, cin, k, k).normal_(0, std, generator=gen)


def _bn_entries(sd, prefix, c, gen=None, randomize=False):
    if randomize:
        sd[prefix + ".weight"] = torch.empty(c).uniform_(0.5, 1.5, generator=gen)
        sd[prefix + ".bias"] = torch.empty(c).normal_(0, 0.1, generator=gen)
        sd[prefix + ".running_mean"] = torch.empty(c).normal_(0, 0.1, generator=gen)
        sd[prefix + ".running_var"] = torch.empty(c).uniform_(0.5, 1.5, generator=gen)
    else:
        sd[prefix + ".weight"] = torch.ones(c)
        sd[prefix + ".bias"] = torch.zeros(c)
        sd[prefix + ".running_mean"] = torch.zeros(c)
        sd[prefix + ".running_var"] = torch.ones(c)
    sd[prefix + ".num_batches_tracked"] = torch.zeros((), dtype=torch.int64)


def resnet101_state(seed=0, num_classes=19, randomize_bn=False, aspp_gain=1.0, w_std=0.01,
                    stages=RESNET101_STAGES, he_init=False, residual_gain=1.0):
    """Deterministic state dict with the 632 keys of DeepLabV2_ResNet101.
    Same init family as deeplabv2.py:135-141 (N(0,0.01) convs, BN gamma=1 beta=0);
    `he_init` switches conv std to sqrt(2/fan_in) and `residual_gain` scales the last BN
    gamma of every block, so that activations keep O(1) scale through the 33 residual
    blocks (well-conditioned parity nets)."""
    gen = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(key, cout, cin, k):
        std = (2.0 / (cin * k * k)) ** 0.5 if he_init else w_std
        sd[key + ".weight"] = _conv_w(gen, cout, cin, k, std)

    conv("model.conv1", 64, 3, 7)
    _bn_entries(sd, "model.bn1", 64, gen, randomize_bn)
    cin = 64
    for li, (planes, blocks, stride, dil) in enumerate(stages, start=1):
        for bi in range(blocks):
            p = "model.layer{}.{}".format(li, bi)
            conv(p + ".conv1", planes, cin, 1)
            _bn_entries(sd, p + ".bn1", planes, gen, randomize_bn)
            conv(p + ".conv2", planes, planes, 3)
            _bn_entries(sd, p + ".bn2", planes, gen, randomize_bn)
            conv(p + ".conv3", planes * 4, planes, 1)
            _bn_entries(sd, p + ".bn3", planes * 4, gen, randomize_bn)
            sd[p + ".bn3.weight"] *= residual_gain
            if bi == 0:
                conv(p + ".downsample.0", planes * 4, cin, 1)
                _bn_entries(sd, p + ".downsample.1", planes * 4, gen, randomize_bn)
            cin = planes * 4
    for i in range(len(ASPP_RATES)):
        p = "model.layer5.conv2d_list.{}".format(i)
        std = (2.0 / (cin * 9)) ** 0.5 if he_init else w_std
        sd[p + ".weight"] = _conv_w(gen, num_classes, cin, 3, std) * aspp_gain
        sd[p + ".bias"] = torch.empty(num_classes).normal_(0, 0.01, generator=gen) * aspp_gain
    return sd


# ------------------------------------------------------------------------------- layers
def batchnorm(sd, prefix, x, train, momentum=BN_MOMENTUM):
    """SyncBatchNorm on one process == F.batch_norm (deeplabv2.py:15).  In train mode the
    running stats in `sd` are updated in place (momentum 0.1, unbiased var), as ATen does."""
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if train:
        sd[prefix + ".num_batches_tracked"] += 1
    return F.batch_norm(x, rm, rv, sd[prefix + ".weight"], sd[prefix + ".bias"], train, momentum, BN_EPS)


def bottleneck(sd, p, x, stride, dilation, bn_train, act=F.relu):
    """deeplabv2.py:79-99: stride sits on the first 1x1 (:59).  `act` = the ReLU (tests may inject a
    mask-driven one, see MaskedRelu)."""
    y = F.conv2d(x, sd[p + ".conv1.weight"], stride=stride)
    y = act(batchnorm(sd, p + ".bn1", y, bn_train))
    y = F.conv2d(y, sd[p + ".conv2.weight"], padding=dilation, dilation=dilation)
    y = act(batchnorm(sd, p + ".bn2", y, bn_train))
    y = batchnorm(sd, p + ".bn3", F.conv2d(y, sd[p + ".conv3.weight"]), bn_train)
    if (p + ".downsample.0.weight") in sd:
        x = batchnorm(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride), bn_train)
    return act(y + x)


class MaskedRelu:
    """ReLU whose on/off pattern is dictated from outside (one mask per call, in call order): y = z*mask.
    Lets a test compare gradients of two fp32 implementations without the handful of units whose
    pre-activation sits within round-off of zero deciding the outcome."""

    def __init__(self, masks):
        self.masks, self.i, self.disagree, self.total = list(masks), 0, 0, 0

    def __call__(self, z):
        m = self.masks[self.i].to(z.dtype)
        self.i += 1
        self.disagree += int(((z.detach() > 0) != (m > 0)).sum())
        self.total += m.numel()
        return z * m


def aspp_sum(sd, p, x):
    """deeplabv2.py:112-116: sum of the four dilated 3x3 classifiers (bias each)."""
    out = None
    for i, r in enumerate(ASPP_RATES):
        o = F.conv2d(x, sd["{}.{}.weight".format(p, i)], sd["{}.{}.bias".format(p, i)], padding=r, dilation=r)
        out = o if out is None else out + o
    return out


def resnet101_logits(sd, x, bn_train=False, stages=RESNET101_STAGES, act=F.relu):
    """deeplabv2.py:160-171."""
    y = F.conv2d(x, sd["model.conv1.weight"], stride=2, padding=3)
    y = act(batchnorm(sd, "model.bn1", y, bn_train))
    y = F.max_pool2d(y, 3, 2, 1, ceil_mode=True)
    for li, (planes, blocks, stride, dil) in enumerate(stages, start=1):
        for bi in range(blocks):
            y = bottleneck(sd, "model.layer{}.{}".format(li, bi), y, stride if bi == 0 else 1, dil, bn_train, act)
    return aspp_sum(sd, "model.layer5.conv2d_list", y)


# ---- VGG-16 (torchvision cfg "D" with BN; third-party layer list) -----------
def vgg16_bn_state(seed=0, randomize_bn=False):
    """`features.{i}` keys of torchvision vgg16_bn().features (conv3x3+bias, BN, ReLU; pools
    at 6,13,23,33,43)."""
    gen = torch.Generator().manual_seed(seed)
    sd, idx, cin = {}, 0, 3
    for v in VGG_D:
        if v == "M":
            idx += 1
            continue
        sd["features.{}.weight".format(idx)] = _conv_w(gen, v, cin, 3, (2.0 / (cin * 9)) ** 0.5)
        sd["features.{}.bias".format(idx)] = torch.empty(v).normal_(0, 0.01, generator=gen)
        _bn_entries(sd, "features.{}".format(idx + 1), v, gen, randomize_bn)
        idx += 3
        cin = v
    return sd


def _vgg_plan():
    """[(kind, torchvision index, channels)] for vgg16_bn().features."""
    plan, idx = [], 0
    for v in VGG_D:
        if v == "M":
            plan.append(("pool", idx, None))
            idx += 1
        else:
            plan.append(("conv", idx, v))
            plan.append(("bn", idx + 1, v))
            plan.append(("relu", idx + 2, None))
            idx += 3
    return plan


def deeplab_vgg16_state(seed=0, num_classes=19, randomize_bn=False, aspp_gain=1.0):
    """DeepLabV2_VGG16(use_bn=True) keys: pools 33,43 dropped and the list re-indexed
    (deeplabv2.py:239-240,255-267) -> dilated convs at 33/36/39, fc6 at 42, fc7 at 44."""
    gen = torch.Generator().manual_seed(seed)
    base = vgg16_bn_state(seed, randomize_bn)
    sd, new = {}, 0
    for kind, old, ch in _vgg_plan():
        if kind == "pool" and old in (33, 43):
            continue
        for k, v in base.items():
            if k.startswith("features.{}.".format(old)):
                sd["features.{}.{}".format(new, k.split(".")[-1])] = v
        new += 1
    sd["features.{}.weight".format(new)] = _conv_w(gen, 1024, 512, 3, (2.0 / (512 * 9)) ** 0.5)
    sd["features.{}.bias".format(new)] = torch.zeros(1024)
    sd["features.{}.weight".format(new + 2)] = _conv_w(gen, 1024, 1024, 3, (2.0 / (1024 * 9)) ** 0.5)
    sd["features.{}.bias".format(new + 2)] = torch.zeros(1024)
    for i in range(len(ASPP_RATES)):
        sd["classifier.conv2d_list.{}.weight".format(i)] = _conv_w(gen, num_classes, 1024, 3, 0.01) * aspp_gain
        sd["classifier.conv2d_list.{}.bias".format(i)] = torch.empty(num_classes).normal_(0, 0.01, generator=gen)
    return sd


def deeplab_vgg16_logits(sd, x, bn_train=False):
    """deeplabv2.py:292-296 with the surgery of :255-267 (convs #34,37,40 -> dilation 2)."""
    new = 0
    for kind, old, ch in _vgg_plan():
        if kind == "pool" and old in (33, 43):
            continue
        p = "features.{}".format(new)
        if kind == "conv":
            d = 2 if old in (34, 37, 40) else 1
            x = F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=d, dilation=d)
        elif kind == "bn":
            x = batchnorm(sd, p, x, bn_train)
        elif kind == "relu":
            x = F.relu(x)
        else:
            x = F.max_pool2d(x, 2, 2)
        new += 1
    for j in (new, new + 2):
        x = F.relu(F.conv2d(x, sd["features.{}.weight".format(j)], sd["features.{}.bias".format(j)], padding=4, dilation=4))
    return aspp_sum(sd, "classifier.conv2d_list", x)


def fcn8s_vgg16_state(seed=0, num_classes=19, randomize_bn=False):
    """VGG16_FCN8s(use_bn=True) keys (fcn.py:27-29,48-58,78,88): block1.{0..23},
    block2.{24..33}, block3.{34..43} (slicing a Sequential keeps the child names)."""
    gen = torch.Generator().manual_seed(seed)
    base = vgg16_bn_state(seed, randomize_bn)
    sd = {}
    for k, v in base.items():
        i = int(k.split(".")[1])
        # nn.Sequential slices keep the original child names -> block2.24.., block3.34..
        blk = "block1" if i < 24 else ("block2" if i < 34 else "block3")
        sd["{}.{}.{}".format(blk, i, k.split(".")[-1])] = v
    sd["vgg_head.0.weight"] = _conv_w(gen, 4096, 512, 7, (2.0 / (512 * 49)) ** 0.5)
    sd["vgg_head.0.bias"] = torch.zeros(4096)
    _bn_entries(sd, "vgg_head.1", 4096, gen, randomize_bn)
    sd["vgg_head.4.weight"] = _conv_w(gen, 4096, 4096, 1, (2.0 / 4096) ** 0.5)
    sd["vgg_head.4.bias"] = torch.zeros(4096)
    _bn_entries(sd, "vgg_head.5", 4096, gen, randomize_bn)
    sd["vgg_head.8.weight"] = _conv_w(gen, num_classes, 4096, 1, 0.01)
    sd["vgg_head.8.bias"] = torch.zeros(num_classes)
    sd["score_pool4.weight"] = _conv_w(gen, num_classes, 512, 1, 0.01)
    sd["score_pool4.bias"] = torch.zeros(num_classes)
    sd["score_pool3.weight"] = _conv_w(gen, num_classes, 256, 1, 0.01)
    sd["score_pool3.bias"] = torch.zeros(num_classes)
    return sd


def _vgg_block(sd, name, lo, hi, x, bn_train):
    for kind, old, ch in _vgg_plan():
        if not (lo <= old < hi):
            continue
        p = "{}.{}".format(name, old)
        if kind == "conv":
            x = F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=1)
        elif kind == "bn":
            x = batchnorm(sd, p, x, bn_train)
        elif kind == "relu":
            x = F.relu(x)
        else:
            x = F.max_pool2d(x, 2, 2)
    return x


def _up2(x):
    return upsample_bilinear_ac(x, 2 * x.shape[2], 2 * x.shape[3])


def fcn8s_vgg16_logits(sd, x, bn_train=False, drop_masks=None):
    """fcn.py:111-134.  Dropout2d (p=.1) is the identity unless per-channel keep masks
    (already divided by 1-p) are injected through `drop_masks` = (m1, m2) [B,4096,1,1]."""
    p3 = _vgg_block(sd, "block1", 0, 24, x, bn_train)
    p4 = _vgg_block(sd, "block2", 24, 34, p3, bn_train)
    p5 = _vgg_block(sd, "block3", 34, 44, p4, bn_train)
    s = F.conv2d(p5, sd["vgg_head.0.weight"], sd["vgg_head.0.bias"], padding=3)
    s = F.relu(batchnorm(sd, "vgg_head.1", s, bn_train))
    if drop_masks is not None:
        s = s * drop_masks[0]
    s = F.conv2d(s, sd["vgg_head.4.weight"], sd["vgg_head.4.bias"])
    s = F.relu(batchnorm(sd, "vgg_head.5", s, bn_train))
    if drop_masks is not None:
        s = s * drop_masks[1]
    s = F.conv2d(s, sd["vgg_head.8.weight"], sd["vgg_head.8.bias"])
    s = _up2(s) + F.conv2d(p4, sd["score_pool4.weight"], sd["score_pool4.bias"])
    s = _up2(s) + F.conv2d(p3, sd["score_pool3.weight"], sd["score_pool3.bias"])
    return s


LOGITS_FN = {
    "deeplabv2_resnet101": resnet101_logits,
    "deeplabv2_vgg16_bn": deeplab_vgg16_logits,
    "fcn_vgg16_bn": fcn8s_vgg16_logits,
}
STATE_FN = {
    "deeplabv2_resnet101": resnet101_state,
    "deeplabv2_vgg16_bn": deeplab_vgg16_state,
    "fcn_vgg16_bn": fcn8s_vgg16_state,
}


def segnet_forward(arch, sd, im, y=None, bn_train=False, **kw):
    """deeplabv2.py:213-227 / :298-312 / fcn.py:136-149: logits -> bilinear(ac=True) to the
    input size -> per-pixel CE averaged over all pixels."""
    logits = LOGITS_FN[arch](sd, im, bn_train, **kw)
    up = upsample_bilinear_ac(logits, im.shape[2], im.shape[3])
    if y is None:
        return logits, up
    outs = {"logits_up": up}
    if arch != "fcn_vgg16_bn":                       # fcn.py:149 returns logits_up only
        outs["logits"] = logits
    return {"loss_ce": ce_mean_all_pixels(up, y)}, outs


def trainable_keys(sd):
    return [k for k in sd if k.split(".")[-1] in ("weight", "bias")]
