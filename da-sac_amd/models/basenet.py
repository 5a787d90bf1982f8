"""Base class of the segmentation networks -- drop-in for the reference's `models/basenet.py`
(/root/reference/models/basenet.py:13-143): same public surface (`from_scratch_layers`, `not_training`,
`bn_freeze`, `lr_mult`, `lr_mult_bias`, `train`, `parameter_groups`, `_resize_as`) and the same
four optimiser groups, so `train.py:93` and `base_trainer.get_optim` work unchanged.

The layers held by subclasses are ordinary `nn.Conv2d` / `nn.SyncBatchNorm` modules used as
*parameter containers* (identical state-dict keys and `isinstance` behaviour); the arithmetic is
executed by `dasac_hip.engine` on hand-written gfx950 kernels, never by these modules' own forward.
"""
import torch.nn as nn

from dasac_hip import engine as E


class BaseNet(nn.Module):

    # layer types whose weight/bias are optimised, and the normalisation types that can be frozen
    _trainable = (nn.Linear, nn.Conv2d, nn.ConvTranspose2d, nn.BatchNorm2d, nn.GroupNorm, nn.InstanceNorm2d, nn.SyncBatchNorm)
    _batchnorm = (nn.BatchNorm2d, nn.SyncBatchNorm, nn.GroupNorm)
    _returns_logits = True          # net_outs carries the stride-8 "logits" next to "logits_up" (fcn.py:149 does not)

    def __init__(self):
        super().__init__()
        self.from_scratch_layers = []   # layers trained with the "new" LR multipliers
        self.not_training = []          # layers whose parameters are frozen by train()
        self.bn_freeze = []             # BN layers that always stay in eval mode
        self._engine = None
        self._grad_sink = None          # dasac_hip.parallel.GradSink when wrapped by OverlappedDataParallel

    # -- LR multipliers [pre-trained, from-scratch] (basenet.py:32-40); subclasses override
    def lr_mult(self):
        return 1., 1.

    def lr_mult_bias(self):
        return 2., 2.

    def _is_learnable(self, layer):
        return isinstance(layer, BaseNet._trainable)

    def _from_scratch(self, net, ignore=[]):
        self.from_scratch_layers += [m for m in net.modules() if self._is_learnable(m)]

    def _freeze_bn(self, net, ignore=[]):
        """Registers every normalisation layer under `net` to be kept in eval mode (basenet.py:49-61)."""
        for m in net.modules():
            if isinstance(m, BaseNet._batchnorm) and m not in ignore:
                self.bn_freeze.append(m)
        print("Frozen BN: ", len(self.bn_freeze))

    def _fix_bn(self, layer):
        if isinstance(layer, nn.BatchNorm2d):
            self.not_training.append(layer)
        elif isinstance(layer, nn.Module):
            for child in layer.children():
                self._fix_bn(child)

    @staticmethod
    def _set_requires_grad(layer, flag):
        for name in ("weight", "bias"):
            p = getattr(layer, name, None)
            if p is not None:
                p.requires_grad = flag
        for child in layer.children():
            BaseNet._set_requires_grad(child, flag)

    def train(self, mode=True):
        """nn.Module.train plus the two freezes of basenet.py:86-100."""
        super().train(mode)
        for layer in self.not_training:
            self._set_requires_grad(layer, False)
        for layer in self.bn_freeze:
            layer.eval()
        return self

    def parameter_groups(self, base_lr, wd):
        """[old weights: lr,wd | old biases: 2lr,0 | new weights: 10lr,wd | new biases: 20lr,0] with the
        subclass multipliers (basenet.py:102-139).  BN gamma/beta count as weight/bias of an old layer."""
        w_old, w_new = self.lr_mult()
        b_old, b_new = self.lr_mult_bias()
        groups = [{"params": [], "weight_decay": wd, "lr": w_old * base_lr},
                  {"params": [], "weight_decay": 0.0, "lr": b_old * base_lr},
                  {"params": [], "weight_decay": wd, "lr": w_new * base_lr},
                  {"params": [], "weight_decay": 0.0, "lr": b_new * base_lr}]
        new = set(id(m) for m in self.from_scratch_layers)
        for m in self.modules():
            if not self._is_learnable(m):
                continue
            first = 2 if id(m) in new else 0
            if m.weight is not None and m.weight.requires_grad:
                groups[first]["params"].append(m.weight)
            if m.bias is not None and m.bias.requires_grad:
                groups[first + 1]["params"].append(m.bias)
        print("Optimising parameter groups: ")
        for i, g in enumerate(groups):
            print("[{}]: # parameters: {}, lr = {:4.3e}".format(i, len(g["params"]), g["lr"]))
        return tuple(groups)

    @staticmethod
    def _resize_as(x, y):
        return E.upsample_bilinear(x, y.size()[-2:])

    # ------------------------------------------------------------------ fused execution
    def _plan(self):
        raise NotImplementedError

    def _bn_frozen(self):
        """True when every BN of the net runs in eval mode (SAC mode / inference)."""
        return all(not m.training for m in self.modules() if isinstance(m, BaseNet._batchnorm))

    def _batch_fits(self, Nb, H, W):
        """True when a pass over Nb crops of H x W keeps every activation inside the kernels' 4 GiB addressing window."""
        if self._engine is None or self._engine.stale():
            self._engine = E.Engine(self._plan())
        return self._engine.largest_tensor_bytes(Nb, H, W) <= (1 << 32) - 4096

    def _logits(self, im):
        if self._engine is None or self._engine.stale():      # parameter objects replaced -> re-capture the plan
            self._engine = E.Engine(self._plan())
        return E.run_plan(self._engine, im, sink=self._grad_sink)

    def _segment(self, im, y, with_logits=True):
        """Shared tail of every backbone forward (deeplabv2.py:213-227, fcn.py:136-149)."""
        logits = self._logits(im)
        logits_up = E.upsample_bilinear(logits, im.size()[-2:])
        if y is None:
            return logits, logits_up
        losses = {"loss_ce": E.ce_mean_all_pixels(logits_up, y).view(1)}
        outs = {"logits_up": logits_up}
        if with_logits:
            outs["logits"] = logits
        return losses, outs


def check_criterion(criterion):
    """The fused CE kernel implements exactly CrossEntropyLoss(ignore_index=255, reduction='none') followed
    by .mean() (train.py:89, deeplabv2.py:223-224); anything else is refused rather than silently changed."""
    if criterion is None:
        return
    ok = isinstance(criterion, nn.CrossEntropyLoss) and criterion.ignore_index == 255 and criterion.reduction == "none" \
        and criterion.weight is None and getattr(criterion, "label_smoothing", 0.0) == 0.0
    if not ok:
        raise ValueError("dasac_hip implements criterion=nn.CrossEntropyLoss(ignore_index=255, reduction='none') only")
