""""PyTorch is allocator, stream and torch.distributed provider".  One whole training iteration (both student
schedules) runs under a TorchDispatchMode that records every ATen operator PyTorch executes: everything that touches more
than a handful of elements must be allocation / view / copy plumbing -- all arithmetic on activations, labels, weights and
gradients is in libdasac_hip.so (which the dispatcher never sees: the kernels are launched through ctypes)."""
from types import SimpleNamespace as NS

import pytest
import torch
import torch.nn as nn
from torch.utils._python_dispatch import TorchDispatchMode

from oracle import nets_ref as N
from oracle.step_ref import DEFAULT_CFG

pytestmark = pytest.mark.gpu

# moving / shaping bytes: no arithmetic on the data
PLUMBING = {"empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided", "zeros", "zeros_like", "ones", "full", "zero_", "fill_",
            "view", "_unsafe_view", "reshape", "as_strided", "narrow", "slice", "select", "expand", "permute", "transpose", "t", "squeeze",
            "unsqueeze", "flatten", "unflatten", "detach", "detach_", "alias", "clone", "contiguous", "copy_", "_to_copy", "to", "cat",
            "lift_fresh", "_local_scalar_dense", "item", "is_pinned", "_pin_memory", "pin_memory", "record_stream", "set_", "resize_",
            "scalar_tensor", "result_type", "_has_compatible_shallow_copy_type", "is_same_size", "equal", "unbind", "split", "chunk", "stack"}
SMALL = 64          # per-class vectors (19), losses ([1]), thetas ([L,2,3] would be 48): arithmetic on these is bookkeeping


class Recorder(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.big = []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.overloadpacket.__name__ if hasattr(func, "overloadpacket") else str(func)
        if name not in PLUMBING:
            sizes = [t.numel() for t in torch.utils._pytree.tree_leaves((args, kwargs, out)) if isinstance(t, torch.Tensor)]
            if sizes and max(sizes) > SMALL:
                self.big.append((name, max(sizes)))
        return out


@pytest.mark.parametrize("fuse", [False, True])
def test_one_training_iteration_runs_no_aten_arithmetic_on_tensors(fuse):
    import models
    import driver
    cfg = NS(**dict(DEFAULT_CFG, INIT_MODEL="", OPT_NESTEROV=False))
    net = models.get_model(cfg, 0, num_classes=19, criterion=nn.CrossEntropyLoss(ignore_index=255, reduction="none"))
    net.backbone.load_state_dict(N.resnet101_state(seed=3, randomize_bn=True, he_init=True, residual_gain=0.25, aspp_gain=0.2), strict=True)
    net.cuda().train()
    optim = driver.make_optimizer(net, cfg)
    src, tgt = driver.synthetic_batches(2, 1, 2, (33, 49), "cuda", seed=5)
    clone = lambda: (tgt[0], tgt[1].clone(), tgt[2], tgt[3], tgt[4])
    driver.sac_train_iteration(net, optim, src, clone(), 2, True, cfg.LR_TARGET, fuse_passes=fuse)       # first call: teacher init, caches
    for update in (True, False):
        with Recorder() as rec:
            driver.sac_train_iteration(net, optim, src, clone(), 2, update, cfg.LR_TARGET, fuse_passes=fuse)
            torch.cuda.synchronize()
        assert not rec.big, sorted(set(rec.big))[:12]
