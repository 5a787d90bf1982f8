"""Fused multi-tensor SGD for the reference's optimiser (base_trainer.py:63-66:
`torch.optim.SGD(param_groups, momentum=MOMENTUM, nesterov=OPT_NESTEROV)` over the four groups of
models/basenet.py:73-95).  Same update rule, `param_groups` and `state[p]["momentum_buffer"]` layout as
torch.optim.SGD (so LR schedules that poke `param_groups[i]["lr"]` and optimiser checkpoints keep working), but one
HIP launch (dasac_sgd_step) instead of ~4 foreach passes per group."""
import ctypes

import numpy as np
import torch

from . import lib as L
from . import ops


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0, nesterov=False, dampening=0.0):
        if nesterov or dampening != 0.0:
            raise NotImplementedError("FusedSGD: plain momentum only (the reference's default OPT_NESTEROV=False)")
        defaults = dict(lr=lr, momentum=momentum, dampening=0.0, weight_decay=weight_decay, nesterov=False)
        super().__init__(params, defaults)
        if len(self.param_groups) > 8:
            raise ValueError("FusedSGD: at most 8 parameter groups")
        self._tables = {}            # first(bool) -> (pointer key, device tensor table, device chunk table, n_tensors, n_chunks)
        self._stash = {}             # parameter -> gradient of an earlier backward pass, summed inside the next step()

    def stash_grads(self):
        """Sets the current gradients aside (p.grad becomes None): the next backward pass then ASSIGNS its gradients instead
        of accumulating into the old ones (one `add_` launch per parameter), and step() applies stash + grad in the update
        kernel -- the same single fp32 addition, in AccumulateGrad's operand order."""
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None:
                    if p in self._stash:
                        self._stash[p] = self._stash[p] + p.grad
                    else:
                        self._stash[p] = p.grad
                    p.grad = None

    def full_grads(self):
        """{parameter: stash + .grad} -- the gradient the next step() will apply, as the reference's `.grad` would hold it
        after both backward passes (for clipping / norm logging between the last backward and step())."""
        out = {}
        for group in self.param_groups:
            for p in group["params"]:
                g, g2 = p.grad, self._stash.get(p)
                if g is None and g2 is None:
                    continue
                out[p] = g if g2 is None else (g2 if g is None else g2 + g)
        return out

    def zero_grad(self, set_to_none=True):
        self._stash.clear()
        super().zero_grad(set_to_none=set_to_none)

    def _table(self, first, rows, device):
        key = tuple(v for r in rows for v in r)
        ent = self._tables.get(first)
        if ent is None or ent[0] != key:
            chunk = L.load().dasac_ema_chunk_elems()
            chunks = [(i, j) for i, r in enumerate(rows) for j in range((r[4] + chunk - 1) // chunk)]
            # The gradients are fresh allocations every step, so this table is rebuilt every step: upload it through pinned
            # memory without blocking.  A pageable source makes `.to(device)` wait for the WHOLE stream -- the backward pass
            # still running on the device -- and the device then idles while the host catches up (measured: ~3 ms per step).
            up = lambda a: torch.from_numpy(a).pin_memory().to(device, non_blocking=True)
            ent = (key, up(np.asarray(rows, dtype=np.int64)), up(np.asarray(chunks, dtype=np.int32).reshape(-1, 2)),
                   len(rows), len(chunks))
            self._tables[first] = ent
        return ent

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.load()
        momentum = self.param_groups[0]["momentum"]
        rows = {True: [], False: []}
        device, keep, touched = None, [], []
        for gi, group in enumerate(self.param_groups):
            if group["momentum"] != momentum or group.get("nesterov") or group.get("dampening", 0.0) != 0.0 or group.get("maximize"):
                raise NotImplementedError("FusedSGD: one momentum for all groups, no nesterov / dampening / maximize")
            for p in group["params"]:
                g2 = self._stash.get(p)
                if p.grad is None:
                    if g2 is None:
                        continue
                    p.grad, g2 = g2, None            # only the stashed pass produced a gradient for this parameter
                L.require_gpu(p, p.grad, g2)
                if p.dtype != torch.float32 or not p.is_contiguous() or p.grad.is_sparse:
                    raise TypeError("FusedSGD: dense contiguous fp32 parameters only")
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if g2 is not None and not g2.is_contiguous():
                    g2 = g2.contiguous()
                st = self.state[p]
                first = st.get("momentum_buffer") is None
                if first:
                    st["momentum_buffer"] = torch.empty_like(p, memory_format=torch.contiguous_format)
                device = p.device
                rows[first or momentum == 0.0].append((p.data_ptr(), g.data_ptr(), 0 if g2 is None else g2.data_ptr(),
                                                       st["momentum_buffer"].data_ptr(), p.numel(), gi))
                keep += [g, g2]                          # a made-contiguous copy must outlive the queued launch
                touched += [p, st["momentum_buffer"]]
        n = len(self.param_groups)
        lr = (ctypes.c_float * n)(*[float(g["lr"]) for g in self.param_groups])
        wd = (ctypes.c_float * n)(*[float(g["weight_decay"]) for g in self.param_groups])
        for first in (True, False):
            if not rows[first]:
                continue
            _, tab, chunks, nt, nc = self._table(first, rows[first], device)
            L.check(lib.dasac_sgd_step(tab.data_ptr(), nt, chunks.data_ptr(), nc, ctypes.cast(lr, ctypes.c_void_p),
                                       ctypes.cast(wd, ctypes.c_void_p), n, float(momentum), int(first), L.stream_ptr()),
                    "dasac_sgd_step")
        ops.bump_versions(touched)       # raw-pointer writes: keep autograd's version counters (engine cache keys) honest
        self._stash.clear()
        return loss
