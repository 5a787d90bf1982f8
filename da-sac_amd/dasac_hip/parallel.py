"""Data-parallel wrapper with the gradient all-reduce OVERLAPPED with the hand-written backward pass.

The reference wraps its model in `torch.nn.parallel.DistributedDataParallel` (/root/reference/train.py:104): DDP's
reducer hooks every parameter's AccumulateGrad node and fires a bucket's all-reduce as soon as its gradients exist, so
the reduction of the upper layers runs under the backward of the lower ones (train.py:133,232).  Here the whole
backbone is ONE autograd.Function (dasac_hip.engine): autograd -- and with it DDP's hooks -- sees all 320 gradients at
once, after the last data-gradient GEMM; under stock DDP both 176 MB reductions of a step are exposed and the reducer
adds its own bucket copies (+13 ms per step measured on ONE rank).

`OverlappedDataParallel` keeps DDP's semantics (parameters and buffers broadcast from rank 0 at construction, buffers
re-broadcast before every forward, gradients averaged over ranks after every backward pass, `.module`,
"module."-prefixed state dict; `no_sync()` is not offered -- the reference never uses it) and moves the reduction
INTO the engine's backward:

  * a `GradSink` per trainable backbone hands the engine slices of ONE flat fp32 buffer to write the parameter
    gradients into (no per-parameter allocations, no bucket copies: the weight-gradient finish kernels, the BN
    parameter-gradient kernel and the channel-sum kernels write straight into the reduction buffer);
  * the slices are laid out in the order the backward pass completes them (layer5 -> conv1) and cut into buckets;
    when the engine reports a bucket's last layer done, `dist.all_reduce(bucket, async_op=True)` is issued on the
    process group's own stream (RCCL over xGMI with backend "nccl") while the engine keeps launching GEMMs;
  * when the backbone's backward returns, the launch stream waits (stream-side, no host sync) for the outstanding
    reductions -- so `.grad` is final for whatever runs next on the stream (clipping, logging, any optimiser).

Averaging: the incoming loss gradient is scaled by 1/world before the backward pass (every parameter gradient is
linear in it) and the buckets are SUM-reduced -- for power-of-two world sizes bit-identical to DDP's
"divide, then sum", and it works on every backend (gloo has no AVG).
Under stock DDP everything still works (tests/test_gpu_ddp.py); this wrapper is the fast path bench.py uses.
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from . import lib as L
from . import ops

_ALIGN = 64          # floats: every gradient slice starts on a 256-byte boundary (dwordx4 kernels, RCCL chunking)


class GradSink:
    """Flat gradient buffer + bucketed asynchronous all-reduce for ONE engine (see the module docstring).

    Protocol (called by dasac_hip.engine._PlanFunction.backward / Engine.backward):
        grad_out = sink.begin(engine, need, grad_out)     # new flat buffer for this pass; returns grad_out / world
        t = sink.alloc(j)                                  # destination of parameter j's gradient (shape of the parameter)
        sink.done([j, ...])                                # these gradients are complete (called in backward order)
        sink.finish()                                      # launch what is left, make the launch stream wait for all
    """

    def __init__(self, process_group=None, bucket_bytes=32 << 20, reduce_single_rank=False):
        self.time_exposed = False                 # bench: time the stream waits at the end of every backward pass
        self._exposed_events = []
        self._device = None
        self.pg = process_group
        self.bucket_bytes = int(bucket_bytes)
        self.reduce_single_rank = bool(reduce_single_rank)
        self._layout_engine, self._layout_key = None, None
        self._offsets, self._shapes, self._bucket_of, self._buckets = None, None, None, None
        self._flat, self._pending, self._works, self._scale = None, None, [], {}
        self.launched, self.launched_early = 0, 0   # statistics (tests): reductions issued / issued before the backward ended

    def world(self):
        return dist.get_world_size(self.pg) if (dist.is_available() and dist.is_initialized()) else 1

    # ------------------------------------------------------------------ layout
    def _build_layout(self, engine, need):
        """Slices in backward completion order (ops reversed, an op's parameters together); buckets of >= bucket_bytes,
        the first one a quarter of that so that the first reduction starts early (DDP does the same with 1 MB)."""
        offsets, shapes, bucket_of, buckets = {}, {}, {}, []
        off, b_start, b_members, limit = 0, 0, [], max(self.bucket_bytes // 4, 1)
        for op in reversed(engine.plan.ops):
            idx = [j for j in getattr(op, "pidx", []) if need[j]]
            if not idx:
                continue
            for j in idx:
                p = engine.params[j]
                offsets[j], shapes[j] = off, tuple(p.shape)
                off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
                b_members.append(j)
            if (off - b_start) * 4 >= limit:
                buckets.append((b_start, off, list(b_members)))
                b_start, b_members, limit = off, [], self.bucket_bytes
        if b_members:
            buckets.append((b_start, off, list(b_members)))
        for b, (_, _, members) in enumerate(buckets):
            for j in members:
                bucket_of[j] = b
        self._offsets, self._shapes, self._bucket_of, self._buckets, self._total = offsets, shapes, bucket_of, buckets, off

    def begin(self, engine, need, grad_out):
        # keyed on the engine OBJECT (held here, compared by identity): BaseNet rebuilds its engine when it is stale and a
        # freed engine's id() can be handed to the new one -- an id-keyed cache would then reuse offsets of another plan
        key = tuple(bool(n) for n in need)
        if self._layout_engine is not engine or self._layout_key != key:
            self._build_layout(engine, need)
            self._layout_engine, self._layout_key = engine, key
        # A FRESH buffer per backward pass: the gradients of an earlier pass (still referenced by .grad or set aside by
        # FusedSGD.stash_grads) keep their own buffer alive; the caching allocator recycles it once they are gone.
        self._flat = torch.empty(max(self._total, 1), dtype=torch.float32, device=grad_out.device)
        self._pending = [len(m) for (_, _, m) in self._buckets]
        self._written = set()
        self._launches = [0] * len(self._buckets)
        self._works = []
        self._device = grad_out.device
        world = self.world()
        if world > 1:
            # average = sum of (gradient / world): scale the root of the backward pass once (exact for power-of-two worlds)
            key = (grad_out.shape[0] * grad_out.shape[1], grad_out.device.index, world)
            sc = self._scale.get(key)
            if sc is None:
                sc = torch.full((key[0],), 1.0 / world, dtype=torch.float32, device=grad_out.device)
                self._scale[key] = sc
            grad_out = ops.scale_planes(grad_out.contiguous(), sc)
        return grad_out

    def alloc(self, j):
        o = self._offsets.get(j)
        if o is None:
            return None
        n = 1
        for d in self._shapes[j]:
            n *= d
        return self._flat[o:o + n].view(self._shapes[j])

    def _launch(self, b, early):
        self._pending[b] = -1
        self._launches[b] += 1
        world = self.world()
        if world == 1 and not self.reduce_single_rank:
            return
        lo, hi, _ = self._buckets[b]
        self._works.append(dist.all_reduce(self._flat[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        self.launched += 1
        self.launched_early += int(early)

    def done(self, indices):
        for j in indices:
            b = self._bucket_of.get(j)
            if b is None or self._pending[b] < 0 or j in self._written:
                continue
            self._written.add(j)
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._launch(b, True)

    def finish(self):
        for b, left in enumerate(self._pending):
            if left >= 0:                          # a layer that produced no gradient in this pass: reduce what is there
                if left > 0:
                    # ONLY the slices nobody wrote (their parameters' grads stay None, but the collective reads them) are
                    # cleared -- the bucket's other members already hold gradients that autograd was handed as views
                    for j in self._buckets[b][2]:
                        if j not in self._written:
                            self.alloc(j).zero_()
                self._launch(b, False)
        assert all(n == 1 for n in self._launches), "GradSink: a bucket was reduced {} times in one pass".format(self._launches)
        # what the launch stream still has to wait for when the backward pass is over = the EXPOSED part of the reduction;
        # timed with an event pair on that stream when asked for (bench.py --gpus N reports it per rank)
        timed = self.time_exposed and self._works and self._device is not None and self._device.type == "cuda"
        if timed:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w in self._works:
            w.wait()                               # nccl: the CURRENT STREAM waits (no host block); gloo: host wait
        if timed:
            ev[1].record()
            self._exposed_events.append(ev)
        self._works = []
        self._flat = None                          # the views handed out keep the storage alive as long as needed


def _exposed_ms(sink):
    """Milliseconds the launch stream waited for outstanding reductions since the last call (synchronises the events)."""
    total = 0.0
    for a, b in sink._exposed_events:
        b.synchronize()
        total += a.elapsed_time(b)
    sink._exposed_events = []
    return total


def _trainable_backbones(module):
    from models.basenet import BaseNet             # drop-in package (da-sac_amd/models)
    out = []
    for m in module.modules():
        if isinstance(m, BaseNet) and hasattr(m, "_plan") and type(m)._plan is not BaseNet._plan \
                and any(p.requires_grad for p in m.parameters(recurse=True)):
            out.append(m)
    return out


class OverlappedDataParallel(nn.Module):
    """DistributedDataParallel's contract on the fused engine (see the module docstring).

    module          the model (SAC / SAC_Baseline), already on its device
    process_group   group for the gradient reduction and the buffer broadcast (default: the world)
    bucket_mb       bucket size of the overlapped reduction
    broadcast_buffers   re-broadcast rank 0's buffers before each forward (DDP default); buffers
                    named in `module._ddp_params_and_buffers_to_ignore` are synchronised ONCE at construction and then left
                    alone (frozen-BN statistics: re-sending them would only invalidate the engine's folded-weight caches)
    """

    def __init__(self, module, device_ids=None, process_group=None, bucket_mb=32, broadcast_buffers=True,
                 reduce_single_rank=False):
        super().__init__()
        self.module = module
        self.process_group = process_group
        self.broadcast_buffers = bool(broadcast_buffers)
        # CUs left to the collective's kernels, which run beside the backward GEMMs: the persistent stream-K grid gives every worker
        # the same matrix work, so a CU that also hosts RCCL workgroups finishes last.  8 by default under world > 1
        # (DASAC_SK_RESERVE_CUS overrides, 0 = none); a single rank keeps the whole chip.
        # The setting is process-global in the library (one chip, one set of CUs): every launch of this process shrinks its
        # stream-K grid while a wrapper with world > 1 is alive -- a different (tile, K-range) partition, i.e. a different
        # summation order than a single-rank run (results agree to fp32 rounding, not bit for bit).  The previous value comes
        # back when the wrapper is closed or collected.
        self._prev_reserved = None
        if self._world() > 1 and "DASAC_SK_RESERVE_CUS" not in os.environ:
            self._prev_reserved = L.load().dasac_set_reserved_cus(8)
        self._sinks = []
        for net in _trainable_backbones(module):
            sink = GradSink(process_group, int(bucket_mb) << 20, reduce_single_rank)
            net._grad_sink = sink
            self._sinks.append(sink)
        self._covered = None                      # checked lazily: the engines are built by the first forward
        ignore = set(getattr(module, "_ddp_params_and_buffers_to_ignore", []))
        self._synced_buffers = [b for n, b in module.named_buffers() if n not in ignore]
        self._sync_module_states()

    def _world(self):
        return dist.get_world_size(self.process_group) if (dist.is_available() and dist.is_initialized()) else 1

    def close(self):
        """Gives the reserved CUs back (idempotent).  Called by __del__; call it yourself when the wrapper is dropped while the
        process keeps computing with the bare module."""
        prev, self._prev_reserved = getattr(self, "_prev_reserved", None), None
        if prev is not None:
            try:
                L.load().dasac_set_reserved_cus(prev)
            except Exception:      # interpreter shutdown: the library may be gone
                pass

    def __del__(self):
        self.close()

    @torch.no_grad()
    def _broadcast(self, tensors):
        """Rank 0's values into every rank's tensors, one collective per dtype (coalesced like DDP's)."""
        if self._world() == 1 or not tensors:
            return
        src = dist.get_global_rank(self.process_group, 0) if self.process_group is not None else 0
        rank = dist.get_rank(self.process_group)
        by_dtype = {}
        for t in tensors:
            by_dtype.setdefault(t.dtype, []).append(t)
        for dtype, group in by_dtype.items():
            flat = torch.cat([t.detach().reshape(-1) for t in group])
            dist.broadcast(flat, src, group=self.process_group)
            if rank != 0:                         # rank 0 keeps its tensors untouched (no version bump, no cache invalidation)
                o = 0
                for t in group:
                    n = t.numel()
                    t.detach().copy_(flat[o:o + n].view(t.shape))
                    o += n

    def _sync_module_states(self):
        """Construction time: EVERY parameter and buffer takes rank 0's value -- including the frozen-BN statistics that
        are exempt from the per-forward broadcast (ranks that loaded different snapshots would otherwise diverge silently)."""
        self._broadcast([p for p in self.module.parameters()] + [b for b in self.module.buffers()])

    def _check_coverage(self):
        """Every trainable parameter must be reduced by somebody.  Stock DDP reduces whatever requires grad; this wrapper only
        what the sunk engines differentiate -- a trainable parameter outside their plans (a new head outside BaseNet, a
        backbone without `_plan`, a `requires_grad` flip after construction) would silently diverge between ranks."""
        trainable = {id(p): n for n, p in self.module.named_parameters() if p.requires_grad}
        key = tuple(sorted(trainable))
        if self._covered == key:
            return
        from models.basenet import BaseNet
        reduced = set()
        for net in self.module.modules():
            if isinstance(net, BaseNet) and net._grad_sink is not None:
                if net._engine is None or net._engine.stale():
                    return                       # plan not captured yet: checked again at the next forward
                reduced.update(id(p) for p in net._engine.params)
        missing = [n for i, n in trainable.items() if i not in reduced]
        if missing:
            raise RuntimeError("OverlappedDataParallel: trainable parameters outside every engine plan would never be "
                               "all-reduced: {} (wrap the model in torch's DistributedDataParallel instead)".format(missing[:4]))
        self._covered = key

    def time_exposed_reduction(self, on=True):
        for sk in self._sinks:
            sk.time_exposed = bool(on)
            sk._exposed_events = []

    def exposed_reduction_ms(self):
        """Stream time spent waiting for gradient reductions at the end of the backward passes since the last call."""
        return sum(_exposed_ms(sk) for sk in self._sinks)

    def forward(self, *args, **kwargs):
        self._check_coverage()
        if self.broadcast_buffers:            # DDP syncs its buffers before EVERY forward (train, eval and no-grad alike)
            self._broadcast(self._synced_buffers)
        return self.module(*args, **kwargs)

    def forward_fused(self, *args, **kwargs):
        """`SAC.forward_fused` (both student passes of an iteration as one) under the same buffer synchronisation; the one
        backward pass that follows reduces source + target gradients together."""
        self._check_coverage()
        if self.broadcast_buffers:
            self._broadcast(self._synced_buffers)
        return self.module.forward_fused(*args, **kwargs)
