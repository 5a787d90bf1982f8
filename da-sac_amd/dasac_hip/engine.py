"""Fused forward/backward executor for the segmentation backbones.

A network (`models.deeplabv2`, `models.fcn`) describes itself once as a flat *plan* of fused ops
over numbered activation slots; this module runs the plan forward and -- hand-written, no autograd
tape inside the backbone -- backward, calling only the C-ABI kernels of libdasac_hip.so:

  ConvOp   implicit-GEMM convolution with the frozen-BN scale/shift (or bias), residual add and ReLU
           in the epilogue ("ABN": conv+BN+ReLU fused).  Several branches (ASPP) are one contraction.
           Backward: channel sums (d beta), split-K weight-gradient GEMM (+ the d gamma dot term),
           data-gradient GEMM whose epilogue accumulates into the producer's gradient and applies the
           producer's ReLU mask -- so no stand-alone elementwise pass exists in the backbone.
  PoolOp   max pooling with the ReLU backward of its producer folded into the gradient routing.
  Up2AddOp FCN-8s skip fusion: up_x2(a) + b.          ScaleOp: Dropout2d with an explicit mask.

The whole plan is exposed to PyTorch as ONE autograd.Function (inputs: image + all parameters), so
DistributedDataParallel sees ordinary parameter gradients.
"""
import torch

from . import ops
from . import lib as L


class ConvOp:
    kind = "conv"

    def __init__(self, src, dst, convs, bn, relu, res):
        c0 = convs[0]
        self.src, self.dst, self.res, self.relu = src, dst, res, bool(relu)
        self.convs, self.bn = list(convs), bn
        assert all(c.stride == c0.stride and c.in_channels == c0.in_channels and c.out_channels == c0.out_channels
                   and c.groups == 1 for c in convs)
        assert c0.stride[0] == c0.stride[1]
        branches = []
        for c in convs:
            assert c.dilation[0] == c.dilation[1] and c.padding[0] == c.padding[1]
            branches.append((c.kernel_size[0], c.kernel_size[1], c.dilation[0], c.padding[0]))
        self.spec = ops.ConvSpec(c0.in_channels, c0.out_channels, branches, c0.stride[0])
        self.has_bias = c0.bias is not None
        assert bn is None or len(convs) == 1
        # few output channels x many taps (ASPP): evaluate as dense 1x1 GEMMs over taps*Cp channels
        self.expanded = ops.ExpandedConv(self.spec) if (len(convs) > 1 and c0.out_channels <= 32 and c0.stride[0] == 1) else None

    def owners(self):
        """(module, attribute) of every parameter, in the order of params()."""
        out = [(c, "weight") for c in self.convs]
        if self.has_bias:
            out += [(c, "bias") for c in self.convs]
        if self.bn is not None:
            out += [(self.bn, "weight"), (self.bn, "bias")]
        return out

    def params(self):
        return [getattr(m, a) for m, a in self.owners()]

    def geometry(self):
        return tuple((c.kernel_size, c.dilation, c.padding, c.stride, c.bias is not None) for c in self.convs)


class PoolOp:
    kind = "pool"

    def __init__(self, src, dst, k, s, p, ceil_mode):
        self.src, self.dst, self.k, self.s, self.p, self.ceil = src, dst, k, s, p, bool(ceil_mode)

    def params(self):
        return []


class Up2AddOp:
    kind = "up2add"

    def __init__(self, src, skip, dst):
        self.src, self.skip, self.dst = src, skip, dst

    def params(self):
        return []


class ScaleOp:
    """Dropout2d(p) in train mode: y = x * keep/(1-p) per (n, c) plane."""
    kind = "drop"

    def __init__(self, src, dst, module):
        self.src, self.dst, self.module = src, dst, module

    def params(self):
        return []


class Plan:
    """Builder used by the model classes."""

    def __init__(self):
        self.ops, self.n_slots, self.output = [], 1, None     # slot 0 = input image

    def _new(self):
        self.n_slots += 1
        return self.n_slots - 1

    def conv(self, src, conv, bn=None, relu=False, res=None):
        dst = self._new()
        self.ops.append(ConvOp(src, dst, [conv], bn, relu, res))
        return dst

    def conv_sum(self, src, convs):
        dst = self._new()
        self.ops.append(ConvOp(src, dst, list(convs), None, False, None))
        return dst

    def maxpool(self, src, k, s, p=0, ceil_mode=False):
        dst = self._new()
        self.ops.append(PoolOp(src, dst, k, s, p, ceil_mode))
        return dst

    def up2_add(self, src, skip):
        dst = self._new()
        self.ops.append(Up2AddOp(src, skip, dst))
        return dst

    def dropout2d(self, src, module):
        dst = self._new()
        self.ops.append(ScaleOp(src, dst, module))
        return dst

    def finish(self, output):
        self.output = output
        return self


def _ver(t):
    return (t.data_ptr(), t._version)


def _copy_into(out, src):
    """A second parameter that receives the same gradient (the bias of every ASPP branch): its own tensor."""
    if out is None:
        return src.clone()
    out.view(-1).copy_(src.view(-1))
    return out


class Engine:
    """Executes a Plan.  Keeps per-layer caches of gather tables (by spatial size) and of packed
    weights / folded BN vectors (invalidated by the parameters' version counters)."""

    def __init__(self, plan):
        self.plan = plan
        self.params, self._owners, self._geometry = [], [], []
        for op in plan.ops:
            op.pidx = []
            for p in op.params():
                op.pidx.append(len(self.params))
                self.params.append(p)
            if op.kind == "conv":
                self._owners += op.owners()
                self._geometry.append((op, op.geometry()))
        self.consumers = [0] * plan.n_slots
        self.producer = [None] * plan.n_slots
        for i, op in enumerate(plan.ops):
            self.producer[op.dst] = op
            for s in self._inputs(op):
                self.consumers[s] += 1
        self.last_use = [0] * plan.n_slots
        for i, op in enumerate(plan.ops):
            for s in self._inputs(op):
                self.last_use[s] = i
        self.last_use[plan.output] = len(plan.ops)
        self._tables, self._packs, self._folds = {}, {}, {}
        self._refresh, self._fold_bufs = {}, {}
        # A ReLU conv's pattern is recorded as bits when the backward pass will apply it in a stride-1 data-gradient epilogue:
        # the mask is applied by the consumer that finishes LAST in backward order = the slot's FIRST consumer in plan order.
        self._bits_wanted = [False] * plan.n_slots
        first_consumer = {}
        for op in plan.ops:
            for s_ in self._inputs(op):
                first_consumer.setdefault(s_, (op, s_ == op.src))
        for slot, (op, is_src) in first_consumer.items():
            self._bits_wanted[slot] = bool(is_src and op.kind == "conv" and op.spec.stride == 1 and slot != 0)

    def stale(self):
        """True when a module no longer holds the Parameter objects (or conv geometry) this engine captured:
        `load_state_dict(assign=True)`, `m.weight = nn.Parameter(...)`, parametrizations, edited dilation/padding.
        In-place updates (optimisers, copy_) are NOT stale -- the caches follow the version counters."""
        return any(getattr(m, a) is not p for (m, a), p in zip(self._owners, self.params)) or \
            any(op.geometry() != geo for op, geo in self._geometry)

    @staticmethod
    def _inputs(op):
        if op.kind == "conv":
            return [op.src] + ([op.res] if op.res is not None else [])
        if op.kind == "up2add":
            return [op.src, op.skip]
        return [op.src]

    # ---------------------------------------------------------------- caches
    def table(self, op, h, w, transposed, device, wgrad=False):
        """Gather table; the forward / data-gradient GEMMs may use the chunk-major K order, the weight
        gradient always the tap-major one."""
        order = 0 if wgrad else ops.gemm_order(op.spec, transposed)
        key = (id(op), h, w, transposed, device.index, order)
        t = self._tables.get(key)
        if t is None:
            t = ops.conv_table(op.spec, h, w, transposed, device, order)
            self._tables[key] = t
        return t

    def fold(self, op):
        """(scale, shift, invstd) of the frozen BN (plus conv bias), or (None, bias_sum, None)."""
        if op.bn is None:
            if not op.has_bias:
                return None, None, None
            if len(op.convs) == 1:
                return None, op.convs[0].bias.detach(), None
            key = ("bsum", id(op)) + tuple(_ver(c.bias) for c in op.convs)
            ent = self._folds.get(id(op))
            if ent is None or ent[0] != key:
                acc = op.convs[0].bias.detach()
                for c in op.convs[1:]:
                    acc = ops.add(acc, c.bias.detach())
                ent = (key, (None, acc, None))
                self._folds[id(op)] = ent
            return ent[1]
        bn, cb = op.bn, (op.convs[0].bias if op.has_bias else None)
        key = (_ver(bn.weight), _ver(bn.bias), _ver(bn.running_mean), _ver(bn.running_var), None if cb is None else _ver(cb))
        ent = self._folds.get(id(op))
        if ent is None or ent[0] != key:
            ent = (key, ops.bn_fold(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps,
                                    None if cb is None else cb.detach()))
            self._folds[id(op)] = ent
        return ent[1]

    def table_e(self, op, h, w, transposed, device):
        key = (id(op), "e", h, w, transposed, device.index)
        t = self._tables.get(key)
        if t is None:
            t = ops.conv_table(op.expanded.spec1, h, w, transposed, device)
            self._tables[key] = t
        return t

    def packed_e(self, op, transposed):
        key = tuple(_ver(c.weight) for c in op.convs) + (ops.PRECISION,)
        slot = (id(op), "e", transposed)
        ent = self._packs.get(slot)
        if ent is None or ent[0] != key:
            buf = op.expanded.pack([c.weight.detach() for c in op.convs], transposed, out=None if ent is None else ent[1])
            ent = (key, buf)
            self._packs[slot] = ent
        return ent[1]

    def packed(self, op, transposed, scale=None):
        key = tuple(_ver(c.weight) for c in op.convs) + ((_ver(scale),) if scale is not None else ()) + (ops.PRECISION,)
        slot = (id(op), transposed)
        ent = self._packs.get(slot)
        if ent is None or ent[0] != key:
            buf = None if ent is None else ent[1]
            buf = ops.conv_pack(op.spec, [c.weight.detach() for c in op.convs], transposed, scale, out=buf,
                                order=ops.gemm_order(op.spec, transposed))
            ent = (key, buf, scale)      # keep `scale` alive: its data_ptr is part of the key
            self._packs[slot] = ent
        return ent[1]

    def largest_tensor_bytes(self, Nb, H, W):
        """Bytes of the largest activation (or expanded-conv intermediate) a pass over an [Nb, *, H, W] input creates.  The conv
        kernels address tensors through buffer descriptors with unsigned 32-bit byte offsets: every one of them has to stay below 4 GiB
        (4 GiB - 4 KiB; rounds 1-4: 2 GiB)."""
        shapes = {0: (None, H, W)}
        biggest = 0
        for op in self.plan.ops:
            _, h, w = shapes[op.src]
            if op.kind == "conv":
                oh, ow = op.spec.out_hw(h, w)
                c = op.spec.cout
                if op.expanded is not None:
                    biggest = max(biggest, Nb * op.expanded.E * oh * ow * 4)
            elif op.kind == "pool":
                oh, ow, c = ops.pool_out(h, op.k, op.s, op.p, op.ceil), ops.pool_out(w, op.k, op.s, op.p, op.ceil), shapes[op.src][0]
            elif op.kind == "up2add":
                oh, ow, c = 2 * h, 2 * w, shapes[op.src][0]
            else:
                oh, ow, c = h, w, shapes[op.src][0]
            shapes[op.dst] = (c, oh, ow)
            biggest = max(biggest, Nb * (c or 1) * oh * ow * 4)
        return biggest

    # ---------------------------------------------------------------- whole-network refresh of folds / packs
    def _refresh_plan(self, device, transposed):
        """Device job tables for `ops.refresh_network`: one fold job per frozen-BN conv, one pack job per (conv, layout) for
        every single-branch, non-expanded conv.  Output buffers are persistent (their pointers sit in the tables)."""
        # batch-statistics BN layers (baseline / AdaBN mode) fold nothing into the weights: their convs are packed un-scaled
        bn_mode = tuple(bool(op.bn is not None and op.bn.training) for op in self.plan.ops if op.kind == "conv")
        key = (device.index, bool(transposed), bn_mode)
        plan = self._refresh.get(key)
        if plan is not None and plan["ptrs"] == self._refresh_ptrs():
            return plan
        lib = L.load()
        fold_jobs, pack_jobs, fold_ops, pack_slots = [], [], [], []
        for op in self.plan.ops:
            if op.kind != "conv" or op.expanded is not None or len(op.convs) != 1:
                continue
            scale = None
            if op.bn is not None and not op.bn.training:
                ent = self._fold_bufs.get(id(op))
                if ent is None:
                    C = op.spec.cout
                    ent = tuple(torch.empty(C, dtype=torch.float32, device=device) for _ in range(3))
                    self._fold_bufs[id(op)] = ent
                scale = ent[0]
                cb = op.convs[0].bias if op.has_bias else None
                fold_jobs.append((op.bn.weight, op.bn.bias, op.bn.running_mean, op.bn.running_var, cb, ent, op.bn.eps, op.spec.cout))
                fold_ops.append(op)
            for tr in ((False, True) if transposed else (False,)):
                M = op.spec.cin if tr else op.spec.cout
                K = op.spec.Kt if tr else op.spec.K
                slot = (id(op), tr)
                old = self._packs.get(slot)
                shape = (lib.dasac_conv_kpad(K), lib.dasac_conv_mpad(M))
                buf = old[1] if (old is not None and tuple(old[1].shape) == shape and not getattr(old[1], "dasac_x3", False)) else \
                    torch.empty(shape, dtype=torch.float32, device=device)
                buf.dasac_x3 = False
                pack_jobs.append((op.convs[0].weight, scale, buf, op.spec.cout, op.spec.cin, op.spec.taps, shape[1], shape[0], int(tr),
                                  ops.gemm_order(op.spec, tr)))
                pack_slots.append((slot, op, buf, scale))
        plan = {"ptrs": self._refresh_ptrs(), "fold_ops": fold_ops, "pack_slots": pack_slots,
                "tables": ops.build_refresh_tables(fold_jobs, pack_jobs, device)}
        self._refresh[key] = plan
        return plan

    def _refresh_ptrs(self):
        out = []
        for op in self.plan.ops:
            if op.kind == "conv":
                out += [p.data_ptr() for p in op.params()]
                if op.bn is not None:
                    out += [op.bn.running_mean.data_ptr(), op.bn.running_var.data_ptr()]
        return tuple(out)

    def refresh(self, device, transposed):
        """After an optimiser / EMA step every folded BN vector and every packed weight operand of the network is stale at
        once: rebuild them ALL in two launches (dasac_bn_fold_multi, dasac_conv_pack_multi) instead of one small launch per
        layer and layout as they are first used (~310 launches per student step).  Only when the whole network is stale and
        runs fp32 -- partial invalidations and the split-bf16 operands keep the per-layer path.  Convolutions in front of a
        batch-statistics BN (round 4: cfg-2 spent 1.5 ms per step in 228 per-layer pack launches) are packed un-scaled by the
        same launch; only frozen BNs have a fold job."""
        if ops.PRECISION != "fp32":
            return
        convs = [op for op in self.plan.ops if op.kind == "conv" and op.expanded is None and len(op.convs) == 1]
        if len(convs) < 8:
            return
        frozen = lambda op: op.bn is not None and not op.bn.training
        # stale = the cached key no longer matches the parameters' version counters
        def pack_key(op, scale):
            return (_ver(op.convs[0].weight),) + ((_ver(scale),) if scale is not None else ()) + (ops.PRECISION,)

        def fold_key(op):
            bn, cb = op.bn, (op.convs[0].bias if op.has_bias else None)
            return (_ver(bn.weight), _ver(bn.bias), _ver(bn.running_mean), _ver(bn.running_var), None if cb is None else _ver(cb))
        for op in convs:                                     # all-or-nothing: the first fresh layer ends the check
            ent = self._packs.get((id(op), False))
            sc = self._folds.get(id(op)) if frozen(op) else None
            fresh_fold = (not frozen(op)) or (sc is not None and sc[0] == fold_key(op))
            if fresh_fold and ent is not None and ent[0] == pack_key(op, sc[1][0] if frozen(op) else None):
                # (the forward layout may have been refreshed alone by a no-grad pass -- AdaBN's target forward right after the
                # optimiser step: the data-gradient layout is then still stale, and one more launch beats 104 per-layer ones)
                ent_t = self._packs.get((id(op), True)) if transposed else ent
                if ent_t is not None and ent_t[0] == ent[0]:
                    return
        plan = self._refresh_plan(device, transposed)
        ops.refresh_network(plan["tables"])
        for op in plan["fold_ops"]:
            ent = self._fold_bufs[id(op)]
            ops.bump_versions([ent[0]])                      # the scale vector is part of the pack keys: it has new contents
            self._folds[id(op)] = (fold_key(op), ent)
        for slot, op, buf, scale in plan["pack_slots"]:
            self._packs[slot] = (pack_key(op, scale), buf, scale)

    # ---------------------------------------------------------------- forward
    def forward(self, x, keep):
        """Runs the plan.  keep=True retains what backward needs; returns (output, saved)."""
        L.require_gpu(x)
        self.refresh(x.device, transposed=keep)
        acts = {0: x}
        saved = {"acts": acts, "aux": {}, "bits": {}}
        for i, op in enumerate(self.plan.ops):
            xin = acts[op.src]
            if op.kind == "conv":
                Nb, _, H, W = xin.shape
                OH, OW = op.spec.out_hw(H, W)
                if op.expanded is not None:
                    _, bias_sum, _ = self.fold(op)
                    acts[op.dst] = op.expanded.forward(xin, self.packed_e(op, False), self.table_e(op, H, W, False, xin.device), bias_sum)
                    if not keep:
                        for s_ in self._inputs(op):
                            if self.last_use[s_] == i and s_ != 0:
                                del acts[s_]
                    continue
                out = torch.empty((Nb, op.spec.cout, OH, OW), dtype=torch.float32, device=xin.device)
                if op.bn is not None and op.bn.training:
                    # batch-statistics BN (baseline / AdaBN mode): raw conv, then stats -> normalise(+res)(+ReLU)
                    cb = op.convs[0].bias.detach() if op.has_bias else None
                    # the GEMM epilogue leaves the per-tile channel sums / sums of squares next to z: no statistics pass over z
                    ts = ops.tile_stats_buffer(Nb, op.spec.cout, OH, OW, xin.device) if ops.stats_ok(op.spec.cout, op.spec.cin) else None
                    ops.conv_gemm(xin, self.packed(op, False, None), self.table(op, H, W, False, xin.device), out, (OH, OW),
                                  op.spec.stride, op.spec.cout, op.spec.K, 1, cb, None, None, False, stats=ts)
                    z = out
                    out, stats = ops.bn_train_forward(z, op.bn, None if op.res is None else acts[op.res], op.relu, tile_stats=ts)
                    if keep:
                        saved["aux"][i] = (z, stats)
                else:
                    scale, shift, _ = self.fold(op)
                    bits = None
                    if keep and op.relu and self._bits_wanted[op.dst] and ops.bits_ok(op.spec.cout, op.spec.cin):
                        bits = ops.ReluBits(Nb, op.spec.cout, OH, OW, xin.device)
                        saved["bits"][op.dst] = bits
                    ops.conv_gemm(xin, self.packed(op, False, scale), self.table(op, H, W, False, xin.device), out, (OH, OW),
                                  op.spec.stride, op.spec.cout, op.spec.K, 1, shift,
                                  None if op.res is None else acts[op.res], None, op.relu, bits_out=bits)
            elif op.kind == "pool":
                out, arg = ops.maxpool_fwd(xin, op.k, op.s, op.p, op.ceil)
                if keep:
                    saved["aux"][i] = arg
            elif op.kind == "up2add":
                up, _, _ = ops.upsample_softmax(xin, (2 * xin.shape[2], 2 * xin.shape[3]))
                out = ops.add(up, acts[op.skip], out=up)
            elif op.kind == "drop":
                m = op.module
                if m.training and m.p > 0:
                    # ATen feature dropout (fcn.py:52,56): per (n, c) plane noise = bernoulli(1-p)/(1-p).  A test can pin
                    # the draw by setting `module.keep_mask` ([B,C], already divided by 1-p) -- parity needs equal masks.
                    keep_mask = getattr(m, "keep_mask", None)
                    if keep_mask is None:
                        keep_mask = ops.dropout_planes(xin.shape[0], xin.shape[1], m.p, xin.device)
                    assert tuple(keep_mask.shape) == tuple(xin.shape[:2]) and keep_mask.is_cuda
                    out = ops.scale_planes(xin, keep_mask)
                    if keep:
                        saved["aux"][i] = keep_mask
                else:
                    out = xin
            else:
                raise AssertionError(op.kind)
            acts[op.dst] = out
            if not keep:
                for s in self._inputs(op):
                    if self.last_use[s] == i and s != 0:
                        del acts[s]
        return acts[self.plan.output], saved

    # ---------------------------------------------------------------- backward
    def backward(self, saved, grad_out, need, trace=None, sink=None):
        """grad_out: gradient w.r.t. the plan output.  need[i]: whether parameter i wants a gradient.
        Returns the list of parameter gradients (None where not needed).  `trace` (debug): dict that
        receives the finished activation gradient of every slot.

        `sink` (dasac_hip.parallel.GradSink or None): where parameter gradients are WRITTEN and who is told when a layer's
        are complete -- `sink.alloc(j)` returns the destination of parameter j (a slice of one flat reduction buffer),
        `sink.done(indices)` is called after each op, in backward order, so that a bucket's all-reduce can start while
        the layers below it are still being differentiated (DistributedDataParallel's overlap, train.py:104,133,232)."""
        acts, aux = saved["acts"], saved["aux"]
        bits = saved.get("bits", {})
        grads = [None] * len(self.params)

        def relu_pattern(slot, M, Cx):
            """What the data-gradient epilogue (output M channels, gathering Cx) masks with: the producer's bit mask when the
            forward recorded one and this GEMM has the bit-mask variant, else the producer's fp32 output."""
            b = bits.get(slot)
            return b if (b is not None and ops.bits_ok(M, Cx)) else acts[slot]

        def dest(j):
            return sink.alloc(j) if (sink is not None and need[j]) else None

        def sums_dest(b_idx, is_bias_grad, cout, device):
            """Per-channel sums of dz: they ARE the bias gradient of a conv without (or with batch-statistics) BN -- then they
            are written straight to that gradient's destination -- and an intermediate of bn_param_grads otherwise."""
            wanted = [j for j in b_idx if need[j]]
            out = dest(wanted[0]) if (is_bias_grad and wanted) else None
            return torch.empty(cout, dtype=torch.float32, device=device) if out is None else out.view(cout)

        g = {self.plan.output: grad_out.contiguous()}
        # the d-gamma dot terms of all frozen-BN convs are slices of ONE vector: [dot_rows, cout] partial rows per layer, every
        # element written by the weight-gradient finish and added in a fixed order by bn_param_grads (no atomics, no fill)
        dot_pool, dot_used = None, 0
        dot_total = sum(ops.dot_rows(op.spec) * op.spec.cout for op in self.plan.ops
                        if op.kind == "conv" and op.bn is not None and op.expanded is None)
        pending = list(self.consumers)
        pending[self.plan.output] = 0

        def relu_producer(slot):
            p = self.producer[slot]
            return p is not None and p.kind == "conv" and p.relu

        def join_identity(slot, t):
            """A pass-through consumer (residual / skip) hands its gradient to `slot`."""
            pending[slot] -= 1
            cur = g.get(slot)
            t = t if cur is None else ops.add(cur, t)
            if pending[slot] == 0 and relu_producer(slot):
                t = ops.relu_mask(t, acts[slot])
            g[slot] = t

        for i in range(len(self.plan.ops) - 1, -1, -1):
            op = self.plan.ops[i]
            dz = g.pop(op.dst, None)
            if dz is None:
                continue
            assert pending[op.dst] == 0
            if trace is not None:
                trace[op.dst] = dz
            xin = acts[op.src]
            if op.kind == "conv":
                spec = op.spec
                Nb, _, H, W = xin.shape
                if op.expanded is not None:
                    ex, nw_ = op.expanded, len(op.convs)
                    d = ex.scatter(dz)
                    if any(need[j] for j in op.pidx[:nw_]):
                        dws = ex.wgrad(d, xin, [c.weight.detach() for c in op.convs], self.table_e(op, H, W, False, xin.device),
                                       outs=[dest(j) for j in op.pidx[:nw_]])
                        for j, dw in zip(op.pidx[:nw_], dws):
                            if need[j]:
                                grads[j] = dw
                    if op.has_bias and any(need[j] for j in op.pidx[nw_:]):
                        wanted = [j for j in op.pidx[nw_:] if need[j]]
                        sums = ops.channel_sums(dz, out=dest(wanted[0]))
                        for n_, j in enumerate(wanted):              # every branch bias sees the same gradient
                            grads[j] = sums if n_ == 0 else _copy_into(dest(j), sums)
                    if op.src != 0:
                        pending[op.src] -= 1
                        last = pending[op.src] == 0
                        mask = relu_pattern(op.src, spec.cin, ex.E) if (last and relu_producer(op.src)) else None
                        g[op.src] = ex.dgrad(d, self.packed_e(op, True), self.table_e(op, H, W, True, dz.device), (H, W),
                                             res=g.get(op.src), mask=mask)
                    acts.pop(op.dst, None)
                    if sink is not None:
                        sink.done([j for j in op.pidx if need[j]])
                    continue
                dy_out = dz                       # gradient w.r.t. the op output (what a residual input receives)
                train_bn = i in aux
                nw = len(op.convs)
                if train_bn:
                    z, stats = aux.pop(i)
                    bn_tail = op.pidx[nw + (nw if op.has_bias else 0):]
                    bn_need = any(need[j] for j in bn_tail)
                    dz, dg, db = ops.bn_train_backward(dz, z, stats, op.bn.weight.detach(), want_params=bn_need,
                                                       outs=(dest(bn_tail[0]), dest(bn_tail[1])) if bn_need else (None, None))
                    del z
                    scale, shift, invstd = None, None, None
                else:
                    scale, shift, invstd = self.fold(op)
                w_need = [need[j] for j in op.pidx[:nw]]
                rest = op.pidx[nw:]
                b_idx = rest[:nw] if op.has_bias else []
                bn_idx = rest[len(b_idx):]
                want_bn = (not train_bn) and op.bn is not None and any(need[j] for j in bn_idx)
                want_bias = any(need[j] for j in b_idx)
                if train_bn and bn_need:
                    grads[bn_idx[0]], grads[bn_idx[1]] = dg, db
                sums, dot = None, None
                if any(w_need) or want_bn:
                    if want_bn:
                        if dot_pool is None:
                            dot_pool = torch.empty(dot_total, dtype=torch.float32, device=dz.device)
                        rows = ops.dot_rows(spec)
                        dot = dot_pool[dot_used:dot_used + rows * spec.cout].view(rows, spec.cout)
                        dot_used += rows * spec.cout
                    if want_bn or want_bias:      # channel sums ride along with the wgrad kernel
                        sums = sums_dest(b_idx, want_bias and (train_bn or op.bn is None), spec.cout, dz.device)
                    dws = ops.conv_wgrad(spec, dz, xin, [c.weight.detach() for c in op.convs], scale=scale, dot=dot,
                                         table=self.table(op, H, W, False, xin.device, wgrad=True), sum_dz=sums,
                                         outs=[dest(j) for j in op.pidx[:nw]])
                    for j, dw in zip(op.pidx[:nw], dws):
                        if need[j]:
                            grads[j] = dw
                elif want_bias:
                    sums = ops.channel_sums(dz, out=sums_dest(b_idx, train_bn or op.bn is None, spec.cout, dz.device))
                if train_bn:
                    if want_bias:
                        grads[b_idx[0]] = sums
                elif op.bn is not None:
                    cb = op.convs[0].bias.detach() if op.has_bias else None
                    dg, db, dcb = ops.bn_param_grads(dot, sums, op.bn.running_mean, invstd, scale, cb,
                                                     want_gamma=want_bn, want_beta=want_bn, want_bias=want_bias,
                                                     outs=(dest(bn_idx[0]) if want_bn else None, dest(bn_idx[1]) if want_bn else None,
                                                           dest(b_idx[0]) if want_bias else None)) \
                        if (want_bn or want_bias) else (None, None, None)
                    if want_bn:
                        grads[bn_idx[0]], grads[bn_idx[1]] = dg, db
                    if want_bias:
                        grads[b_idx[0]] = dcb
                elif want_bias:
                    wanted = [j for j in b_idx if need[j]]
                    for n_, j in enumerate(wanted):       # every branch bias sees the same gradient
                        grads[j] = sums if n_ == 0 else _copy_into(dest(j), sums)
                if op.src != 0:
                    pending[op.src] -= 1
                    last = pending[op.src] == 0
                    mask = (relu_pattern(op.src, spec.cin, spec.cout) if spec.stride == 1 else acts[op.src]) \
                        if (last and relu_producer(op.src)) else None
                    OH, OW = dz.shape[2:]
                    g[op.src] = ops.conv_dgrad(spec, dz, None, (H, W), scale=scale, res=g.get(op.src), mask=mask,
                                               table=self.table(op, OH, OW, True, dz.device),
                                               packed=self.packed(op, True, scale))
                if op.res is not None:
                    join_identity(op.res, dy_out)
                if sink is not None:
                    sink.done([j for j in op.pidx if need[j]])
            elif op.kind == "pool":
                assert self.consumers[op.src] == 1
                pending[op.src] -= 1
                g[op.src] = ops.maxpool_bwd(dz, acts[op.dst], aux[i], xin.shape[2:], op.k, op.s, op.p,
                                            relu_mask=relu_producer(op.src))
            elif op.kind == "up2add":
                join_identity(op.skip, dz)
                join_identity(op.src, ops.upsample_bwd(dz, xin.shape[2:]))
            elif op.kind == "drop":
                m = aux.get(i)
                join_identity(op.src, dz if m is None else ops.scale_planes(dz, m))
            # the activation of this op's output is no longer needed
            acts.pop(op.dst, None)
        return grads


class _PlanFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, sink, x, *params):
        keep = any(p.requires_grad for p in params)
        out, saved = engine.forward(x, keep)
        ctx.engine, ctx.sink = engine, sink
        ctx.saved = saved if keep else None
        return out

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.saved is None:
            raise RuntimeError("dasac_hip engine: the activations of this forward were already consumed by a backward pass "
                               "(retain_graph / a second backward through the same forward is not supported)")
        need = list(ctx.needs_input_grad[3:])
        sink = ctx.sink
        if sink is not None:
            grad_out = sink.begin(ctx.engine, need, grad_out)
        grads = ctx.engine.backward(ctx.saved, grad_out, need, sink=sink)
        if sink is not None:
            sink.finish()         # the launch stream waits for the outstanding bucket reductions: .grad is final for any consumer
        ctx.saved = None
        return (None, None, None) + tuple(grads)


def run_plan(engine, x, sink=None):
    """logits = plan(x); differentiable w.r.t. every parameter of the plan.  `sink`: see Engine.backward."""
    if torch.is_grad_enabled() and any(p.requires_grad for p in engine.params):
        return _PlanFunction.apply(engine, sink, x, *engine.params)
    out, _ = engine.forward(x, keep=False)
    return out


# --------------------------------------------------------------------------------------------------
# head functions with gradients
# --------------------------------------------------------------------------------------------------
class _Upsample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, size):
        ctx.low = tuple(logits.shape[2:])
        up, _, _ = ops.upsample_softmax(logits, size)
        return up

    @staticmethod
    def backward(ctx, g):
        return ops.upsample_bwd(g, ctx.low), None


def upsample_bilinear(logits, size):
    """F.interpolate(logits, size, mode='bilinear', align_corners=True) (deeplabv2.py:217).  The result remembers the
    low-resolution tensor it came from, so that a loss on it can send its gradient straight there (`_CELossLow`)."""
    up = _Upsample.apply(logits, tuple(int(s) for s in size))
    up._dasac_low = (logits, up._version)      # valid only while `up` still holds U(logits): see _ce
    return up


class _SplitBatch(torch.autograd.Function):
    """x [B, ...] -> (x[:n], x[n:]) as two contiguous tensors sharing x's storage; backward writes the two gradients side by
    side into one buffer (a missing one is zero)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n, ctx.shape = int(n), tuple(x.shape)
        return x.narrow(0, 0, ctx.n), x.narrow(0, ctx.n, x.shape[0] - ctx.n)

    @staticmethod
    def backward(ctx, ga, gb):
        ref = ga if ga is not None else gb
        g = torch.empty(ctx.shape, dtype=ref.dtype, device=ref.device)
        for part, grad in ((g.narrow(0, 0, ctx.n), ga), (g.narrow(0, ctx.n, ctx.shape[0] - ctx.n), gb)):
            if grad is None:
                part.zero_()
            else:
                part.copy_(grad)
        return g, None


def split_batch(x, n):
    return _SplitBatch.apply(x, n)


class _CELoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits_up, labels, class_weight, conf):
        loss, _, _ = ops.ce_loss(logits_up, labels, class_weight, conf)
        ctx.save_for_backward(logits_up, labels, class_weight, conf)
        return loss

    @staticmethod
    def backward(ctx, g):
        logits_up, labels, class_weight, conf = ctx.saved_tensors
        _, dl, _ = ops.ce_loss(logits_up, labels, class_weight, conf, want_grad=True, gscale=g.contiguous())
        return dl, None, None, None


class _CELossLow(torch.autograd.Function):
    """The same loss value as _CELoss, differentiated w.r.t. the LOW-resolution logits that `logits_up` was upsampled from:
    backward is one pass over logits_up (dasac_ce_loss_bwd_low) -- the full-resolution gradient (359.5 MB at 8 x 769^2)
    is never written or read back.  `logits_up` enters detached; its own autograd edge (for other consumers) is untouched."""

    @staticmethod
    def forward(ctx, logits_low, logits_up, labels, class_weight, conf):
        loss, _, _ = ops.ce_loss(logits_up, labels, class_weight, conf)
        ctx.save_for_backward(logits_up, labels, class_weight, conf)
        ctx.low_hw = tuple(logits_low.shape[2:])
        return loss

    @staticmethod
    def backward(ctx, g):
        logits_up, labels, class_weight, conf = ctx.saved_tensors
        return ops.ce_loss_bwd_low(logits_up, labels, ctx.low_hw, class_weight, conf, gscale=g.contiguous()), None, None, None, None


def _ce(logits_up, labels, class_weight, conf):
    low, version = getattr(logits_up, "_dasac_low", None) or (None, None)
    # The shortcut sends the loss gradient straight to the low-resolution logits, past `logits_up`'s own autograd edge.
    # It is taken only while that is indistinguishable from the long way round: `logits_up` was not edited in place since
    # it was upsampled (version counter) and nobody observes its gradient (retain_grad / tensor hooks).  Otherwise the
    # loss is differentiated w.r.t. logits_up itself and autograd continues through the upsampling's own backward.
    watched = logits_up.retains_grad or bool(getattr(logits_up, "_backward_hooks", None))
    if low is not None and version == logits_up._version and not watched and low.requires_grad and torch.is_grad_enabled() \
            and tuple(low.shape[:2]) == tuple(logits_up.shape[:2]):
        return _CELossLow.apply(low, logits_up.detach(), labels, class_weight, conf)
    return _CELoss.apply(logits_up, labels, class_weight, conf)


def ce_mean_all_pixels(logits_up, labels):
    """criterion(logits_up, y).mean().view(1) with CrossEntropyLoss(ignore_index=255, reduction='none')
    (deeplabv2.py:223-224): the mean runs over ALL pixels, ignored ones included."""
    return _ce(logits_up, labels, None, None)


def focal_ce(logits_up, labels, class_weight, conf=None):
    """sac.py:119-149 loss value ([1]); conf given -> `_focal_ce_conf` broadcast form."""
    return _ce(logits_up, labels, class_weight, conf)
