"""Self-supervised augmentation consistency head -- drop-in for the reference's `models/sac.py`
(/root/reference/models/sac.py): `SAC_Baseline` and `SAC` with the same constructor, `forward`
signature, buffers (`running_conf`, `slow_init`), output dictionaries and state-dict layout.

What differs is underneath: the teacher pass (upsample -> softmax -> class prior -> pad mask ->
warp to the reference frame -> multi-view pooling -> warp back -> per-class thresholds -> labels) and
the focal cross-entropy are a handful of fused gfx950 kernels (dasac_hip.ops) instead of ~40 eager
ATen calls; there is no CPU path.
"""
import torch
import torch.distributed as dist

from dasac_hip import ops
from dasac_hip import engine as E
from .basenet import BaseNet

_EMA_KEYS = ("weight", "bias", "running_mean", "running_var")


class SAC_Baseline(BaseNet):
    """Source-only / AdaBN wrapper (sac.py:15-38)."""

    def __init__(self, cfg, backbone, rank, **kwargs):
        super().__init__()
        self.backbone = backbone
        self.world_size = dist.get_world_size() if dist.is_initialized() else 1
        if dist.is_initialized():
            print("World size: ", self.world_size)
        self.rank = rank
        if "criterion" in kwargs:
            self.criterion = kwargs["criterion"]

    def forward(self, x=None, y=None, x2=None, use_teacher=False, update_teacher=False):
        return self.backbone(x, y)

    def parameter_groups(self, base_lr, wd):
        return self.backbone.parameter_groups(base_lr, wd)


class SAC(SAC_Baseline):

    def __init__(self, cfg, backbone, slow_copy, rank, **kwargs):
        super().__init__(cfg, backbone, rank, **kwargs)
        self.cfg = cfg
        self.pool_func = self._get_op(cfg.CONF_POOL)
        self.loss_func = self._get_op(cfg.LOSS)
        self.register_buffer("running_conf", torch.zeros(kwargs["num_classes"]))      # moving class prior chi
        self.slow_net = slow_copy                                                      # momentum teacher
        self.slow_net.eval()
        for p in self.slow_net.parameters():
            p.requires_grad = False
        self.register_buffer("slow_init", torch.Tensor([False]))
        self._ema_plan = None
        self._class_vectors = None
        # False (default): per-class thresholds / focal weights are evaluated by the host's ATen kernels -- the reference's own
        # arithmetic, label maps bit-equal on equal probabilities, one 19-float round trip per target step.  True: on the device in
        # the class-prior kernel (within 1 ULP, no host synchronisation: see ops.DeviceClassVectors)
        self.device_thresholds = False
        self._exempt_frozen_bn_buffers_from_ddp_broadcast()

    def _exempt_frozen_bn_buffers_from_ddp_broadcast(self):
        """DistributedDataParallel(broadcast_buffers=True) (train.py:104) re-sends EVERY buffer from rank 0 before each
        forward.  For `running_conf` / `slow_init` that is behaviour to keep (rank 0's class prior wins).
        The running statistics of frozen BatchNorm layers -- all of the teacher's, and the student's in SAC mode -- never
        diverge between ranks (nobody writes them; the teacher's move by the same EMA everywhere), so re-sending them
        changes no value; it only bumps their version counters, which would make the engine re-fold every BN and re-pack
        176 MB of scale-folded weights per network on every forward (measured: +17 ms per cfg-3 step).  DDP honours this
        module attribute when it is constructed."""
        import torch.nn as nn
        frozen = set(id(m) for m in getattr(self.backbone, "bn_freeze", []))
        names = []
        for prefix, net in (("backbone", self.backbone), ("slow_net", self.slow_net)):
            for mname, m in net.named_modules():
                if isinstance(m, BaseNet._batchnorm) and (prefix == "slow_net" or id(m) in frozen):
                    for bname, _ in m.named_buffers(recurse=False):
                        names.append(".".join(x for x in (prefix, mname, bname) if x))
        self._ddp_params_and_buffers_to_ignore = names

    @torch.no_grad()
    def broadcast_frozen_buffers(self, src=0, group=None):
        """The buffers exempted above are left out of DistributedDataParallel's construction-time synchronisation as well.
        Ranks that built or loaded different frozen-BN statistics (from-scratch init, a resume where only rank 0 reads the
        snapshot) would stay divergent without any error -- call this once after load / resume, before wrapping in DDP
        (`dasac_hip.parallel.OverlappedDataParallel` does it itself).  No-op without a process group."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return
        names = set(self._ddp_params_and_buffers_to_ignore)
        bufs = [b for n, b in self.named_buffers() if n in names and b.is_floating_point()]
        flat = torch.cat([b.reshape(-1) for b in bufs])
        dist.broadcast(flat, src, group=group)
        if dist.get_rank(group) != src:
            o = 0
            for b in bufs:
                b.copy_(flat[o:o + b.numel()].view(b.shape))
                o += b.numel()

    def _get_op(self, name):
        op_name = "_{}".format(name)
        assert hasattr(self, op_name), "Pooling OP {} not found".format(op_name)
        return getattr(self, op_name)

    # ------------------------------------------------------------------ momentum teacher (sac.py:70-102)
    def _ema_pairs(self):
        fast, slow = self.backbone.state_dict(), self.slow_net.state_dict()
        keys = [k for k in fast if k.split(".")[-1] in _EMA_KEYS]
        return [fast[k] for k in keys], [slow[k] for k in keys]

    @torch.no_grad()
    def _momentum_update(self, update=False):
        """First call: teacher <- student, chi <- beta, returns [0.].  Afterwards: sum over tensors of
        ||teacher - student||_2 (always, sac.py:374) and, if `update`, the EMA step -- one launch."""
        if not self.slow_init[0]:
            print(">>> Re-initialising ")
            self.running_conf.fill_(self.cfg.THRESHOLD_BETA)
            self.slow_init[0] = True
            self.slow_net.load_state_dict(self.backbone.state_dict())
            return torch.Tensor([0.]).type_as(self.running_conf)
        fast, slow = self._ema_pairs()
        key = tuple(t.data_ptr() for t in fast + slow)
        if self._ema_plan is None or self._ema_plan[0] != key:
            self._ema_plan = (key, ops.EmaPlan(fast, slow))
        return self._ema_plan[1].run(self.cfg.NET_MOMENTUM, update).view(1)

    # ------------------------------------------------------------------ class prior (sac.py:104-117,151-152)
    @torch.no_grad()
    def _update_running_conf(self, probs, tolerance=1e-8):
        B, C, H, W = probs.size()
        sums = ops.class_sums(probs)
        ops.class_state(self.running_conf, sums, B, H * W, self.cfg.THRESHOLD_BETA, self.cfg.STAT_MOMENTUM, True,
                        self.cfg.FOCAL_P, want_disc=False, want_focal=False)

    def _threshold_discount(self):
        """1 - exp(-chi/beta), evaluated by the CPU ATen kernels the reference uses (bit-equal thresholds)."""
        disc, _ = ops.class_vectors(self.running_conf, self.cfg.THRESHOLD_BETA, self.cfg.FOCAL_P)
        return disc

    def _focal_weight(self, p):
        _, fw = ops.class_vectors(self.running_conf, self.cfg.THRESHOLD_BETA, p, want_disc=False)
        return fw

    # ------------------------------------------------------------------ losses (sac.py:119-149)
    def _focal_ce(self, logits, pseudo_gt, teacher_probs, p=3):
        fw = self._focal_weight(p)
        with torch.no_grad():
            _, _, per_class = ops.ce_loss(logits.detach(), pseudo_gt, fw, None, want_per_class=True)
        return E.focal_ce(logits, pseudo_gt, fw, None).view(()), per_class

    def _focal_ce_conf(self, logits, pseudo_gt, teacher_probs, p=3):
        """Confidence-weighted focal CE; keeps the reference's [B,B,H,W] broadcast (sac.py:148)."""
        fw = self._focal_weight(p)
        with torch.no_grad():
            _, _, per_class = ops.ce_loss(logits.detach(), pseudo_gt, fw, None, want_per_class=True)
        return E.focal_ce(logits, pseudo_gt, fw, teacher_probs).view(()), per_class

    # ------------------------------------------------------------------ pseudo labels (sac.py:154-187)
    @torch.no_grad()
    def _pseudo_labels_probs(self, probs, ignore_augm, discount=True):
        disc = self._threshold_discount() if discount else None
        return ops.pseudo_labels(probs, ignore_augm, self.cfg.RUN_CONF_UPPER, self.cfg.RUN_CONF_LOWER, disc, want_idx=True)

    # ------------------------------------------------------------------ multi-view fusion
    @torch.no_grad()
    def _gather(self, tensor, T):
        """If a group's views are spread over ranks, fetch the missing ones (sac.py:198-216)."""
        B = tensor.size(0)
        stride = max(1, T // B)
        if stride == 1:
            return tensor
        parts = [torch.empty_like(tensor) for _ in range(self.world_size)]
        dist.all_gather(parts, tensor.contiguous())
        first = stride * (self.rank * B // T)
        return torch.cat(parts[first:first + stride], 0)

    @torch.no_grad()
    def _avg_pool(self, probs, T, tolerance=0.1):
        """sac.py:238-269 on views that are already aligned (and coverage-weighted): gather the group's missing
        views, sum over the T views, normalise over classes; the result is repeated for the T0 = min(T, B) views
        this rank holds.  (`_refine` runs the same arithmetic fused with the warps.)"""
        T0 = min(T, probs.size(0))
        probs = self._gather(probs, T)
        pooled, mask, _ = ops.warp_pool(probs, None, None, T, "avg_pool", tolerance)
        N, C, H, W = pooled.shape
        return (pooled[:, None].expand(N, T0, C, H, W).flatten(0, 1).contiguous(),
                mask[:, None].expand(N, T0, 1, H, W).flatten(0, 1).contiguous())

    @torch.no_grad()
    def _minentropy_pool(self, probs, T, tolerance=0.1):
        """sac.py:218-236: every view takes the probs of its group's lowest-entropy view (written back into `probs`
        like the reference does, :234).  No cross-rank gather exists for this pooling: all T views must be local."""
        BT, C, H, W = probs.size()
        if BT % T:
            raise RuntimeError("minentropy_pool needs whole groups of T={} views on a rank, got {} (the reference's "
                               "view(-1,T,1,H,W) at sac.py:222 fails the same way)".format(T, BT))
        pooled, mask, _ = ops.warp_pool(probs, None, None, T, "minentropy_pool", tolerance)
        N = BT // T
        probs.view(N, T, C, H, W).copy_(pooled[:, None].expand(N, T, C, H, W))
        return probs, mask[:, None].expand(N, T, 1, H, W).flatten(0, 1).contiguous()

    @torch.no_grad()
    def _refine(self, frames, pred_logits, T, affine, affine_inv, ignore_mask, pool=True, debug=True):
        """sac.py:271-313 as five launches: upsample+softmax+prior sums+pad mask, class-state update,
        warp+pool, warp back (+ the diagnostic frame warp)."""
        B, _, h, w = frames.size()
        _, probs, sums = ops.upsample_softmax(pred_logits, (h, w), ignore_mask, want_up=False, want_probs=True,
                                              want_sums=self.training)
        if self.device_thresholds:
            # opt-in: threshold discount / focal weights from the SAME launch that updates chi -- no host round trip in the step
            disc, fw = ops.class_state(self.running_conf, sums, B, h * w, self.cfg.THRESHOLD_BETA, self.cfg.STAT_MOMENTUM,
                                       bool(self.training), self.cfg.FOCAL_P, want_disc=True, want_focal=True)
            self._class_vectors = ops.DeviceClassVectors(disc, fw)
        else:
            if self.training:
                ops.class_state(self.running_conf, sums, B, h * w, self.cfg.THRESHOLD_BETA, self.cfg.STAT_MOMENTUM, True,
                                self.cfg.FOCAL_P, want_disc=False, want_focal=False)
            self._class_vectors = ops.HostClassVectors(self.running_conf)      # chi is final for this step: start its D2H
        diags = {}
        if not pool:
            return probs, diags
        T_local = min(T, B)            # views of a group held by this rank (sac.py:244)
        affine, affine_inv = affine.contiguous(), affine_inv.contiguous()
        if T_local < T:
            # views sharded across ranks: every rank needs all T aligned views of its group (sac.py:246).  The warp
            # is per view, so gathering the un-warped probs and their thetas and warping here yields the same bits
            # as gathering the other ranks' warped views, and keeps warp + pool one fused pass.
            if self.cfg.CONF_POOL != "avg_pool":
                raise RuntimeError("{} has no cross-rank gather: a rank must hold whole groups of T={} views, got {} "
                                   "(sac.py:222 fails in the reference)".format(self.cfg.CONF_POOL, T, B))
            probs_g, aff_g, inv_g = self._gather(probs, T), self._gather(affine, T), self._gather(affine_inv, T)
            pooled, mask, aligned_g = ops.warp_pool(probs_g, aff_g, inv_g, T, self.cfg.CONF_POOL, want_aligned=True)
            lo = (self.rank * B) % T
            diags["teacher_aligned"] = aligned_g[lo:lo + B].contiguous()
        else:
            pooled, mask, aligned = ops.warp_pool(probs, affine, affine_inv, T, self.cfg.CONF_POOL, want_aligned=True)
            diags["teacher_aligned"] = aligned
        if debug:
            diags["frames_aligned"] = ops.warp_affine(frames, affine)
        refined = ops.warp_back(pooled, mask, affine_inv, T_local)
        return refined, diags

    # ------------------------------------------------------------------ forward (sac.py:315-378)
    def forward(self, x, y=None, x2=None, affine=None, affine_inv=None,
                use_teacher=False, update_teacher=False, reset_teacher=False, T=None, teacher=False):
        """x: student crops [B,3,H,W]; y: labels [B,H,W] (255 ignore, -1 augmentation padding);
        x2: the same crops without photometric noise; T: views per target image."""
        if y is None:                                  # inference
            return self.slow_net(x) if teacher else self.backbone(x)
        if reset_teacher:
            self.slow_init[0] = False
        ignore_mask = ops.label_pad_mask(y, -1, 255)    # (y == -1), and y <- 255 there in place like the reference (sac.py:337-338)
        losses, net_outs = self.backbone(x, y)
        if update_teacher:
            print("Updating the teacher")
            losses["teacher_diff"] = self._momentum_update(True)
        if use_teacher:
            self.slow_net.eval()
            with torch.no_grad():
                slow_logits, slow_logits_up = self.slow_net(x2)
                probs_teacher, diags = self._refine(x2, slow_logits, T, affine, affine_inv, ignore_mask, pool=self.cfg.CONF_POOL_ON)
                # thresholds / focal weights from chi by the reference's own CPU arithmetic (bit-equal label maps);
                # the 19-float round trip was started inside _refine and lands while the warps run
                disc, fw = self._class_vectors.finish(self.cfg.THRESHOLD_BETA, self.cfg.FOCAL_P, self.cfg.CONF_DISCOUNT)
                pseudo_labels, teacher_conf, _ = ops.pseudo_labels(probs_teacher, ignore_mask, self.cfg.RUN_CONF_UPPER,
                                                                   self.cfg.RUN_CONF_LOWER, disc)
            conf = teacher_conf if self.cfg.LOSS == "focal_ce_conf" else None
            losses["self_ce"] = E.focal_ce(net_outs["logits_up"], pseudo_labels, fw, conf).view(1)
            net_outs["teacher_init"] = slow_logits_up
            net_outs["teacher_refined"] = probs_teacher
            net_outs["teacher_conf"] = teacher_conf
            net_outs["teacher_labels"] = pseudo_labels
            net_outs["running_conf"] = self.running_conf
            losses["teacher_diff"] = self._momentum_update(False)
            net_outs.update(diags)
        return losses, net_outs

    def forward_fused(self, src_x, src_y, x, y, x2, affine, affine_inv, update_teacher=False, T=None):
        """One training iteration's two student passes as ONE: `forward(src_x, src_y)` and
        `forward(x, y, x2, affine, affine_inv, use_teacher=True, update_teacher=..., T=T)` (train.py:128,219-222), with
        the student evaluated once on the concatenated batch [source crops; target crops].

        Why this is the same computation: between the two passes of an iteration the reference takes no optimiser step
        (train.py:132-138,231-233), so both see the same student weights; in SAC mode every BatchNorm is frozen
        (models/__init__.py:29), so the samples of a batch do not interact; and the momentum teacher (EMA step, forward,
        refinement, pseudo labels) never reads the student's activations.  Back-propagating loss_src + LR_TARGET * self_ce
        through the one pass yields d loss_src + LR_TARGET * d self_ce -- what the two backward passes accumulate in .grad --
        up to the summation order of the weight-gradient reductions.  What it buys: every student GEMM runs once over 16
        crops instead of twice over 8 (half the launches, half the split-K slab traffic, one gradient all-reduce per
        iteration instead of two).
        Returns (source losses, target losses, net_outs) exactly as the two calls would (the target's ground-truth CE
        `loss_ce` is still evaluated and still never back-propagated).  Needs equal source / target crop
        sizes and frozen BN; not for the baseline (AdaBN) mode."""
        assert self.backbone._bn_frozen(), "forward_fused: batch-statistics BN couples the samples of a pass"
        assert tuple(src_x.shape[1:]) == tuple(x.shape[1:]), "forward_fused: source and target crops must have one size"
        ops.label_pad_mask(src_y, -1, 255)        # forward() rewrites -1 in WHATEVER labels it is given (sac.py:337-338): both passes
        ignore_mask = ops.label_pad_mask(y, -1, 255)
        losses_tgt = {}
        if update_teacher:
            print("Updating the teacher")
            losses_tgt["teacher_diff"] = self._momentum_update(True)
        self.slow_net.eval()
        with torch.no_grad():
            slow_logits, slow_logits_up = self.slow_net(x2)
            probs_teacher, diags = self._refine(x2, slow_logits, T, affine, affine_inv, ignore_mask, pool=self.cfg.CONF_POOL_ON)
            disc, fw = self._class_vectors.finish(self.cfg.THRESHOLD_BETA, self.cfg.FOCAL_P, self.cfg.CONF_DISCOUNT)
            pseudo_labels, teacher_conf, _ = ops.pseudo_labels(probs_teacher, ignore_mask, self.cfg.RUN_CONF_UPPER,
                                                               self.cfg.RUN_CONF_LOWER, disc)
        n_src = src_x.shape[0]
        logits = self.backbone._logits(torch.cat([src_x, x], 0))
        logits_src, logits_tgt = E.split_batch(logits, n_src)
        up_src = E.upsample_bilinear(logits_src, src_x.shape[-2:])
        up_tgt = E.upsample_bilinear(logits_tgt, x.shape[-2:])
        losses_src = {"loss_ce": E.ce_mean_all_pixels(up_src, src_y).view(1)}
        losses_tgt["loss_ce"] = E.ce_mean_all_pixels(up_tgt, y).view(1)
        conf = teacher_conf if self.cfg.LOSS == "focal_ce_conf" else None
        losses_tgt["self_ce"] = E.focal_ce(up_tgt, pseudo_labels, fw, conf).view(1)
        net_outs = {"logits_up": up_tgt}
        if self.backbone._returns_logits:
            net_outs["logits"] = logits_tgt
        net_outs.update(teacher_init=slow_logits_up, teacher_refined=probs_teacher, teacher_conf=teacher_conf,
                        teacher_labels=pseudo_labels, running_conf=self.running_conf)
        losses_tgt["teacher_diff"] = self._momentum_update(False)
        net_outs.update(diags)
        return losses_src, losses_tgt, net_outs

    def parameter_groups(self, base_lr, wd):
        return self.backbone.parameter_groups(base_lr, wd)
