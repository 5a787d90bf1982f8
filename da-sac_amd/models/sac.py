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
        sums = probs.sum((0, 2, 3), dtype=torch.float64)
        ops.class_state(self.running_conf, sums, B, H * W, self.cfg.THRESHOLD_BETA, self.cfg.STAT_MOMENTUM, True,
                        self.cfg.FOCAL_P, want_disc=False, want_focal=False)

    def _threshold_discount(self):
        disc, _ = ops.class_state(self.running_conf, None, 1, 1, self.cfg.THRESHOLD_BETA, self.cfg.STAT_MOMENTUM, False,
                                  self.cfg.FOCAL_P, want_focal=False)
        return disc

    def _focal_weight(self, p):
        _, fw = ops.class_state(self.running_conf, None, 1, 1, self.cfg.THRESHOLD_BETA, self.cfg.STAT_MOMENTUM, False, p,
                                want_disc=False)
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

    def _avg_pool(self, probs, T, tolerance=0.1):
        raise RuntimeError("pooling runs fused inside _refine (dasac_warp_pool); not callable on its own")

    _minentropy_pool = _avg_pool

    @torch.no_grad()
    def _refine(self, frames, pred_logits, T, affine, affine_inv, ignore_mask, pool=True, debug=True):
        """sac.py:271-313 as five launches: upsample+softmax+prior sums+pad mask, class-state update,
        warp+pool, warp back (+ the diagnostic frame warp)."""
        B, _, h, w = frames.size()
        _, probs, sums = ops.upsample_softmax(pred_logits, (h, w), ignore_mask, want_up=False, want_probs=True,
                                              want_sums=self.training)
        if self.training:
            ops.class_state(self.running_conf, sums, B, h * w, self.cfg.THRESHOLD_BETA, self.cfg.STAT_MOMENTUM, True,
                            self.cfg.FOCAL_P, want_disc=False, want_focal=False)
        diags = {}
        if not pool:
            return probs, diags
        T_local = min(T, B)            # views of a group held by this rank (sac.py:244)
        affine, affine_inv = affine.contiguous(), affine_inv.contiguous()
        if T_local < T:
            # views sharded across ranks: every rank needs all T warped views of its group.  The warp is
            # linear, so gather the un-warped probs and thetas of the group instead (sac.py:246).
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
        ignore_mask = (y == -1)
        y[ignore_mask] = 255                           # in place, like the reference (sac.py:338)
        losses, net_outs = self.backbone(x, y)
        if update_teacher:
            print("Updating the teacher")
            losses["teacher_diff"] = self._momentum_update(True)
        if use_teacher:
            self.slow_net.eval()
            with torch.no_grad():
                slow_logits, slow_logits_up = self.slow_net(x2)
                probs_teacher, diags = self._refine(x2, slow_logits, T, affine, affine_inv, ignore_mask, pool=self.cfg.CONF_POOL_ON)
                disc = self._threshold_discount() if self.cfg.CONF_DISCOUNT else None
                pseudo_labels, teacher_conf, _ = ops.pseudo_labels(probs_teacher, ignore_mask, self.cfg.RUN_CONF_UPPER,
                                                                   self.cfg.RUN_CONF_LOWER, disc)
            fw = self._focal_weight(self.cfg.FOCAL_P)
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

    def parameter_groups(self, base_lr, wd):
        return self.backbone.parameter_groups(base_lr, wd)
