"""Oracle (CPU, test infrastructure) for the SAC module state machine and the training-step
order.  Follows /root/reference/models/sac.py:315-378 (forward), :70-102 (teacher), and
/root/reference/train.py:119-155 + :211-250 + :266-298 (source step, target step, loop),
/root/reference/base_trainer.py:63-66 (SGD momentum), models/basenet.py:102-139 (groups).
"""
import copy

import torch

from . import head_ref as H
from . import nets_ref as N

DEFAULT_CFG = dict(                      # core/config.py:130-159 + deeplabv2_resnet101_train.yaml
    ARCH="deeplabv2_resnet101", BASELINE=False, LR=2.5e-4, LR_TARGET=5.0, WEIGHT_DECAY=5e-4,
    MOMENTUM=0.9, STAT_MOMENTUM=0.99, NET_MOMENTUM=0.99, NET_MOMENTUM_ITER=100,
    CONF_DISCOUNT=True, CONF_POOL_ON=True, CONF_POOL="avg_pool", FOCAL_P=3, LOSS="focal_ce_conf",
    RUN_CONF_UPPER=0.75, RUN_CONF_LOWER=0.2, THRESHOLD_BETA=1e-3,
)


class SacOracle:
    """Student / momentum-teacher pair over flat state dicts (reference checkpoint keys)."""

    def __init__(self, student_sd, cfg=None, num_classes=19, net_kwargs=None, gather=None):
        self.gather = gather                                  # cross-rank view gather (sac.py:198-216) or None
        self.cfg = dict(DEFAULT_CFG)
        self.cfg.update(cfg or {})
        self.arch = self.cfg["ARCH"].lower()
        self.net_kwargs = net_kwargs or {}
        self.student = {k: v.clone() for k, v in student_sd.items()}
        for k in N.trainable_keys(self.student):
            self.student[k].requires_grad_(True)
        self.teacher = {k: v.detach().clone() for k, v in student_sd.items()}
        self.running_conf = torch.zeros(num_classes)          # sac.py:53-54
        self.slow_init = torch.zeros(1)                       # sac.py:63
        self.training = True
        # SAC mode freezes every BN (models/__init__.py:29); baseline trains them
        self.bn_train = bool(self.cfg["BASELINE"])

    # ---- sac.py:70-102
    def momentum_update(self, update):
        if not bool(self.slow_init[0]):
            self.running_conf.fill_(self.cfg["THRESHOLD_BETA"])
            self.slow_init[0] = 1.0
            for k, v in self.student.items():
                self.teacher[k].copy_(v.detach())
            return torch.zeros(1)
        with torch.no_grad():
            fast = {k: v.detach() for k, v in self.student.items()}
            return H.momentum_update(self.teacher, fast, self.cfg["NET_MOMENTUM"], update)

    def _net(self, sd, im, y=None, bn_train=False):
        return N.segnet_forward(self.arch, sd, im, y, bn_train=bn_train, **self.net_kwargs)

    # ---- sac.py:315-378
    def forward(self, x, y=None, x2=None, affine=None, affine_inv=None, use_teacher=False,
                update_teacher=False, reset_teacher=False, T=None, teacher=False):
        c = self.cfg
        if y is None:
            return self._net(self.teacher if teacher else self.student, x)
        if c["BASELINE"]:
            return self._net(self.student, x, y, bn_train=self.training and self.bn_train)
        if reset_teacher:
            self.slow_init[0] = 0.0
        ignore_mask = (y == -1)
        y[ignore_mask] = 255                                   # in place, like :338
        losses, outs = self._net(self.student, x, y, bn_train=False)
        if update_teacher:
            losses["teacher_diff"] = self.momentum_update(True)
        if use_teacher:
            with torch.no_grad():
                slow_logits, slow_up = self._net(self.teacher, x2)
                refined, chi, diags = H.refine(
                    x2, slow_logits, T, affine, affine_inv, ignore_mask, self.running_conf,
                    beta=c["THRESHOLD_BETA"], stat_momentum=c["STAT_MOMENTUM"], training=self.training,
                    pool=c["CONF_POOL_ON"], pool_kind=c["CONF_POOL"], gather=self.gather)
                self.running_conf.copy_(chi)
                disc = H.threshold_discount(self.running_conf, c["THRESHOLD_BETA"]) if c["CONF_DISCOUNT"] else None
                labels, conf, _ = H.pseudo_labels(refined, ignore_mask, c["RUN_CONF_UPPER"], c["RUN_CONF_LOWER"], disc)
            if c["LOSS"] == "focal_ce_conf":
                loss, _ = H.focal_ce_conf(outs["logits_up"], labels, conf, self.running_conf, c["FOCAL_P"])
            else:
                loss, _ = H.focal_ce(outs["logits_up"], labels, self.running_conf, c["FOCAL_P"])
            losses["self_ce"] = loss.mean().view(1)
            outs["teacher_init"] = slow_up
            outs["teacher_refined"] = refined
            outs["teacher_conf"] = conf
            outs["teacher_labels"] = labels
            outs["running_conf"] = self.running_conf
            losses["teacher_diff"] = self.momentum_update(False)
            outs.update(diags)
        return losses, outs

    # ---- models/basenet.py:102-139 with the multipliers of deeplabv2.py:203-211
    def param_groups(self):
        lr, wd = self.cfg["LR"], self.cfg["WEIGHT_DECAY"]
        new_prefixes = {
            "deeplabv2_resnet101": ("model.layer5.",),
            "deeplabv2_vgg16_bn": ("classifier.", "features.42.", "features.44."),
            "fcn_vgg16_bn": ("vgg_head.", "score_pool4.", "score_pool3."),
        }[self.arch]
        groups = [dict(keys=[], lr=lr, wd=wd), dict(keys=[], lr=2 * lr, wd=0.0),
                  dict(keys=[], lr=10 * lr, wd=wd), dict(keys=[], lr=20 * lr, wd=0.0)]
        for k in N.trainable_keys(self.student):
            is_new = k.startswith(new_prefixes)
            is_bias = k.endswith(".bias")
            groups[2 * int(is_new) + int(is_bias)]["keys"].append(k)
        return groups


class SgdOracle:
    """torch.optim.SGD(momentum=.9, nesterov=False) restated (base_trainer.py:63-66):
    g += wd*p ; buf = g (first step) or m*buf + g ; p -= lr*buf."""

    def __init__(self, model, momentum=0.9):
        self.model, self.momentum, self.bufs = model, momentum, {}
        self.groups = model.param_groups()

    def zero_grad(self):
        for p in self.model.student.values():
            p.grad = None

    @torch.no_grad()
    def step(self):
        for g in self.groups:
            for k in g["keys"]:
                p = self.model.student[k]
                if p.grad is None:
                    continue
                d = p.grad + g["wd"] * p if g["wd"] != 0 else p.grad.clone()
                if k not in self.bufs:
                    self.bufs[k] = d.clone()
                else:
                    self.bufs[k].mul_(self.momentum).add_(d)
                p.add_(self.bufs[k], alpha=-g["lr"])


def sac_train_iteration(model, optim, src_batch, tgt_batch, T, update_teacher):
    """One iteration of train.py:266-298 in SAC mode on one rank:
    source fwd -> zero_grad -> source bwd (no optimiser step, :132-138) ->
    target fwd (teacher update first when asked) -> (LR_TARGET*self_ce) bwd -> SGD step (:231-233)."""
    xs, ys = src_batch
    losses_src, _ = model.forward(xs, ys)
    optim.zero_grad()
    losses_src["loss_ce"].mean().backward()
    f1, gt, f2, aff, aff_inv = tgt_batch
    losses_tgt, outs = model.forward(f1, gt, f2, aff, aff_inv, use_teacher=True,
                                     update_teacher=update_teacher, T=T)
    (model.cfg["LR_TARGET"] * losses_tgt["self_ce"].mean()).backward()
    optim.step()
    return ({k: float(v.detach().mean()) for k, v in losses_src.items()},
            {k: float(v.detach().mean()) for k, v in losses_tgt.items()}, outs)


def baseline_train_iteration(model, optim, src_batch, tgt_images):
    """train.py:274-289 in baseline (AdaBN) mode: source fwd/bwd/step, then a no-grad
    train-mode forward of the target crops whose only effect is the BN running stats."""
    xs, ys = src_batch
    losses, _ = model.forward(xs, ys)
    optim.zero_grad()
    losses["loss_ce"].mean().backward()
    optim.step()
    with torch.no_grad():
        dummy = torch.zeros(tgt_images.shape[0], tgt_images.shape[2], tgt_images.shape[3], dtype=torch.int64)
        model.forward(tgt_images, dummy)
    return {k: float(v.detach().mean()) for k, v in losses.items()}


def view_slice_index(world, rank, N_groups, L):
    """train.py:186-209 index math: which gathered tensor and which view range a rank keeps.
    Returns None when whole groups fit on a rank, else (index0, index1, index1_end)."""
    assert (N_groups * L) % world == 0, "Batch size does not fit world size"
    per = N_groups * L // world
    if per >= L:
        return None
    flat = rank * per
    return flat // L, flat % L, flat % L + per


def gather_index(world, rank, B, T):
    """sac.py:203-214: list slice of the all_gather'ed teacher probs a rank concatenates."""
    stride = max(1, T // B)
    if stride <= 1:
        return None
    lo = stride * (rank * B // T)
    return lo, lo + stride


# --------------------------------------------------------------------------------------------------
# several ranks in one process: what DistributedDataParallel + the two hand-rolled all_gathers do
# (train.py:104,157-209; models/sac.py:198-216), with threads standing in for ranks
# --------------------------------------------------------------------------------------------------
class ThreadWorld:
    """`world` oracle ranks as threads; every collective is a barrier around a shared slot list."""

    def __init__(self, world):
        import threading
        self.world = world
        self._slots = [None] * world
        self._bar = threading.Barrier(world)

    def all_gather(self, rank, item):
        self._slots[rank] = item
        self._bar.wait()
        out = list(self._slots)
        self._bar.wait()
        return out

    def gather_views(self, rank):
        """models/sac.py:198-216 for this rank."""
        def fn(tensor, T):
            idx = gather_index(self.world, rank, tensor.size(0), T)
            if idx is None:
                return tensor
            parts = self.all_gather(rank, tensor)
            return torch.cat(parts[idx[0]:idx[1]], 0)
        return fn

    def prep_batch(self, rank, loaded, N_groups, L):
        """train.py:157-209: `loaded` [B,L,...] is what this rank's loader delivered."""
        idx = view_slice_index(self.world, rank, N_groups, L)
        if idx is None:
            return loaded.flatten(0, 1)
        assert loaded.size(1) == L
        parts = self.all_gather(rank, loaded)
        return parts[idx[0]].flatten(0, 1)[idx[1]:idx[2]]

    def average_grads(self, rank, model):
        """DDP's bucketed all-reduce(SUM)/world after a backward pass (train.py:104)."""
        mine = {k: v.grad for k, v in model.student.items() if v.requires_grad and v.grad is not None}
        parts = self.all_gather(rank, mine)
        for k in mine:
            model.student[k].grad = sum(p[k] for p in parts) / self.world

    def broadcast_buffers(self, rank, model):
        """DDP(broadcast_buffers=True): before every forward all buffers take rank 0's values (quirk 5)."""
        src = self.all_gather(rank, model)[0]
        if rank != 0:
            model.running_conf.copy_(src.running_conf)
            model.slow_init.copy_(src.slow_init)
            for sd_mine, sd_src in ((model.student, src.student), (model.teacher, src.teacher)):
                for k, v in sd_src.items():
                    if k.split(".")[-1] in ("running_mean", "running_var", "num_batches_tracked"):
                        sd_mine[k].copy_(v.detach())
        self._bar.wait()

    def run(self, fn):
        """Runs fn(rank) on every rank; returns the results in rank order (first exception re-raised)."""
        import threading
        out, err = [None] * self.world, []

        def body(r):
            try:
                out[r] = fn(r)
            except BaseException as e:       # noqa: a failing rank must not leave the others at a barrier
                err.append(e)
                self._bar.abort()
        threads = [threading.Thread(target=body, args=(r,)) for r in range(self.world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        real = [e for e in err if not isinstance(e, threading.BrokenBarrierError)]
        if err:
            raise (real or err)[0]
        return out


def sharded_sac_iteration(tw, rank, model, optim, src_batch, loaded_tgt, N_groups, L, update_teacher):
    """train.py:266-298 on rank `rank` of a ThreadWorld: DDP buffer broadcast before each forward, gradient
    averaging after each backward, `_prep_batch` on the five loaded target tensors."""
    xs, ys = src_batch
    tw.broadcast_buffers(rank, model)
    losses_src, _ = model.forward(xs, ys)
    optim.zero_grad()
    losses_src["loss_ce"].mean().backward()
    tw.average_grads(rank, model)
    f1, gt, f2, aff, aff_inv = (tw.prep_batch(rank, t, N_groups, L) for t in loaded_tgt)
    tw.broadcast_buffers(rank, model)
    losses_tgt, outs = model.forward(f1, gt.clone(), f2, aff, aff_inv, use_teacher=True, update_teacher=update_teacher, T=L)
    (model.cfg["LR_TARGET"] * losses_tgt["self_ce"].mean()).backward()
    tw.average_grads(rank, model)
    optim.step()
    return ({k: float(v.detach().mean()) for k, v in losses_src.items()},
            {k: float(v.detach().mean()) for k, v in losses_tgt.items()}, outs)
