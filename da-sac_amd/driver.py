"""Minimal training driver reproducing the reference's step order on one rank
(/root/reference/train.py:119-155 source step, :211-250 target step, :266-298 loop body,
/root/reference/base_trainer.py:47-73 optimiser factory).  Works on the bare module or on a
DistributedDataParallel wrapper (backend "nccl" == RCCL on ROCm)."""
import torch


def make_optimizer(net, cfg_model, fused=True):
    """`BaseTrainer.get_optim` (base_trainer.py:47-73) over the model's four parameter groups (train.py:93-96):
      OPT == "SGD" (the default, configs/*.yaml): SGD(momentum, nesterov=OPT_NESTEROV) -- the fused multi-tensor HIP optimiser
          (same update rule / state layout; plain momentum only) or, with fused=False / nesterov, torch.optim.SGD itself;
      OPT == "Adam": torch.optim.Adam(lr, betas=(BETA1, 0.999), weight_decay) (base_trainer.py:57-61);
      any other name in torch.optim: optim(params, lr=LR) (base_trainer.py:68-69); unknown names raise NotImplementedError.
    Every group carries its own lr / weight_decay (basenet.py:102-139), so the keyword defaults below only fill the gaps,
    exactly as in the reference."""
    core = net.module if hasattr(net, "module") else net
    groups = core.parameter_groups(cfg_model.LR, cfg_model.WEIGHT_DECAY)
    opt = getattr(cfg_model, "OPT", "SGD")
    if not hasattr(torch.optim, opt):
        print("Optimiser {} not supported".format(opt))
        raise NotImplementedError
    if opt == "Adam":
        upd = torch.optim.Adam(groups, lr=cfg_model.LR, betas=(getattr(cfg_model, "BETA1", 0.5), 0.999), weight_decay=cfg_model.WEIGHT_DECAY)
    elif opt == "SGD":
        nesterov = getattr(cfg_model, "OPT_NESTEROV", False)
        if fused and not nesterov:
            from dasac_hip.optim import FusedSGD
            upd = FusedSGD(groups, lr=cfg_model.LR, momentum=cfg_model.MOMENTUM, weight_decay=cfg_model.WEIGHT_DECAY)
        else:
            upd = torch.optim.SGD(groups, lr=cfg_model.LR, momentum=cfg_model.MOMENTUM, nesterov=nesterov, weight_decay=cfg_model.WEIGHT_DECAY)
    else:
        upd = getattr(torch.optim, opt)(groups, lr=cfg_model.LR)
    upd.zero_grad()
    return upd


def train_epoch(net, optim, loader_source, loader_target, cfg_model, group_size, target_only=False, on_iteration=None):
    """The loop body of `Trainer.train_epoch` (train.py:266-298) -- the per-iteration POLICY around the two step functions:
    baseline mode runs the source step and the no-grad target forward (AdaBN, :281-289); SAC mode refreshes the momentum
    teacher on every NET_MOMENTUM_ITER-th iteration of the epoch (`update_teacher = i % NET_MOMENTUM_ITER == 0`, :294), skips
    the source pass under TRAIN.TARGET_ONLY (:274-276).  Batches are what the reference's loaders yield, already on the device:
    (image, masks_gt) and (frames1, frames_gt, frames2, affine, affine_inv).  Returns the number of iterations."""
    n = 0
    for i, (batch_source, batch_target) in enumerate(zip(loader_source, loader_target)):
        if cfg_model.BASELINE:
            out = (baseline_train_iteration(net, optim, batch_source, batch_target[0]), None, None)
        else:
            update_teacher = i % cfg_model.NET_MOMENTUM_ITER == 0
            out = sac_train_iteration(net, optim, batch_source, batch_target, group_size, update_teacher, cfg_model.LR_TARGET,
                                      target_only=target_only)
        if on_iteration is not None:
            on_iteration(i, out)
        n = i + 1
    return n


def _dist_state(rank, world):
    import torch.distributed as dist
    on = dist.is_available() and dist.is_initialized()
    if world is None:
        world = dist.get_world_size() if on else 1
    if rank is None:
        rank = dist.get_rank() if on else 0
    return rank, world


def prep_batch(tensor, num_groups, group_size, rank=None, world=None, device=None, exchange="all_gather"):
    """Rank-local slice of one loaded target tensor [B, L, ...] (train.py:157-209).

    Every rank's loader delivers whole groups of L views.  With N*L/world >= L the groups stay where they were
    loaded: the result is just [B*L, ...].  Otherwise a group is spread over L/per consecutive ranks
    (per = N*L/world views each): rank r keeps views [f % L, f % L + per) of the first group loaded by rank
    f // L, f = r*per (the groups loaded by the other ranks are dropped, as in the reference).
    exchange="all_gather" moves the tensors exactly like train.py:194-195; "p2p" sends each rank only the `per`
    views it keeps (same result, 1/world of the bytes -- xGMI links are point-to-point anyway)."""
    import torch.distributed as dist
    rank, world = _dist_state(rank, world)
    assert (num_groups * group_size) % world == 0, "Batch size does not fit world size"
    per = num_groups * group_size // world
    if per >= group_size:
        if device is not None:
            tensor = tensor.to(device, non_blocking=True)
        return tensor.flatten(0, 1)
    assert tensor.size(1) == group_size, "Loaded sequence is incorrect {} vs. {}".format(tensor.size(1), group_size)
    # WHERE the exchange runs.  RCCL ("nccl") moves device tensors, stream-ordered: upload first, exchange on the device.  gloo
    # is a host transport: its collectives stage device tensors through pinned memory, but its send / recv hand the tensor's
    # data pointer to the TCP transport as it is -- with a device tensor the host reads (writes) VRAM through the PCIe BAR with
    # no ordering against the stream that fills (consumes) it.  Found by the 8-rank one-device test of round 6 (ranks that are
    # sender and receiver at once got torn slices).  With gloo the loader's HOST tensor is exchanged and only the `per` views
    # this rank keeps cross PCIe afterwards.
    host_exchange = dist.get_backend() == "gloo"
    if device is not None and not host_exchange:
        tensor = tensor.to(device, non_blocking=True)
    if host_exchange and tensor.is_cuda:
        tensor = tensor.cpu()
    first = rank * per
    owner, lo = first // group_size, first % group_size
    tensor = tensor.contiguous()
    if exchange == "all_gather":
        parts = [torch.empty_like(tensor) for _ in range(world)]
        dist.all_gather(parts, tensor)
        mine = parts[owner].flatten(0, 1)[lo:lo + per]
    else:
        assert exchange == "p2p", exchange
        flat = tensor.flatten(0, 1)
        mine = flat[lo:lo + per].clone() if owner == rank else torch.empty_like(flat[:per])
        work = []
        for dst in range(world):                    # what this rank owes the ranks whose slice lives here
            if dst != rank and (dst * per) // group_size == rank:
                d_lo = (dst * per) % group_size
                work.append(dist.P2POp(dist.isend, flat[d_lo:d_lo + per].contiguous(), dst))
        if owner != rank:
            work.append(dist.P2POp(dist.irecv, mine, owner))
        if work:
            for req in dist.batch_isend_irecv(work):
                req.wait()
    if device is not None and host_exchange:
        mine = mine.to(device, non_blocking=True)
    return mine


def sac_train_iteration(net, optim, src_batch, tgt_batch, group_size, update_teacher, lr_target, target_only=False,
                        sum_grads_in_optimizer=True, fuse_passes=False):
    """source fwd -> zero_grad -> source bwd (gradients kept) -> target fwd (teacher EMA first when asked)
    -> (LR_TARGET * self_ce) bwd -> one optimiser step.  Returns (source losses, target losses, net_outs)
    with the losses still on the device (no host sync here).  TRAIN.TARGET_ONLY skips the source pass altogether
    (train.py:274-276) and clears the gradients before the target backward (train.py:226-227).

    The reference lets autograd add the target-pass gradients onto the source-pass ones in `.grad` (320 `add_` launches for
    ResNet-101).  With `FusedSGD` the source gradients are set aside instead (`stash_grads`) and the update kernel applies
    source + target -- on one rank the same sum, bit for bit.  Between the target backward and step(), and after it,
    `.grad` holds the target-pass gradient only: anything that reads gradients before the step (clipping, norm logging)
    must use `optim.full_grads()` or pass sum_grads_in_optimizer=False (or another optimiser) for the reference's `.grad`
    contents.  Under data parallelism the update is mean_r(src_r) + mean_r(tgt_r) where the reference's DDP reduces
    mean_r(mean(src) + tgt_r): equal in exact arithmetic, one rounding apart in fp32 (not bit-identical).

    fuse_passes=True: the student runs ONCE over [source crops; target crops] and ONE backward pass differentiates
    loss_ce + LR_TARGET * self_ce (`SAC.forward_fused`: same weights in both passes, frozen BN, teacher independent of the
    student -- the same gradient sum, half the launches, one gradient all-reduce per iteration).  Needs the bare module or
    `dasac_hip.parallel.OverlappedDataParallel`; silently runs the two-pass order where it does not apply (target_only, stock
    DistributedDataParallel, batch-statistics BN, crops of different sizes, or a concatenated batch whose largest activation
    would leave the kernels' 4 GiB addressing window (FCN-8s at 16 crops of 512x1024 peaks at exactly 2 GiB: fused since round 5)."""
    core = net.module if hasattr(net, "module") else net
    if fuse_passes and not target_only and hasattr(net, "forward_fused") and hasattr(core, "backbone") and core.backbone._bn_frozen() \
            and tuple(src_batch[0].shape[1:]) == tuple(tgt_batch[0].shape[1:]) \
            and core.backbone._batch_fits(src_batch[0].shape[0] + tgt_batch[0].shape[0], *src_batch[0].shape[-2:]):
        images, masks = src_batch
        frames1, frames_gt, frames2, affine, affine_inv = tgt_batch
        losses_src, losses_tgt, outs = net.forward_fused(images, masks, frames1, frames_gt, frames2, affine, affine_inv,
                                                         update_teacher=update_teacher, T=group_size)
        optim.zero_grad()
        (losses_src["loss_ce"].mean() + lr_target * losses_tgt["self_ce"].mean()).backward()
        optim.step()
        return losses_src, losses_tgt, outs
    losses_src = {}
    if not target_only:
        images, masks = src_batch
        losses_src, _ = net(images, masks)
        optim.zero_grad()
        losses_src["loss_ce"].mean().backward()
        if sum_grads_in_optimizer and hasattr(optim, "stash_grads"):
            optim.stash_grads()
    frames1, frames_gt, frames2, affine, affine_inv = tgt_batch
    losses_tgt, outs = net(frames1, frames_gt, frames2, affine, affine_inv, use_teacher=True,
                           update_teacher=update_teacher, T=group_size)
    if target_only:
        optim.zero_grad()
    (lr_target * losses_tgt["self_ce"].mean()).backward()
    optim.step()
    return losses_src, losses_tgt, outs


def reduce_losses(losses, world=None):
    """train.py:243-246: every logged loss is summed over ranks and divided by the world size (one collective for the
    whole dict instead of one per key); returns python floats (the only host sync of a step)."""
    import torch.distributed as dist
    _, world = _dist_state(None, world)
    keys = sorted(losses)
    if not keys:
        return {}
    packed = torch.cat([losses[k].detach().reshape(-1)[:1] for k in keys])
    if world > 1:
        dist.all_reduce(packed)
        packed = packed / world
    return dict(zip(keys, packed.tolist()))


def baseline_train_iteration(net, optim, src_batch, tgt_images):
    """Baseline / AdaBN mode (train.py:274-289)."""
    images, masks = src_batch
    losses, _ = net(images, masks)
    optim.zero_grad()
    losses["loss_ce"].mean().backward()
    optim.step()
    with torch.no_grad():
        dummy = torch.zeros(tgt_images.shape[0], tgt_images.shape[2], tgt_images.shape[3], dtype=torch.int64, device=tgt_images.device)
        net(tgt_images, dummy)
    return losses


# --------------------------------------------------------------------------------------------------
# synthetic workload: weights, crops, labels and the four view affines
# --------------------------------------------------------------------------------------------------
def init_synthetic_weights(net, seed=0, classifier_gain=6.0):
    """Random-init weights of the right architecture, scaled like a trained network so that the
    reference's SGD hyper-parameters are stable: He-normal convs, BN gamma~U(.5,1.5), beta/mean~N(0,.1),
    var~U(.5,1.5); the BN that closes each residual branch x0.1 and the shortcut BN x0.3 keep the
    residual stream at E[f^2]~0.05-0.3 (full-range fp32 data, nothing zero-filled); classifier weights
    x6 give stride-8 logits of std~3 (peaked, unsaturated softmax: ~1/3 of the pseudo-labels fire)."""
    import torch.nn as nn
    gen = torch.Generator().manual_seed(seed)
    core = net.backbone if hasattr(net, "backbone") else net
    with torch.no_grad():
        for name, m in core.named_modules():
            if isinstance(m, nn.Conv2d):
                fan_in = m.in_channels * m.kernel_size[0] * m.kernel_size[1]
                w = torch.empty(m.weight.shape).normal_(0, (2.0 / fan_in) ** 0.5, generator=gen)
                if "conv2d_list" in name or name.startswith(("score_pool", "vgg_head.8")):
                    w *= classifier_gain
                m.weight.copy_(w)
                if m.bias is not None:
                    m.bias.copy_(torch.empty(m.bias.shape).normal_(0, 0.01, generator=gen))
            elif isinstance(m, (nn.SyncBatchNorm, nn.BatchNorm2d)):
                gain = 0.1 if name.endswith("bn3") else (0.3 if name.endswith("downsample.1") else 1.0)
                bgain = 0.3 if name.endswith("downsample.1") else 1.0
                m.weight.copy_(torch.empty(m.weight.shape).uniform_(0.5, 1.5, generator=gen) * gain)
                m.bias.copy_(torch.empty(m.bias.shape).normal_(0, 0.1, generator=gen) * bgain)
                m.running_mean.copy_(torch.empty(m.bias.shape).normal_(0, 0.1, generator=gen))
                m.running_var.copy_(torch.empty(m.bias.shape).uniform_(0.5, 1.5, generator=gen))
    return net


def view_affines(params, crop_h, crop_w):
    """theta / theta^-1 [L,2,3] of the augmented views, params = (dy, dx, alpha_deg, scale, flip) per view
    -- the formulas of /root/reference/datasets/dataloader_target.py:220-262."""
    import math
    L = len(params)
    theta = torch.zeros(L, 2, 3)
    ar = float(crop_h) / float(crop_w)
    for i, (dy, dx, alpha, scale, flip) in enumerate(params):
        s, c = math.sin(alpha * math.pi / 180.0), math.cos(alpha * math.pi / 180.0)
        theta[i, 0, 0], theta[i, 0, 1] = flip * c, s * ar
        theta[i, 1, 0], theta[i, 1, 1] = -s / ar, c
        theta[i, 0, 2] = -(c * dx + s * dy) / float(crop_w // 2)
        theta[i, 1, 2] = -(-s * dx + c * dy) / float(crop_h // 2)
        theta[i] *= scale
    inv = theta.clone()
    inv[:, 0, 1] = theta[:, 1, 0] * ar ** 2
    inv[:, 1, 0] = theta[:, 0, 1] / ar ** 2
    inv[:, 0, 2] = -(inv[:, 0, 0] * theta[:, 0, 2] + inv[:, 0, 1] * theta[:, 1, 2])
    inv[:, 1, 2] = -(inv[:, 1, 0] * theta[:, 0, 2] + inv[:, 1, 1] * theta[:, 1, 2])
    inv /= torch.tensor([p[3] for p in params], dtype=torch.float32).view(-1, 1, 1) ** 2
    return theta, inv


# identity | zoom .7 + shift + flip | zoom .5 + shift | flip only   
BENCH_VIEWS = [(0.0, 0.0, 0.0, 1.0, 1.0), (40.0, -100.0, 0.0, 0.7, -1.0), (-60.0, 30.0, 0.0, 0.5, 1.0), (0.0, 0.0, 0.0, 1.0, -1.0)]


def synthetic_batches(batch, groups, views, size, device, seed=0, num_classes=19):
    """(source batch, target batch) of the cfg-3 shape: N(0,1) crops, random labels with a 16-px ignore
    border, target labels with 3 padded (-1) rows, frames2 = frames1 + 0.01*noise."""
    H, W = size
    gen = torch.Generator().manual_seed(seed)
    xs = torch.randn(batch, 3, H, W, generator=gen)
    ys = torch.randint(0, num_classes, (batch, H, W), generator=gen)
    ys[:, :16] = 255
    ys[:, -16:] = 255
    ys[:, :, :16] = 255
    ys[:, :, -16:] = 255
    B = groups * views
    f1 = torch.randn(B, 3, H, W, generator=gen)
    f2 = f1 + 0.01 * torch.randn(B, 3, H, W, generator=gen)
    gt = torch.randint(0, num_classes, (B, H, W), generator=gen)
    gt[:, :3] = -1
    sc = min(H, W) / 769.0
    params = [(dy * sc, dx * sc, a, s, f) for (dy, dx, a, s, f) in BENCH_VIEWS[:views]]
    theta, inv = view_affines(params, H, W)
    to = lambda t: t.to(device)
    return (to(xs), to(ys)), (to(f1), to(gt), to(f2), to(theta.repeat(groups, 1, 1)), to(inv.repeat(groups, 1, 1)))


def self_consistent_labels(net, images, border=16):
    """Source labels = the freshly initialised network's own argmax (with the ignore border): a
    converged-model regime, so that the reference's SGD hyper-parameters (LR 2.5e-4, x10 on the
    classifier, LR_TARGET 5) stay numerically stable on synthetic data for any number of steps."""
    core = net.backbone if hasattr(net, "backbone") else net
    was = core.training
    core.eval()
    with torch.no_grad():
        ys = torch.cat([core(images[i:i + 1])[1].argmax(1) for i in range(images.shape[0])], 0)
    core.train(was)
    ys[:, :border] = 255
    ys[:, -border:] = 255
    ys[:, :, :border] = 255
    ys[:, :, -border:] = 255
    return ys


def _classifier_layers(core):
    import torch.nn as nn
    return [m for n, m in core.named_modules()
            if isinstance(m, nn.Conv2d) and ("conv2d_list" in n or n.startswith(("score_pool", "vgg_head.8")))]


def calibrate_classifier(net, image, target_std=3.0):
    """Rescales the (linear) classifier layers of a synthetic-weight net so that its stride-8 logits have the given
    standard deviation on `image` -- peaked but unsaturated softmax whatever the backbone's feature scale is."""
    core = net.backbone if hasattr(net, "backbone") else net
    was = core.training
    core.eval()
    with torch.no_grad():
        std = float(core(image)[0].std())
        f = target_std / max(std, 1e-12)
        for m in _classifier_layers(core):
            m.weight.mul_(f)
            if m.bias is not None:
                m.bias.mul_(f)
    core.train(was)
    return f


# --------------------------------------------------------------------------------------------------
# checkpoints and validation
# --------------------------------------------------------------------------------------------------
def save_checkpoint(path, net, optim, score, epoch):
    """File layout of the reference's utils/checkpoints.py:62-74: {"model": state dict with the DDP
    "module." prefix, "opt": optimiser state, "score", "epoch"}."""
    core = net.module if hasattr(net, "module") else net
    model = {"module." + k: v for k, v in core.state_dict().items()}
    torch.save({"model": model, "opt": optim.state_dict() if optim is not None else None, "score": score, "epoch": epoch}, path)


def load_checkpoint(path, net, optim=None, map_location="cpu"):
    """utils/checkpoints.py:49-60: strict=False, so a baseline snapshot (backbone only) loads into SAC and
    leaves the teacher / class prior to the first `_momentum_update`.  Accepts keys with or without "module."."""
    blob = torch.load(path, map_location=map_location)
    core = net.module if hasattr(net, "module") else net
    model = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in blob["model"].items()}
    missing, unexpected = core.load_state_dict(model, strict=False)
    if optim is not None and blob.get("opt") is not None:
        optim.load_state_dict(blob["opt"])
    return blob.get("epoch", 0), blob.get("score", 0.0), missing, unexpected


# Cityscapes train id -> label id (cityscapesScripts `labels`; what infer_val.py:60-65 `convert_to_cs` applies)
CITYSCAPES_TRAIN_TO_ID = (7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33)


def infer_label_maps(net, image, lut=None, teacher=False, want_conf=False):
    """infer_val.py:160-163 + the writer's argmax / id mapping, without materialising logits_up or the softmax:
    backbone -> low-resolution logits -> ONE kernel -> uint8 label map [B,H,W] on the device (PNG writing stays on
    the host).  `lut`: uint8 tensor / sequence (e.g. CITYSCAPES_TRAIN_TO_ID) or None for train ids."""
    from dasac_hip import ops
    core = net.module if hasattr(net, "module") else net
    backbone = core
    if hasattr(core, "backbone"):
        backbone = core.slow_net if teacher else core.backbone
    if lut is not None and not torch.is_tensor(lut):
        lut = torch.tensor(list(lut), dtype=torch.uint8, device=image.device)
    with torch.no_grad():
        logits = backbone._logits(image)
        return ops.infer_labels(logits, image.shape[-2:], lut, want_conf)


def validation_iou(net, batches, num_classes=19):
    """mIoU over (image, label) batches: argmax + per-class tp/fp/fn in one kernel pass per batch
    (train.py:339-469, utils/metrics.py:9-53); counts are all-reduced when a process group exists."""
    import torch.distributed as dist
    from dasac_hip import ops
    core = net.module if hasattr(net, "module") else net
    was = core.training
    core.eval()
    counts = None
    with torch.no_grad():
        for image, gt in batches:
            _, logits_up = core(image)
            counts = ops.iou_counts(logits_up, gt, counts)
    core.train(was)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counts)
    iou, _, _ = summarise_iou(counts)
    return float(iou.mean()), iou


def summarise_iou(counts):
    """`Jaccard.summarise` (utils/metrics.py:40-53) on int64 counts [3, C] = (tp, fp, fn): per-class
    (jaccard, precision, recall) = tp / max(1e-3, .) in float32 like the reference (a class that never occurs scores 0)."""
    tp, fp, fn = (counts[i].to(torch.float32).cpu() for i in range(3))
    floor = torch.tensor(1e-3)
    return tp / torch.maximum(floor, fn + fp + tp), tp / torch.maximum(floor, tp + fp), tp / torch.maximum(floor, tp + fn)
