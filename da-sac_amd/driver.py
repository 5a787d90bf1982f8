"""Minimal training driver reproducing the reference's step order on one rank
(/root/reference/train.py:119-155 source step, :211-250 target step, :266-298 loop body,
/root/reference/base_trainer.py:47-73 optimiser factory).  Works on the bare module or on a
DistributedDataParallel wrapper (backend "nccl" == RCCL on ROCm)."""
import torch


def make_optimizer(net, cfg_model):
    """SGD(momentum, no nesterov) over the model's four parameter groups (base_trainer.py:63-66)."""
    core = net.module if hasattr(net, "module") else net
    groups = core.parameter_groups(cfg_model.LR, cfg_model.WEIGHT_DECAY)
    return torch.optim.SGD(groups, momentum=cfg_model.MOMENTUM, nesterov=getattr(cfg_model, "OPT_NESTEROV", False))


def sac_train_iteration(net, optim, src_batch, tgt_batch, group_size, update_teacher, lr_target, target_only=False):
    """source fwd -> zero_grad -> source bwd (gradients kept) -> target fwd (teacher EMA first when asked)
    -> (LR_TARGET * self_ce) bwd -> one optimiser step.  Returns (source losses, target losses, net_outs)
    with the losses still on the device (no host sync here)."""
    images, masks = src_batch
    losses_src, _ = net(images, masks)
    optim.zero_grad()
    losses_src["loss_ce"].mean().backward()
    frames1, frames_gt, frames2, affine, affine_inv = tgt_batch
    losses_tgt, outs = net(frames1, frames_gt, frames2, affine, affine_inv, use_teacher=True,
                           update_teacher=update_teacher, T=group_size)
    if target_only:
        optim.zero_grad()
    (lr_target * losses_tgt["self_ce"].mean()).backward()
    optim.step()
    return losses_src, losses_tgt, outs


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
