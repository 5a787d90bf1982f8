"""Golden-vector generator.  Runs ONLY in the build container (needs /root/reference):

    cd /root/repo && python -B tests/golden/make_goldens.py

Imports the reference (visinf/da-sac) on CPU, feeds it seeded synthetic inputs and stores
inputs + the reference's outputs as small .npz fixtures next to this file.  Network weights
are NOT stored: both sides regenerate them from `oracle.nets_ref.*_state(seed)` (pure
torch.Generator arithmetic), and the generator loads that state dict into the reference
module -- which also pins the checkpoint key layout.

G-numbers follow SURVEY.md 8(c).
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True

from ref_import import import_reference  # noqa: E402

ref_models, ref_cfg = import_reference()
from core.config import cfg_from_file  # noqa: E402  (reference)
from oracle import nets_ref, head_ref  # noqa: E402

torch.set_num_threads(8)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def make_sac(arch_yaml="deeplabv2_resnet101_train.yaml", **overrides):
    cfg_from_file(os.path.join("/root/reference/configs", arch_yaml))
    for k, v in overrides.items():
        setattr(ref_cfg.MODEL, k, v)
    crit = nn.CrossEntropyLoss(ignore_index=255, reduction="none")
    net = ref_models.get_model(ref_cfg.MODEL, 0, num_classes=19, criterion=crit)
    return net


def view_params():
    # (dy, dx, alpha, scale, flip): identity, zoom+shift+flip, zoom+shift, flip only
    return [(0.0, 0.0, 0.0, 1.0, 1.0), (4.0, -9.0, 0.0, 0.7, -1.0), (-6.0, 5.0, 0.0, 0.5, 1.0), (0.0, 0.0, 0.0, 1.0, -1.0)]


def g3_bilinear():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 5, 9, 13, generator=g)
    y = F.interpolate(x, (65, 97), mode="bilinear", align_corners=True)
    row = torch.randn(1, 1, 97, 2, generator=g)
    yrow = F.interpolate(row, (769, 9), mode="bilinear", align_corners=True)
    x2 = torch.randn(1, 3, 4, 6, generator=g)
    y2 = F.interpolate(x2, scale_factor=2, mode="bilinear", align_corners=True)   # fcn.py:109
    save("g3_bilinear", x=x, y=y, row=row, yrow=yrow, x2=x2, y2=y2)


def _target_inputs(N, T, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    B = N * T
    frames = torch.randn(B, 3, H, W, generator=g)
    logits = torch.randn(B, 19, (H - 1) // 8 + 1, (W - 1) // 8 + 1, generator=g) * 3.0
    ignore = torch.zeros(B, H, W, dtype=torch.bool)
    ignore[:, :3] = True
    ignore[1::T, :, -5:] = True
    # dataloader_target.py:220-262 through the reference's own (unbound) methods
    from datasets.dataloader_target import DataTarget
    cfg_from_file("/root/reference/configs/deeplabv2_resnet101_train.yaml")
    ref_cfg.DATASET.CROP_SIZE = [H, W]
    ref_cfg.TRAIN.GROUP_SIZE = T

    class _Shim:
        cfg = ref_cfg
    aff = DataTarget._get_affine(_Shim, view_params()[:T])
    inv = DataTarget._get_affine_inv(_Shim, aff, view_params()[:T])
    ref_cfg.DATASET.CROP_SIZE = [512, 1024]
    ref_cfg.TRAIN.GROUP_SIZE = 4
    return frames, logits, ignore, aff.repeat(N, 1, 1), inv.repeat(N, 1, 1)


def g4_refine():
    H, W, N, T = 33, 49, 2, 4
    frames, logits, ignore, aff, inv = _target_inputs(N, T, H, W, 4)
    out = dict(frames=frames, logits=logits, ignore=ignore, affine=aff, affine_inv=inv, T=T)
    for kind in ("avg_pool", "minentropy_pool"):
        net = make_sac(CONF_POOL=kind)
        net.train()
        net.running_conf.fill_(ref_cfg.MODEL.THRESHOLD_BETA)
        net.running_conf[3] = 0.07
        out["chi_in"] = net.running_conf.clone()
        refined, diags = net._refine(frames, logits.clone(), T, aff, inv, ignore)
        out[kind + "_refined"] = refined
        out[kind + "_chi_out"] = net.running_conf.clone()
        out[kind + "_teacher_aligned"] = diags["teacher_aligned"]
        out[kind + "_frames_aligned"] = diags["frames_aligned"]
    ref_cfg.MODEL.CONF_POOL = "avg_pool"
    # stand-alone warp goldens (ATen affine_grid + grid_sample, sac.py:289-301)
    grid = F.affine_grid(aff, size=(N * T, 19, H, W), align_corners=False)
    out["warp_ones_inv"] = F.grid_sample(torch.ones(N * T, 1, H, W), F.affine_grid(inv, size=(N * T, 1, H, W), align_corners=False), align_corners=False)
    out["warp_frames"] = F.grid_sample(frames, F.affine_grid(aff, size=frames.size(), align_corners=False), align_corners=False)
    save("g4_refine", **out)
    save("g9_affine", params=np.array(view_params(), dtype=np.float64), crop=np.array([H, W]), affine=aff[:T], affine_inv=inv[:T])


def g5_pseudo_labels():
    net = make_sac()
    g = torch.Generator().manual_seed(5)
    B, C, H, W = 3, 19, 21, 37
    probs = torch.softmax(torch.randn(B, C, H, W, generator=g) * 4.0, 1)
    probs[0, :, 2, 3] = 0.0                       # fully masked pixel -> k=0, m=0 -> 255
    probs[1, :, 5, 5] = 0.0
    probs[1, 4, 5, 5] = 0.5
    probs[1, 9, 5, 5] = 0.5                       # tie -> lowest index
    probs[2, :, 0, 0] = 0.0
    probs[2, 18, 0, 0] = 1.0                      # sets peak of class 18 to 1.0
    probs[2, :, 0, 1] = 0.0
    probs[2, 18, 0, 1] = 0.75                     # == UPPER*peak when disc==1: strict '>' fails
    ignore = torch.zeros(B, H, W, dtype=torch.bool)
    ignore[:, :2] = True
    chi = torch.rand(C, generator=g) * 0.004       # around beta so discounts spread over (0,1)
    chi[7] = 0.3
    out = dict(probs=probs, ignore=ignore, chi=chi)
    for disc in (True, False):
        net.running_conf.copy_(chi)
        lab, conf, idx = net._pseudo_labels_probs(probs.clone(), ignore, disc)
        tag = "disc" if disc else "nodisc"
        out["labels_" + tag] = lab
        out["conf_" + tag] = conf
        out["idx_" + tag] = idx
    out["discount"] = net._threshold_discount()
    out["upper"], out["lower"], out["beta"] = ref_cfg.MODEL.RUN_CONF_UPPER, ref_cfg.MODEL.RUN_CONF_LOWER, ref_cfg.MODEL.THRESHOLD_BETA
    save("g5_pseudo_labels", **out)


def g6_losses():
    net = make_sac()
    g = torch.Generator().manual_seed(6)
    B, C, H, W = 3, 19, 17, 23
    logits = (torch.randn(B, C, H, W, generator=g) * 2).requires_grad_(True)
    y = torch.randint(0, C, (B, H, W), generator=g)
    y[torch.rand(B, H, W, generator=g) < 0.3] = 255
    conf = torch.rand(B, 1, H, W, generator=g)
    chi = torch.rand(C, generator=g) * 0.3
    chi[2] = -0.05                                  # exercises clamp(0.)
    net.running_conf.copy_(chi)
    out = dict(logits=logits, y=y, conf=conf, chi=chi, p=3)
    loss, per_class = net._focal_ce_conf(logits, y, conf, 3)
    (grad,) = torch.autograd.grad(loss, logits)
    out.update(conf_loss=loss, conf_per_class=per_class, conf_grad=grad)
    loss, per_class = net._focal_ce(logits, y, conf, 3)
    (grad,) = torch.autograd.grad(loss.mean(), logits)
    out.update(plain_loss=loss, plain_per_class=per_class, plain_grad=grad)
    # backbone criterion path, deeplabv2.py:223-224
    crit = nn.CrossEntropyLoss(ignore_index=255, reduction="none")
    l2 = crit(logits, y).mean()
    (grad,) = torch.autograd.grad(l2, logits)
    out.update(ce_loss=l2, ce_grad=grad)
    save("g6_losses", **out)


def g7_state_sequences():
    net = make_sac()
    g = torch.Generator().manual_seed(7)
    net.train()
    seq = []
    probs_list = []
    net.running_conf.zero_()
    for it in range(4):
        probs = torch.softmax(torch.randn(2, 19, 9, 11, generator=g) * (1 + it), 1)
        if it == 1:
            net.running_conf.fill_(ref_cfg.MODEL.THRESHOLD_BETA)     # what _momentum_update does on init
        net._update_running_conf(probs)
        probs_list.append(probs)
        seq.append(net.running_conf.clone())
    # momentum update on a two-tensor toy pair through the reference's own code path:
    # init call, then distance-only, then update, then distance-only
    sd = nets_ref.resnet101_state(seed=11, randomize_bn=True)
    net.backbone.load_state_dict(sd, strict=True)
    net.slow_init[0] = 0.0
    r = [net._momentum_update(True).clone()]
    with torch.no_grad():
        for i, p in enumerate(net.backbone.parameters()):
            p.add_(0.01 * ((i % 7) - 3))
        net.backbone.model.bn1.running_mean.add_(0.5)
    r.append(net._momentum_update(False).clone())
    r.append(net._momentum_update(True).clone())
    r.append(net._momentum_update(False).clone())
    probe = net.slow_net.state_dict()
    save("g7_state", probs=torch.stack(probs_list), chi_seq=torch.stack(seq), diffs=torch.cat(r),
         slow_conv1=probe["model.conv1.weight"], slow_bn1_mean=probe["model.bn1.running_mean"],
         slow_l3_w=probe["model.layer3.5.conv2.weight"][:4, :4],
         chi_after_init=net.running_conf.clone(), momentum=ref_cfg.MODEL.NET_MOMENTUM)


def _sampled(t, n=64):
    flat = t.detach().reshape(-1)
    m = min(n, flat.numel())
    idx = (torch.arange(m, dtype=torch.int64) * (flat.numel() - 1)) // max(m - 1, 1)
    return flat[idx]


GRAD_PROBE_KEYS = ["model.conv1.weight", "model.bn1.weight", "model.bn1.bias", "model.layer1.0.conv1.weight",
                   "model.layer1.0.downsample.0.weight", "model.layer2.0.conv1.weight", "model.layer2.3.bn2.weight",
                   "model.layer3.0.conv2.weight", "model.layer3.11.bn3.bias", "model.layer3.22.conv3.weight",
                   "model.layer4.2.conv2.weight", "model.layer4.0.downsample.1.weight",
                   "model.layer5.conv2d_list.0.weight", "model.layer5.conv2d_list.3.bias"]


def g2_resnet():
    crit = nn.CrossEntropyLoss(ignore_index=255, reduction="none")
    out = {}
    for tag, freeze, (B, H, W) in (("eval", True, (1, 65, 65)), ("train", False, (2, 33, 49))):
        net = ref_models.DeepLabV2_ResNet101(num_classes=19, criterion=crit, freeze_bn=freeze)
        sd = nets_ref.resnet101_state(seed=2, randomize_bn=True, he_init=True, residual_gain=0.25, aspp_gain=0.2)
        missing = net.load_state_dict(sd, strict=True)
        net.train()
        g = torch.Generator().manual_seed(20 + B)
        x = torch.randn(B, 3, H, W, generator=g)
        y = torch.randint(0, 19, (B, H, W), generator=g)
        y[:, :2] = 255
        losses, outs = net(x, y)
        losses["loss_ce"].backward()
        named = dict(net.named_parameters())
        out[tag + "_x"], out[tag + "_y"] = x, y
        out[tag + "_logits"] = outs["logits"]
        out[tag + "_logits_up_s"] = _sampled(outs["logits_up"], 512)
        out[tag + "_loss"] = losses["loss_ce"]
        for k in GRAD_PROBE_KEYS:
            out[tag + "_g_" + k] = _sampled(named[k].grad, 64)
            out[tag + "_gn_" + k] = named[k].grad.norm()
            out[tag + "_gm_" + k] = named[k].grad.abs().max()
        if not freeze:
            st = net.state_dict()
            out["train_rm_bn1"] = st["model.bn1.running_mean"]
            out["train_rv_l3"] = st["model.layer3.4.bn2.running_var"]
            out["train_nbt"] = st["model.bn1.num_batches_tracked"]
    save("g2_resnet101", **out)


def g10_vgg():
    crit = nn.CrossEntropyLoss(ignore_index=255, reduction="none")
    net = ref_models.DeepLabV2_VGG16(num_classes=19, criterion=crit, use_bn=True, freeze_bn=True)
    sd = nets_ref.deeplab_vgg16_state(seed=10, randomize_bn=True)
    net.load_state_dict(sd, strict=True)
    net.train()
    g = torch.Generator().manual_seed(10)
    x = torch.randn(1, 3, 321, 321, generator=g)
    y = torch.randint(0, 19, (1, 321, 321), generator=g)
    losses, outs = net(x, y)
    losses["loss_ce"].backward()
    named = dict(net.named_parameters())
    save("g10_vgg16_deeplab", x_seed=10, logits=outs["logits"], loss=losses["loss_ce"],
         logits_up_s=_sampled(outs["logits_up"], 512),
         g_first=_sampled(named["features.0.weight"].grad, 64),
         g_fc6=_sampled(named["features.42.weight"].grad, 64),
         g_cls_bias=named["classifier.conv2d_list.2.bias"].grad,
         keys=np.array(sorted(net.state_dict().keys())))

    net = ref_models.VGG16_FCN8s(19, criterion=crit, use_bn=True, freeze_bn=True, drop_rate=0.0)
    sd = nets_ref.fcn8s_vgg16_state(seed=12, randomize_bn=True)
    net.load_state_dict(sd, strict=True)
    net.train()
    x = torch.randn(1, 3, 64, 96, generator=g)
    y = torch.randint(0, 19, (1, 64, 96), generator=g)
    losses, outs = net(x, y)
    losses["loss_ce"].backward()
    named = dict(net.named_parameters())
    save("g10_fcn8s", x=x, y=y, logits_up=outs["logits_up"], loss=losses["loss_ce"],
         g_head0=_sampled(named["vgg_head.0.weight"].grad, 64), g_sp3=named["score_pool3.weight"].grad.reshape(-1)[:64],
         g_first=_sampled(named["block1.0.weight"].grad, 64),
         keys=np.array(sorted(net.state_dict().keys())))


def g8_two_steps():
    """Two iterations of train.py:266-298 (SAC mode, one rank) on the reference module with
    torch.optim.SGD over the reference's own parameter groups."""
    H, W, N, T, Bs = 33, 49, 1, 4, 2
    net = make_sac()
    sd = nets_ref.resnet101_state(seed=8, randomize_bn=True, he_init=True, residual_gain=0.25, aspp_gain=0.2)
    net.backbone.load_state_dict(sd, strict=True)
    net.train()
    groups = net.parameter_groups(ref_cfg.MODEL.LR, ref_cfg.MODEL.WEIGHT_DECAY)
    optim = torch.optim.SGD(groups, momentum=ref_cfg.MODEL.MOMENTUM, nesterov=False)
    g = torch.Generator().manual_seed(8)
    rec = dict(H=H, W=W, N=N, T=T, Bs=Bs)
    probe_keys = ["model.conv1.weight", "model.layer3.7.conv2.weight", "model.layer5.conv2d_list.1.weight", "model.bn1.bias",
                  "model.layer5.conv2d_list.2.bias", "model.layer4.1.bn2.weight"]
    for it in range(2):
        xs = torch.randn(Bs, 3, H, W, generator=g)
        ys = torch.randint(0, 19, (Bs, H, W), generator=g)
        ys[:, :2, :] = 255
        f1, _, ignore, aff, inv = _target_inputs(N, T, H, W, 80 + it)
        f2 = f1 + 0.01 * torch.randn(f1.shape, generator=g)
        gt = torch.randint(0, 19, (N * T, H, W), generator=g)
        gt[ignore] = -1
        rec.update({"it%d_xs" % it: xs, "it%d_ys" % it: ys, "it%d_f1" % it: f1, "it%d_f2" % it: f2,
                    "it%d_gt" % it: gt.clone(), "affine": aff, "affine_inv": inv})
        # --- train.py:119-138
        losses, _ = net(xs, ys)
        optim.zero_grad()
        losses["loss_ce"].mean().backward()
        rec["it%d_src_loss" % it] = losses["loss_ce"].detach().clone()
        # --- train.py:211-233
        losses, outs = net(f1, gt, f2, aff, inv, use_teacher=True, update_teacher=(it % 100 == 0), T=T)
        (ref_cfg.MODEL.LR_TARGET * losses["self_ce"].mean()).backward()
        optim.step()
        for k, v in losses.items():
            rec["it%d_%s" % (it, k)] = v.detach().clone()
        rec["it%d_labels" % it] = outs["teacher_labels"].to(torch.uint8)
        rec["it%d_conf_s" % it] = _sampled(outs["teacher_conf"], 256)
        rec["it%d_chi" % it] = net.running_conf.clone()
        st = net.backbone.state_dict()
        for k in probe_keys:
            rec["it%d_p_%s" % (it, k)] = _sampled(st[k], 64)
            rec["it%d_pn_%s" % (it, k)] = st[k].norm()
    rec["n_labelled"] = sum(int((rec["it%d_labels" % i] != 255).sum()) for i in range(2))
    save("g8_two_steps", **rec)
    print("labelled pixels:", rec["n_labelled"])


def keys_fixture():
    net = make_sac()
    sd = net.state_dict()
    names = sorted(sd.keys())
    shapes = ["x".join(str(s) for s in sd[k].shape) for k in names]
    groups = net.parameter_groups(1.0, 1.0)
    id2name = {id(p): n for n, p in net.named_parameters()}
    gk = {"group%d" % i: np.array([id2name[id(p)] for p in g["params"]]) for i, g in enumerate(groups)}
    save("keys_sac_resnet101", names=np.array(names), shapes=np.array(shapes),
         group_lr=np.array([g["lr"] for g in groups]), group_wd=np.array([g["weight_decay"] for g in groups]), **gk)


# ------------------------------------------------------------------------------------------------
# G11: rank -> view index tables of Trainer._prep_batch (train.py:157-209) and SAC._gather
# (models/sac.py:198-216), captured by running the reference's own (unbound) methods in `world`
# gloo processes on CPU.  train.py imports at module level a few packages this image lacks
# (setproctitle, tensorboard, torchvision.utils); none of them is touched by the two methods, so
# empty module objects of those names are registered first.
# ------------------------------------------------------------------------------------------------
G11_CASES = [(1, 2, 4), (2, 1, 2), (2, 2, 4), (4, 2, 4), (4, 1, 4), (8, 2, 4), (8, 4, 4), (8, 16, 4)]   # (world, N, L)


class _StaysOnCpu(torch.Tensor):
    """`tensor.cuda(gpu)` (train.py:183) is a no-op in this GPU-less container."""

    def cuda(self, *a, **k):
        return self


def _g11_rank(rank, world, port, cases, q):
    import types
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for name in ("setproctitle", "torch.utils.tensorboard", "torchvision.utils"):
        m = types.ModuleType(name)
        m.SummaryWriter = object
        sys.modules.setdefault(name, m)
    import torchvision
    torchvision.utils = sys.modules["torchvision.utils"]
    import train as ref_train                      # the reference's train.py
    from models.sac import SAC as RefSAC
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {}
    for (w, N, L) in cases:
        ref_cfg.TRAIN.NUM_GROUPS, ref_cfg.TRAIN.GROUP_SIZE = N, L
        shim = types.SimpleNamespace(cfg=ref_cfg, world_size=world, gpu=rank, rank=rank)
        Bl = max(1, N // world)                   # datasets/__init__.py:66 loader batch of target images
        b = torch.arange(Bl).view(Bl, 1, 1)
        t = torch.arange(L).view(1, L, 1)
        loaded = (rank * 1000 + b * 10 + t).expand(Bl, L, 2).contiguous().float().as_subclass(_StaysOnCpu)
        got = ref_train.Trainer._prep_batch(shim, loaded)
        got = torch.Tensor(got)[:, 0].long() if got.dim() == 2 else got.as_subclass(torch.Tensor)[..., 0].long()
        out["prep_w%d_N%d_L%d" % (w, N, L)] = got.numpy()
        B = got.shape[0]
        mine = (rank * 100 + torch.arange(B)).float().view(B, 1)
        gathered = RefSAC._gather(shim, mine, L)
        out["gather_w%d_N%d_L%d" % (w, N, L)] = gathered[:, 0].long().numpy()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def g11_index_tables():
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    rec = {}
    for world in sorted({c[0] for c in G11_CASES}):
        cases = [c for c in G11_CASES if c[0] == world]
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        q = ctx.Queue()
        procs = [ctx.Process(target=_g11_rank, args=(r, world, port, cases, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = dict(q.get(timeout=600) for _ in procs)
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
        for key in res[0]:
            rec[key] = np.stack([res[r][key] for r in range(world)])       # [world, ...]
    rec["cases"] = np.array(G11_CASES)
    save("g11_index_tables", **rec)


# ------------------------------------------------------------------------------------------------
# G12: the K augmented views of one target crop through the reference's own transform classes
# (datasets/tf_target.py: GuidedRandHFlip :141-157, MaskRandScaleCrop :159-239, ToTensorMask/Normalize/ApplyMask
# :33-98) on PIL images, python `random` seeded.  torchvision is absent here; the four functional helpers the
# classes call are one-line wrappers over Pillow / torch and are stood in for as such (crop, hflip, pad, to_tensor).
# ------------------------------------------------------------------------------------------------
def _install_tv_functional():
    from PIL import Image, ImageOps
    F_ = sys.modules["torchvision.transforms.functional"]
    F_.crop = lambda img, top, left, h, w: img.crop((left, top, left + w, top + h))
    F_.hflip = lambda img: img.transpose(Image.FLIP_LEFT_RIGHT)
    F_.pad = lambda img, padding, fill=0, padding_mode="constant": ImageOps.expand(img, border=tuple(padding), fill=fill)
    F_.to_tensor = lambda pic: torch.from_numpy(np.array(pic, np.uint8, copy=True)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)


def g12_views():
    import random
    from PIL import Image
    _install_tv_functional()
    import datasets.tf_target as tft
    from datasets.dataloader_target import DataTarget
    from datasets.dataloader_base import DLBase

    class _Numpy1:
        """tf_target.py:36 calls np.array(pic, np.int32, copy=False), which NumPy 2 (this image: 2.2) rejects when a copy
        is needed; NumPy 1.x -- what the reference was written for -- copied silently (= np.asarray)."""

        def __getattr__(self, name):
            return getattr(np, name)

        @staticmethod
        def array(obj, dtype=None, copy=True):
            return np.array(obj, dtype) if copy else np.asarray(obj, dtype)
    tft.np = _Numpy1()
    base = DLBase()
    rec = dict(mean=np.array(base.MEAN), std=np.array(base.STD))
    H, W, L = 64, 96, 4
    for case, (zoom, seed) in enumerate((((0.5, 1.0), 121), ((0.5, 1.2), 7), ((0.5, 1.0), 5))):
        gen = np.random.RandomState(seed)
        # a smooth image (so that bilinear taps matter), block labels, a padded margin in the mask (MaskRandCrop fill=1)
        yy, xx = np.mgrid[0:H, 0:W]
        img = np.stack([(127 + 120 * np.sin(xx / (3.0 + c) + yy / 5.0) + gen.randint(-6, 7, (H, W))).clip(0, 255) for c in range(3)], -1).astype(np.uint8)
        lab = (gen.randint(0, 19, (H // 8, W // 8)).repeat(8, 0).repeat(8, 1)).astype(np.uint8)
        msk = np.zeros((H, W), np.uint8)
        msk[:, W - 7:] = 1
        msk[:3] = 1
        images = [Image.fromarray(img) for _ in range(L)]
        labels = [Image.fromarray(lab, "L") for _ in range(L)]
        masks = [Image.fromarray(msk, "L") for _ in range(L)]
        random.seed(1000 + seed)
        out = tft.GuidedRandHFlip()(images, labels, masks)
        images, labels, masks, params = tft.MaskRandScaleCrop(list(zoom))(*out)
        u8 = np.stack([np.array(im) for im in images])
        lab_u8 = np.stack([np.array(x) for x in labels])
        msk_u8 = np.stack([np.array(x) for x in masks])
        post = tft.Compose([tft.ToTensorMask(), tft.Normalize(mean=base.MEAN, std=base.STD), tft.ApplyMask(-1)])
        frames, gts = post(images, labels, masks)
        cfg_from_file("/root/reference/configs/deeplabv2_resnet101_train.yaml")
        ref_cfg.DATASET.CROP_SIZE, ref_cfg.TRAIN.GROUP_SIZE = [H, W], L

        class _Shim:
            cfg = ref_cfg
        aff = DataTarget._get_affine(_Shim, params)
        inv = DataTarget._get_affine_inv(_Shim, aff, params)
        ref_cfg.DATASET.CROP_SIZE, ref_cfg.TRAIN.GROUP_SIZE = [512, 1024], 4
        t = "c%d_" % case
        rec.update({t + "image": img, t + "label": lab, t + "mask": msk, t + "zoom": np.array(zoom), t + "seed": 1000 + seed,
                    t + "params": np.array(params, dtype=np.float64), t + "views_u8": u8, t + "labels_u8": lab_u8, t + "masks_u8": msk_u8,
                    t + "gt": torch.stack(gts).to(torch.int16), t + "affine": aff, t + "affine_inv": inv})
        if case < 2:
            rec[t + "frames"] = torch.stack(frames)
        print("case", case, "params", params)
    rec["n_cases"] = 3
    save("g12_views", **rec)


# ------------------------------------------------------------------------------------------------
# G13: the photometric augmentations of the student's views through the reference's own classes
# (datasets/tf_target.py: RandGaussianBlur :331-349, MaskRandJitter :365-390, MaskRandGreyscale :351-363) on PIL
# images, python `random` and torch's global RNG seeded.  Blur and greyscale are Pillow end to end.  MaskRandJitter
# delegates to torchvision.transforms.ColorJitter, and torchvision is ABSENT from this image (the reference pins no
# version): `_ColorJitterStandIn` below restates its published control flow (torchvision >= 0.8: get_params draws
# torch.randperm(4) and four torch uniforms; functional_pil applies PIL.ImageEnhance.{Brightness,Contrast,Color} and the
# HSV hue shift) -- the pixel arithmetic inside it is the real Pillow's.  So this fixture pins blur / greyscale / the
# ImageEnhance + HSV arithmetic and the reference's own draw order and probabilities; ColorJitter's internal draw order
# is restated, not pinned.
# ------------------------------------------------------------------------------------------------
class _ColorJitterStandIn:
    def __init__(self, brightness=0, contrast=0, saturation=0, hue=0):
        self.brightness = [max(0., 1. - brightness), 1. + brightness] if brightness else None
        self.contrast = [max(0., 1. - contrast), 1. + contrast] if contrast else None
        self.saturation = [max(0., 1. - saturation), 1. + saturation] if saturation else None
        self.hue = [-hue, hue] if hue else None
        self.log = []

    def __call__(self, img):
        from PIL import Image, ImageEnhance
        order = torch.randperm(4).tolist()
        fac = [None if r is None else float(torch.empty(1).uniform_(r[0], r[1])) for r in (self.brightness, self.contrast, self.saturation, self.hue)]
        self.log.append((order, fac))
        for k in order:
            if fac[k] is None:
                continue
            if k == 0:
                img = ImageEnhance.Brightness(img).enhance(fac[k])
            elif k == 1:
                img = ImageEnhance.Contrast(img).enhance(fac[k])
            elif k == 2:
                img = ImageEnhance.Color(img).enhance(fac[k])
            else:
                h, s_, v = img.convert("HSV").split()
                np_h = np.array(h, dtype=np.uint8)
                with np.errstate(over="ignore"):
                    np_h += np.int32(fac[k] * 255).astype(np.uint8)
                img = Image.merge("HSV", (Image.fromarray(np_h, "L"), s_, v)).convert("RGB")
        return img


def g13_photometric():
    import random
    from PIL import Image
    _install_tv_functional()
    F_ = sys.modules["torchvision.transforms.functional"]

    def to_grayscale(img, num_output_channels=1):          # torchvision functional_pil.to_grayscale
        img = img.convert("L")
        if num_output_channels == 3:
            a = np.array(img, dtype=np.uint8)
            img = Image.fromarray(np.dstack([a, a, a]), "RGB")
        return img
    F_.to_grayscale = to_grayscale
    sys.modules["torchvision.transforms"].ColorJitter = _ColorJitterStandIn
    import datasets.tf_target as tft
    rec = {}
    H, W, L = 48, 80, 4
    cases = ((11, 0.4, 0.2), (12, 0.4, 0.9), (13, 0.5, 0.0), (14, 0.05, 0.5))      # seed, jitter, greyscale p
    for case, (seed, jitter, grey_p) in enumerate(cases):
        gen = np.random.RandomState(seed)
        yy, xx = np.mgrid[0:H, 0:W]
        img = np.stack([(127 + 110 * np.sin(xx / (2.0 + c) + yy / (4.0 + c)) + gen.randint(-20, 21, (H, W))).clip(0, 255) for c in range(3)], -1).astype(np.uint8)
        img[:6, :9] = (255, 0, 0)
        img[-5:, -7:] = 128
        images = [Image.fromarray(img) for _ in range(L)]
        random.seed(2000 + seed)
        torch.manual_seed(2000 + seed)
        blur, jit, grey = tft.RandGaussianBlur(), tft.MaskRandJitter(jitter), tft.MaskRandGreyscale(grey_p)
        state = random.getstate()
        out, _, _ = tft.Compose([blur, jit, grey])(images, [None] * L, [None] * L)
        t = "c%d_" % case
        rec.update({t + "image": img, t + "seed": 2000 + seed, t + "jitter": jitter, t + "grey_p": grey_p,
                    t + "out_u8": np.stack([np.array(im) for im in out])})
        # the draws, replayed from the saved generator state in the classes' order, for the test's sampler to reproduce
        random.setstate(state)
        radii = [random.uniform(.1, 2.) for _ in range(L)]
        hits = [random.random() < 0.5 for _ in range(L)]
        greys = [grey_p > random.random() for _ in range(L)]
        assert sum(hits) == len(jit.jitter.log)
        rec[t + "radii"] = np.array(radii)
        rec[t + "jitter_on"] = np.array(hits)
        rec[t + "grey_on"] = np.array(greys)
        rec[t + "jitter_order"] = np.array([o for o, _ in jit.jitter.log], dtype=np.int64).reshape(-1, 4)
        rec[t + "jitter_factors"] = np.array([f for _, f in jit.jitter.log], dtype=np.float64).reshape(-1, 4)
        print("case", case, "radii", np.round(radii, 3), "jitter", hits, "grey", greys)
    rec["n_cases"] = len(cases)
    save("g13_photometric", **rec)


def g14_jaccard():
    """utils/metrics.py:9-53 `Jaccard` (the validation metric of train.py:339-469) over three batches: ignore pixels, a class
    that never occurs (17: absent from gt and from the predictions), one that is predicted but never true (18), and a few
    gt = -1 pixels (counted as false positives of whatever is predicted there).  The class moves its counters to a GPU in
    __init__ (`.cuda(gpu)`); on this CPU-only container that one call is made a no-op for the construction."""
    # (the reference is already importable: import_reference() ran at module load)
    from utils.metrics import Jaccard
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        metric = Jaccard(19, 0)
    finally:
        torch.Tensor.cuda = real_cuda
    g = torch.Generator().manual_seed(14)
    rec = {}
    for b in range(3):
        logits = torch.randn(2, 19, 23, 31, generator=g) * 2
        logits[:, 17] = -1e3
        gt = torch.randint(0, 17, (2, 23, 31), generator=g)
        gt[torch.rand(2, 23, 31, generator=g) < 0.15] = 255
        gt[0, :2, :5] = -1
        if b == 1:
            gt[1] = 255                                   # a fully ignored image
        pred = logits.argmax(1)
        metric.add_sample(pred.clone(), gt.clone())       # add_sample edits its prediction argument in place
        rec.update({"logits%d" % b: logits.numpy(), "gt%d" % b: gt.numpy(),
                    "counts%d" % b: torch.stack([metric.tps, metric.fps, metric.fns]).to(torch.int64).numpy()})
    j, p, r = metric.summarise()
    rec.update(jaccards=j.numpy(), precision=p.numpy(), recall=r.numpy())
    print("g14: mIoU", float(j.mean()), "tp", metric.tps.sum().item())
    save("g14_jaccard", **rec)


def g15_inference():
    """infer_val.py:160-166 + the result writer's label maps (:78-88): `_, logits = model(image)` ends in deeplabv2.py:217
    `F.interpolate(logits, orig_size, mode="bilinear", align_corners=True)`, then `F.softmax(logits, 1)`, `np.argmax(masks_raw, 0)
    .astype(np.uint8)` and `convert_to_cs` (:62-67) over `tools.category.labels`.  infer_val.py itself cannot be imported here
    (imageio is absent), so the generator compiles the reference's OWN `convert_to_cs` function definition out of its source
    file and binds it to the reference's own label table; the three ATen / numpy call lines are restated verbatim.  Stored:
    low-resolution logits, the train-id and label-id maps, the winning probability and the train id -> label id table that
    `convert_to_cs` realises for ids 0..18."""
    import ast
    from tools.category import labels as CS_LABELS            # reference
    src = open("/root/reference/infer_val.py").read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "convert_to_cs"]
    assert len(fn) == 1
    ns = {"np": np, "CS_LABELS": CS_LABELS}
    exec(compile(ast.Module(body=fn, type_ignores=[]), "/root/reference/infer_val.py", "exec"), ns)
    convert_to_cs = ns["convert_to_cs"]
    g = torch.Generator().manual_seed(15)
    rec = {}
    for name, (B, C, h, w, H, W) in dict(a=(2, 19, 23, 31, 177, 241), b=(1, 19, 33, 65, 257, 513)).items():
        logits_low = torch.randn(B, C, h, w, generator=g) * 3
        logits = F.interpolate(logits_low, (H, W), mode="bilinear", align_corners=True)         # deeplabv2.py:217
        masks_pred = F.softmax(logits, 1)                                                        # infer_val.py:162
        preds, preds_cs = [], []
        for i in range(B):
            masks_raw = masks_pred[i].cpu().numpy()                                              # :80
            pred = np.argmax(masks_raw, 0).astype(np.uint8)                                      # :81
            preds.append(pred)
            preds_cs.append(convert_to_cs(pred))                                                 # :86
        top2 = masks_pred.topk(2, dim=1).values
        rec.update({"logits_" + name: logits_low.numpy(), "size_" + name: np.array([H, W]), "pred_" + name: np.stack(preds),
                    "pred_cs_" + name: np.stack(preds_cs), "conf_" + name: top2[:, 0].numpy().astype(np.float16),
                    # pixels whose two best classes are closer than 1e-4 (fp32 noise of another upsampling order may flip them)
                    "tie_" + name: np.packbits((top2[:, 0] - top2[:, 1]).numpy() < 1e-4)})
    lut = convert_to_cs(np.arange(19, dtype=np.uint8))
    print("g15: train id -> label id", lut.tolist())
    save("g15_inference", lut=lut, **rec)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g3", "g4", "g5", "g6", "g7", "g2", "g10", "g8", "keys", "g11", "g12", "g13", "g14", "g15"]
    table = dict(g3=g3_bilinear, g4=g4_refine, g5=g5_pseudo_labels, g6=g6_losses, g7=g7_state_sequences,
                 g2=g2_resnet, g10=g10_vgg, g8=g8_two_steps, keys=keys_fixture, g11=g11_index_tables, g12=g12_views,
                 g13=g13_photometric, g14=g14_jaccard, g15=g15_inference)
    for w in which:
        table[w]()
