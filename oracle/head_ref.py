"""Oracle (CPU, test infrastructure) for the SAC head: everything between the backbone's
stride-8 logits and the self-supervised loss.  Closed-form restatements; each function
names the reference lines (relative to /root/reference) it follows.

All tensors fp32 NCHW / int64 labels, exactly as the reference passes them.
"""
import math

import torch

IGNORE = 255


# --------------------------------------------------------------------------------------
# bilinear upsampling, align_corners=True   (models/deeplabv2.py:217, models/sac.py:275)
# ATen: upsample_bilinear2d, scale = (in-1)/(out-1), src = scale*dst, i1 = min(i0+1, in-1)
# --------------------------------------------------------------------------------------
def _axis_taps_ac(n_in, n_out):
    scale = torch.tensor((n_in - 1) / (n_out - 1) if n_out > 1 else 0.0, dtype=torch.float32)
    dst = torch.arange(n_out, dtype=torch.float32)
    src = scale * dst
    i0 = src.to(torch.int64).clamp_(max=n_in - 1)
    i1 = (i0 + 1).clamp_(max=n_in - 1)
    w1 = src - i0.to(torch.float32)
    w0 = 1.0 - w1
    return i0, i1, w0, w1


def upsample_bilinear_ac(x, out_h, out_w):
    """x [B,C,h,w] -> [B,C,out_h,out_w]; out = wh0*(ww0*a + ww1*b) + wh1*(ww0*c + ww1*d)."""
    _, _, h, w = x.shape
    r0, r1, a0, a1 = _axis_taps_ac(h, out_h)
    c0, c1, b0, b1 = _axis_taps_ac(w, out_w)
    top, bot = x[:, :, r0], x[:, :, r1]
    t = b0 * top[..., c0] + b1 * top[..., c1]
    b = b0 * bot[..., c0] + b1 * bot[..., c1]
    return a0[:, None] * t + a1[:, None] * b


# --------------------------------------------------------------------------------------
# affine_grid + grid_sample (bilinear, zeros padding, align_corners=False)
# (models/sac.py:289-290,295-296,300-301,309-310)
# --------------------------------------------------------------------------------------
def affine_source_coords(theta, H, W):
    """Pixel-space sampling coordinates (ix, iy), each [B,H,W], for theta [B,2,3].

    affine_grid(ac=False): base x_j = (2j+1)/W - 1;  g = theta @ (x, y, 1)
    grid_sample(ac=False): ix = ((gx+1)*W - 1)/2
    """
    xb = (2.0 * torch.arange(W, dtype=torch.float32) + 1.0) / W - 1.0
    yb = (2.0 * torch.arange(H, dtype=torch.float32) + 1.0) / H - 1.0
    th = theta.to(torch.float32)
    gx = th[:, 0, 0, None, None] * xb[None, None, :] + th[:, 0, 1, None, None] * yb[None, :, None] \
        + th[:, 0, 2, None, None]
    gy = th[:, 1, 0, None, None] * xb[None, None, :] + th[:, 1, 1, None, None] * yb[None, :, None] \
        + th[:, 1, 2, None, None]
    ix = ((gx + 1.0) * W - 1.0) / 2.0
    iy = ((gy + 1.0) * H - 1.0) / 2.0
    return ix, iy


def _corner_taps(ix, iy, H, W):
    """Four (flat index, weight) taps with the zero-padding rule folded into the weight."""
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    x1, y1 = x0 + 1.0, y0 + 1.0
    wx1, wx0 = ix - x0, x1 - ix
    wy1, wy0 = iy - y0, y1 - iy
    taps = []
    for (yy, wy) in ((y0, wy0), (y1, wy1)):
        for (xx, wx) in ((x0, wx0), (x1, wx1)):
            inb = (xx >= 0) & (xx <= W - 1) & (yy >= 0) & (yy <= H - 1)
            wgt = torch.where(inb, wx * wy, torch.zeros_like(wx))
            idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).to(torch.int64)
            taps.append((idx, wgt))
    return taps


def warp_affine(x, theta):
    """grid_sample(x, affine_grid(theta)) for x [B,C,H,W]."""
    B, C, H, W = x.shape
    ix, iy = affine_source_coords(theta, H, W)
    flat = x.reshape(B, C, H * W)
    out = torch.zeros_like(flat)
    for idx, wgt in _corner_taps(ix, iy, H, W):
        idx = idx.reshape(B, 1, H * W).expand(B, C, H * W)
        out = out + torch.gather(flat, 2, idx) * wgt.reshape(B, 1, H * W)
    return out.reshape(B, C, H, W)


def warp_coverage(theta, H, W):
    """grid_sample(ones, affine_grid(theta)) -> [B,1,H,W]: sum of the in-bounds bilinear
    weights (models/sac.py:299-301)."""
    ix, iy = affine_source_coords(theta, H, W)
    cov = torch.zeros_like(ix)
    for _, wgt in _corner_taps(ix, iy, H, W):
        cov = cov + wgt
    return cov[:, None]


# --------------------------------------------------------------------------------------
# class prior  (models/sac.py:104-117)
# --------------------------------------------------------------------------------------
def update_running_conf(running_conf, probs, beta, stat_momentum, tolerance=1e-8):
    """Returns the new chi[C].  Classes still sitting exactly at beta adopt the batch mean
    first (:111-113), then everybody takes the EMA step (:116-117)."""
    B, C, H, W = probs.shape
    avg = probs.mean(0).view(C, -1).mean(-1)
    chi = running_conf.clone()
    fresh = (avg > tolerance) & (chi == beta)
    chi[fresh] = avg[fresh]
    chi = chi * stat_momentum
    chi = chi + (1 - stat_momentum) * avg
    return chi


def threshold_discount(running_conf, beta):
    """1 - exp(-chi/beta)  (models/sac.py:151-152)."""
    return 1.0 - torch.exp(-running_conf / beta)


def focal_weight(running_conf, p):
    """(1 - max(chi,0))**p  (models/sac.py:120,135)."""
    return (1 - running_conf.clamp(0.0)) ** p


# --------------------------------------------------------------------------------------
# pseudo labels  (models/sac.py:154-187) -- integer output, BIT-EXACT contract
# --------------------------------------------------------------------------------------
def pseudo_labels(probs, ignore_augm, upper, lower, disc=None):
    """probs [B,C,H,W] fp32, ignore_augm bool [B,H,W], disc [C] or None.

    (m,k) = max_c probs (ties -> lowest c); peak[b,c] = max{m : k==c} (0 if class absent);
    thr = clamp_min(peak*upper*disc, lower) in that op order (:168-174);
    label = k if m > thr[b,k] else 255 (:175-182); augmentation padding -> 255 (:185).
    Requires lower > 0 (then the `sum != 1` test of :178 reduces to the single compare).
    Returns labels int64 [B,H,W], max_conf fp32 [B,1,H,W], max_idx int64 [B,1,H,W].
    """
    assert lower > 0
    B, C, H, W = probs.shape
    m, k = probs.max(1)                                   # [B,H,W]
    peak = torch.zeros(B, C, dtype=probs.dtype)
    peak.scatter_reduce_(1, k.view(B, -1), m.view(B, -1), reduce="amax", include_self=True)
    thr = peak * upper
    if disc is not None:
        thr = thr * disc.view(1, C)
    thr = thr.clamp_min(lower)
    keep = m > torch.gather(thr, 1, k.view(B, -1)).view(B, H, W)
    labels = torch.where(keep, k, torch.full_like(k, IGNORE))
    labels[ignore_augm] = IGNORE
    return labels, m[:, None], k[:, None]


# --------------------------------------------------------------------------------------
# multi-view fusion  (models/sac.py:238-269 `_avg_pool`, :218-236 `_minentropy_pool`)
# --------------------------------------------------------------------------------------
def avg_pool_views(aligned, T, T0=None, tolerance=0.1):
    """aligned [N*T,C,H,W] (all views of each group present) ->
    pooled [N*T0,C,H,W], mask [N*T0,1,H,W]."""
    NT, C, H, W = aligned.shape
    T0 = T if T0 is None else T0
    g = aligned.view(-1, T, C, H, W)
    S = g.sum(1)                                          # [N,C,H,W]
    Z = S.sum(1, keepdim=True)                            # [N,1,H,W]
    mask = (Z > tolerance).to(aligned.dtype)
    S = S / Z.clamp_min(1e-3)
    N = S.shape[0]
    pooled = S[:, None].expand(N, T0, C, H, W).reshape(N * T0, C, H, W)
    mask = mask[:, None].expand(N, T0, 1, H, W).reshape(N * T0, 1, H, W)
    return pooled, mask


def entropy_map(probs, eps=1e-5):
    """models/sac.py:189-196."""
    ent = -(probs * torch.log((probs + eps) / (1 + eps))).sum(1, keepdim=True)
    ent = torch.where(probs.sum(1, keepdim=True) < 0.1, torch.full_like(ent, 1.0 / eps), ent)
    return ent


def minentropy_pool_views(aligned, T, tolerance=0.1):
    """models/sac.py:218-236: every view of a group takes the probs of the group's
    lowest-entropy view (first minimum wins); mask from the pre-selection sum."""
    NT, C, H, W = aligned.shape
    g = aligned.view(-1, T, C, H, W)
    ent = entropy_map(aligned).view(-1, T, 1, H, W)
    pick = ent.argmin(1, keepdim=True).expand(-1, 1, C, H, W)
    mask = (g.sum(1, keepdim=True).sum(2, keepdim=True) > tolerance)
    sel = g.gather(1, pick).expand(-1, T, C, H, W)
    mask = mask.expand(-1, T, 1, H, W).to(aligned.dtype)
    return sel.reshape(NT, C, H, W), mask.reshape(NT, 1, H, W)


def refine(frames, teacher_logits, T, affine, affine_inv, ignore_mask, running_conf, *,
           beta, stat_momentum, training=True, pool=True, pool_kind="avg_pool", gather=None):
    """models/sac.py:271-313.  Returns (refined probs, new running_conf, diags).

    Note the reference's frame mix (quirk 4): the coverage of the *inverse* warp (view
    frame) multiplies the probs already warped into the *reference* frame (:299-305).

    `gather(tensor, T)`: the cross-rank `_gather` of :198-216 when a rank holds B < T views of
    a group (only `_avg_pool` calls it, :246; the pooled result is expanded to T0 = min(T, B)
    views, :244,264-265).  `_minentropy_pool` has no gather: with B < T its `view(-1,T,...)`
    (:222) raises, and so does this restatement.
    """
    B, _, H, W = frames.shape
    up = upsample_bilinear_ac(teacher_logits, H, W)
    probs = torch.softmax(up, 1)
    chi = running_conf
    if training:
        chi = update_running_conf(running_conf, probs, beta, stat_momentum)
    probs = probs * (1 - ignore_mask[:, None].to(probs.dtype))
    diags = {}
    if not pool:
        return probs, chi, diags
    aligned = warp_affine(probs, affine)
    diags["teacher_aligned"] = aligned
    diags["frames_aligned"] = warp_affine(frames, affine)
    cover = warp_coverage(affine_inv, H, W)
    if pool_kind == "avg_pool":
        views = aligned * cover
        if gather is not None:
            views = gather(views, T)
        pooled, mask = avg_pool_views(views, T, min(T, B))
    elif pool_kind == "minentropy_pool":
        if B % T:
            raise RuntimeError("shape '[-1, {}, 1, {}, {}]' is invalid for input of size {}".format(T, H, W, B * H * W))
        pooled, mask = minentropy_pool_views(aligned * cover, T)
    else:
        raise AssertionError("Pooling OP _{} not found".format(pool_kind))
    refined = warp_affine(pooled, affine_inv) * warp_affine(mask, affine_inv)
    return refined, chi, diags


# --------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------
def ce_per_pixel(logits, target, weight=None):
    """F.cross_entropy(..., ignore_index=255, reduction='none'): -w[y]*log_softmax(x)[y]."""
    B, C, H, W = logits.shape
    lsm = torch.log_softmax(logits, 1)
    valid = target != IGNORE
    safe = torch.where(valid, target, torch.zeros_like(target))
    nll = -torch.gather(lsm, 1, safe[:, None])[:, 0]
    if weight is not None:
        nll = nll * weight[safe]
    return torch.where(valid, nll, torch.zeros_like(nll))


def ce_mean_all_pixels(logits_up, target):
    """models/deeplabv2.py:223-224 -- mean over ALL pixels, ignored ones included (quirk 2)."""
    return ce_per_pixel(logits_up, target).mean().view(1)


def focal_ce(logits_up, pseudo, running_conf, p):
    """models/sac.py:119-132 (non-default LOSS='focal_ce')."""
    ce = ce_per_pixel(logits_up, pseudo, focal_weight(running_conf, p))
    return ce.mean(), _per_class(ce, pseudo, logits_up.shape[1])


def focal_ce_conf(logits_up, pseudo, teacher_conf, running_conf, p):
    """models/sac.py:134-149.  `ce[B,H,W] * conf[B,1,H,W]` broadcasts to [B,B,H,W] (quirk 1):
    loss = sum_hw (sum_i conf_i)(sum_j ce_j) / (B*B*H*W)."""
    B, C, H, W = logits_up.shape
    ce = ce_per_pixel(logits_up, pseudo, focal_weight(running_conf, p))
    loss = (teacher_conf[:, 0].sum(0) * ce.sum(0)).sum() / float(B * B * H * W)
    return loss, _per_class(ce, pseudo, C)


def _per_class(ce, pseudo, C):
    """models/sac.py:123-130 / 138-145: ignored pixels land in class 0 with ce = 0."""
    B, H, W = ce.shape
    idx = torch.where(pseudo == IGNORE, torch.zeros_like(pseudo), pseudo)
    acc = torch.zeros(B, C, H * W, dtype=ce.dtype)
    acc.scatter_(1, idx.view(B, 1, -1), ce.detach().view(B, 1, -1))
    return acc.mean(-1).mean(0)


# --------------------------------------------------------------------------------------
# momentum teacher  (models/sac.py:70-102)
# --------------------------------------------------------------------------------------
_EMA_SUFFIXES = ("weight", "bias", "running_mean", "running_var")


def momentum_update(slow, fast, momentum, update):
    """Sum over tensors of ||slow-fast||_2 (pre-update), optional in-place EMA on `slow`.
    Keys ending in num_batches_tracked are skipped (:89)."""
    total = torch.zeros(())
    for key, val in fast.items():
        if key.split(".")[-1] not in _EMA_SUFFIXES:
            continue
        total = total + torch.norm(slow[key] - val)
        if update:
            slow[key].mul_(momentum).add_(val * (1.0 - momentum))
    return total.view(1)


# --------------------------------------------------------------------------------------
# view affines  (datasets/dataloader_target.py:220-262), used to build synthetic inputs
# --------------------------------------------------------------------------------------
def view_affines(params, crop_h, crop_w):
    """params: list of (dy, dx, alpha_deg, scale, flip).  Returns theta, theta_inv [L,2,3]."""
    L = len(params)
    theta = torch.zeros(L, 2, 3)
    ar = float(crop_h) / float(crop_w)
    for i, (dy, dx, alpha, scale, flip) in enumerate(params):
        s, c = math.sin(alpha * math.pi / 180.0), math.cos(alpha * math.pi / 180.0)
        theta[i, 0, 0], theta[i, 0, 1] = flip * c, s * ar
        theta[i, 1, 0], theta[i, 1, 1] = -s / ar, c
        theta[i, 0, 2] = -1.0 * (c * dx + s * dy) / float(crop_w // 2)
        theta[i, 1, 2] = -1.0 * (-s * dx + c * dy) / float(crop_h // 2)
        theta[i] *= scale
    inv = theta.clone()
    inv[:, 0, 1] = theta[:, 1, 0] * ar ** 2
    inv[:, 1, 0] = theta[:, 0, 1] / ar ** 2
    inv[:, 0, 2] = -1 * (inv[:, 0, 0] * theta[:, 0, 2] + inv[:, 0, 1] * theta[:, 1, 2])
    inv[:, 1, 2] = -1 * (inv[:, 1, 0] * theta[:, 0, 2] + inv[:, 1, 1] * theta[:, 1, 2])
    inv /= torch.tensor([p[3] for p in params], dtype=torch.float32).view(-1, 1, 1) ** 2
    return theta, inv


def infer_labels(logits, out_h, out_w, lut=None):
    """/root/reference/infer_val.py:160-163 (`_, logits = model(image); masks_pred = F.softmax(logits, 1)`) followed by the
    result writer's `argmax` over classes and `convert_to_cs` (infer_val.py:60-65, train id -> label id).
    Returns (uint8 labels [B,H,W], winning probability [B,H,W], second-best gap [B,H,W])."""
    up = upsample_bilinear_ac(logits, out_h, out_w)
    probs = torch.softmax(up, dim=1)
    top2 = probs.topk(2, dim=1).values
    idx = probs.argmax(dim=1)
    lab = idx if lut is None else lut.long()[idx]
    return lab.to(torch.uint8), top2[:, 0], top2[:, 0] - top2[:, 1]
