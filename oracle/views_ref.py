"""Oracle (CPU, test infrastructure) for the K augmented views of one target crop -- the tail of
`DataTarget.__getitem__` (/root/reference/datasets/dataloader_target.py:281-306):

    GuidedRandHFlip          datasets/tf_target.py:141-157     per-view horizontal flip (python `random`)
    MaskRandScaleCrop        datasets/tf_target.py:159-239     per-view zoom window, crop/pad + resize back to the crop size
    ToTensorMask/Normalize/ApplyMask   tf_target.py:33-98      u8 -> fp32 /255, (x-mean)/std, padding mask -> image 0 / label -1
    _get_affine/_get_affine_inv        dataloader_target.py:220-262  (restated in head_ref.view_affines)

The pixel work of the reference is done by a third-party dependency, Pillow (present in this image: 12.2.0;
the reference pins no version): `Image.resize(size, BILINEAR)` = libImaging/Resample.c (separable triangle filter,
coefficients normalised in double, quantised to 22-bit fixed point, horizontal pass rounded to u8, then vertical pass)
and `Image.resize(size, NEAREST)` = libImaging/Geometry.c ImagingScaleAffine (source index = (int)(box0 + scale/2 +
k*scale) with the position accumulated by repeated double additions).  Both are restated here in numpy and pinned
against Pillow itself (tests/test_views_cpu.py) and against the reference's transform classes (golden g12).
Integer/byte work: the contract is BIT-EXACT.
"""
import math

import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2          # Resample.c: 8 bits for the result, 2 for headroom


def resample_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, support 1) filter over the
    full input range.  Returns (bounds int32 [out,2] = (first tap, tap count), coeffs int32 [out,ksize], ksize)."""
    scale = filterscale = float(in_size) / float(out_size)
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(ksize, dtype=np.float64)
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            a = -a if a < 0 else a
            w[x] = 1.0 - a if a < 1.0 else 0.0
        ww = float(w[:xmax].sum()) if xmax else 0.0
        # Resample.c accumulates ww in tap order; numpy's pairwise sum over <= 7 doubles is the same left-to-right sum
        ww = 0.0
        for x in range(xmax):
            ww += w[x]
        if ww != 0.0:
            w[:xmax] /= ww
        for x in range(xmax):
            v = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _clip8(acc):
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bilinear_u8(img, out_h, out_w):
    """Pillow `Image.resize((out_w, out_h), BILINEAR)` of a u8 image [H,W] or [H,W,C]: horizontal pass (u8 result),
    then vertical pass, both in 22-bit fixed point with round-half-up."""
    img = np.asarray(img, dtype=np.uint8)
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    H, W, C = img.shape
    bh, kh, _ = resample_coeffs(W, out_w)
    bv, kv, _ = resample_coeffs(H, out_h)
    tmp = np.zeros((H, out_w, C), dtype=np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_w):
        x0, n = bh[xx]
        acc = np.full((H, C), 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(n):
            acc += src[:, x0 + x, :] * int(kh[xx, x])
        tmp[:, xx, :] = _clip8(acc)
    out = np.zeros((out_h, out_w, C), dtype=np.uint8)
    src = tmp.astype(np.int64)
    for yy in range(out_h):
        y0, n = bv[yy]
        acc = np.full((out_w, C), 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for y in range(n):
            acc += src[y0 + y, :, :] * int(kv[yy, y])
        out[yy] = _clip8(acc)
    return out[:, :, 0] if squeeze else out


def nearest_index_table(in_size, out_size):
    """Geometry.c ImagingScaleAffine: source index per output position, -1 = outside.  The position is accumulated
    with repeated additions in double, exactly as the C loop does."""
    a = float(in_size) / float(out_size)
    pos = a * 0.5
    tab = np.zeros(out_size, dtype=np.int32)
    for x in range(out_size):
        xin = -1 if pos < 0.0 else int(pos)
        tab[x] = xin if 0 <= xin < in_size else -1
        pos += a
    return tab


def resize_nearest(img, out_h, out_w):
    """Pillow `Image.resize((out_w, out_h), NEAREST)` for [H,W] / [H,W,C] arrays (positions outside keep 0)."""
    img = np.asarray(img)
    H, W = img.shape[:2]
    ty, tx = nearest_index_table(H, out_h), nearest_index_table(W, out_w)
    out = img[np.clip(ty, 0, H - 1)][:, np.clip(tx, 0, W - 1)].copy()
    out[ty < 0] = 0
    out[:, tx < 0] = 0
    return out


# ------------------------------------------------------------------------------------------------
# parameter draws (python `random`, same call order as the reference) and the view pipeline
# ------------------------------------------------------------------------------------------------
def sample_view_params(rng, n_views, H, W, zoom_range, guided_hflip=True):
    """GuidedRandHFlip.__call__ (tf_target.py:141-157) then MaskRandScaleCrop.__call__/get_params (:159-239) for one
    group of `n_views` copies of an H x W crop.  `rng`: a `random.Random` (the reference uses the module-level one).
    Returns a list of dicts: flip (+1/-1), window (ii, jj, h, w) or None when the view keeps the crop, scale s, and the
    affine parameter row [dy, dx, alpha, 1/s, flip] that feeds `_get_affine`."""
    views = [dict(flip=1.0, window=None, s=1.0, affine=[0., 0., 0., 1., 1.]) for _ in range(n_views)]
    if guided_hflip:
        for v in views:
            if rng.random() > 0.5:
                v["flip"] = -1.0
                v["affine"][4] *= -1
    if zoom_range[1] - zoom_range[0] > 0:                 # dataloader_target.py:113-114
        i2, j2 = H / 2, W / 2
        for k, v in enumerate(views):
            if k == 0:
                continue
            s = rng.uniform(zoom_range[0], zoom_range[1])
            new_h, new_w = int(s * H), int(s * W)
            if s < 1.:
                ii, jj = rng.randint(0, H - new_h), rng.randint(0, W - new_w)
            else:
                ii, jj = rng.randint(H - new_h, 0), rng.randint(W - new_w, 0)
            if s == 1.:
                continue
            v["affine"][0] = ii + new_h / 2 - i2
            v["affine"][1] = jj + new_w / 2 - j2
            v["affine"][3] = 1 / s
            v["window"], v["s"] = (ii, jj, new_h, new_w), s
    return views


def _window(arr, ii, jj, h, w, fill):
    """F.crop (window inside the image, s < 1) or F.pad with `fill` (window around it, s > 1): tf_target.py:206-236."""
    H, W = arr.shape[:2]
    out = np.full((h, w) + arr.shape[2:], fill, dtype=arr.dtype)
    y0, x0, y1, x1 = max(ii, 0), max(jj, 0), min(ii + h, H), min(jj + w, W)
    out[y0 - ii:y1 - ii, x0 - jj:x1 - jj] = arr[y0:y1, x0:x1]
    return out


def make_views_u8(image, label, mask, views):
    """image u8 [H,W,3], label u8 [H,W], mask u8 [H,W] (0 = valid) -> per-view u8 arrays after flip + zoom window."""
    H, W = label.shape
    out = []
    for v in views:
        im, lb, mk = image, label, mask
        if v["flip"] < 0:
            im, lb, mk = im[:, ::-1], lb[:, ::-1], mk[:, ::-1]
        if v["window"] is not None:
            ii, jj, h, w = v["window"]
            im = resize_bilinear_u8(_window(im, ii, jj, h, w, 0), H, W)
            lb = resize_nearest(_window(lb, ii, jj, h, w, 1), H, W)
            mk = resize_nearest(_window(mk, ii, jj, h, w, 1), H, W)
        out.append((np.ascontiguousarray(im), np.ascontiguousarray(lb), np.ascontiguousarray(mk)))
    return out


def post_transform(views_u8, mean, std, ignore_label=-1):
    """ToTensorMask + Normalize + ApplyMask (tf_target.py:33-98): frames fp32 [L,3,H,W], labels int64 [L,H,W]."""
    frames, labels = [], []
    for im, lb, mk in views_u8:
        x = torch.from_numpy(im.transpose(2, 0, 1).copy()).to(torch.float32).div(255)
        for t, m, s in zip(x, mean, std):
            t.sub_(m).div_(s)
        m = torch.from_numpy(mk.astype(np.int32)) > 0.
        x *= (1. - m.type_as(x))
        y = torch.from_numpy(lb.astype(np.int32))
        y[m] = ignore_label
        frames.append(x)
        labels.append(y.long())
    return torch.stack(frames), torch.stack(labels)
