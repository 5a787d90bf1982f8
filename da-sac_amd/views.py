"""K augmented views of one target crop on the device -- the tail of the reference's
`DataTarget.__getitem__` (/root/reference/datasets/dataloader_target.py:281-306):

    GuidedRandHFlip        datasets/tf_target.py:141-157    per-view horizontal flip
    MaskRandScaleCrop      datasets/tf_target.py:159-239    per-view zoom window: crop (s < 1) or pad (s > 1) + resize back
    ToTensorMask / Normalize / ApplyMask   tf_target.py:33-98
    _get_affine / _get_affine_inv          dataloader_target.py:220-262   (driver.view_affines)

The reference runs this in DataLoader worker processes with Pillow, one PIL image per view (4 views x 3 resizes of a
512x1024 crop per target image) and ships fp32 frames through pinned memory; here the loader hands over ONE u8 crop
(+ label, padding mask) and a single launch (dasac_make_views) emits all L normalised views, labels with -1 padding
and nothing else crosses PCIe.  Parameter draws use python's `random` in the reference's call order, so a seeded
`random.Random` reproduces the reference's views exactly; the pixels are byte-exact with Pillow's fixed-point
resampling (tables built below in double precision, like Resample.c / Geometry.c).

The student's frames (`images1`) additionally go through `tf_augm` (dataloader_target.py:116-123,292-296):
    RandGaussianBlur     tf_target.py:331-349     PIL GaussianBlur(radius ~ U(.1, 2)) per view
    MaskRandJitter       tf_target.py:365-390     with probability p torchvision ColorJitter (4 adjustments, random order)
    MaskRandGreyscale    tf_target.py:351-363     with probability p greyscale
on the u8 views, before ToTensor / Normalize / ApplyMask: `dasac_view_photometric`, byte-exact with Pillow's BoxBlur.c /
Blend.c / Convert.c (the arithmetic torchvision's PIL backend delegates to).  The teacher's frames (`images2`) stay clean.
"""
import math
import random

import numpy as np
import torch

from dasac_hip import lib as L
import driver

MEAN = (0.485, 0.456, 0.406)          # datasets/dataloader_base.py:39-40
STD = (0.229, 0.224, 0.225)
_KS = 8                               # taps reserved per output position (csrc/views.hip: kViewKs)
_PREC = 22


def _bilinear_tables(in_size, out_size):
    """Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc (triangle filter): (first tap, count) and the 22-bit
    fixed-point coefficients of every output position, vectorised over positions, taps summed left to right."""
    scale = filterscale = float(in_size) / float(out_size)
    if filterscale < 1.0:
        filterscale = 1.0
    support = filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    if ksize > _KS:
        raise NotImplementedError("zoom window {}x the crop needs {} taps per pixel (> {})".format(scale, ksize, _KS))
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum(np.trunc(center - support + 0.5).astype(np.int64), 0)
    xmax = np.minimum(np.trunc(center + support + 0.5).astype(np.int64), in_size) - xmin
    taps = np.arange(_KS, dtype=np.float64)[None, :]
    a = np.abs((taps + xmin[:, None] - center[:, None] + 0.5) * (1.0 / filterscale))
    w = np.where((a < 1.0) & (taps < xmax[:, None]), 1.0 - a, 0.0)
    ww = np.zeros(out_size, dtype=np.float64)
    for x in range(_KS):
        ww = ww + w[:, x]
    w = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    kk = np.trunc(0.5 + w * float(1 << _PREC)).astype(np.int32)
    return np.stack([xmin, xmax], 1).astype(np.int32), kk


def _nearest_table(in_size, out_size):
    """Pillow Geometry.c ImagingScaleAffine: (int)(scale/2 + k*scale) with the position accumulated by repeated additions."""
    a = float(in_size) / float(out_size)
    pos = np.cumsum(np.concatenate([[a * 0.5], np.full(out_size - 1, a)]))       # sequential double additions
    idx = np.where(pos < 0.0, -1, np.trunc(pos)).astype(np.int64)
    return np.where((idx >= 0) & (idx < in_size), idx, -1).astype(np.int32)


def sample_views(rng, n_views, H, W, zoom_range=(0.5, 1.0), guided_hflip=True):
    """The random draws of GuidedRandHFlip.__call__ and MaskRandScaleCrop.get_params in the reference's order
    (tf_target.py:141-157,164-182,193-204).  Returns per view: flip (+1 / -1), window (ii, jj, h, w) or None, and the
    parameter row (dy, dx, alpha, 1/s, flip) consumed by `driver.view_affines`."""
    views = [dict(flip=1.0, window=None, affine=[0., 0., 0., 1., 1.]) for _ in range(n_views)]
    if guided_hflip:
        for v in views:
            if rng.random() > 0.5:
                v["flip"] = -1.0
                v["affine"][4] *= -1
    if zoom_range[1] - zoom_range[0] > 0:
        for k, v in enumerate(views):
            if k == 0:                              # the first copy stays un-zoomed (tf_target.py:195-196)
                continue
            s = rng.uniform(zoom_range[0], zoom_range[1])
            h, w = int(s * H), int(s * W)
            if s < 1.:
                ii, jj = rng.randint(0, H - h), rng.randint(0, W - w)
            else:
                ii, jj = rng.randint(H - h, 0), rng.randint(W - w, 0)
            if s == 1.:
                continue
            v["affine"][0], v["affine"][1], v["affine"][3] = ii + h / 2 - H / 2, jj + w / 2 - W / 2, 1 / s
            v["window"] = (ii, jj, h, w)
    return views


def sample_photometric(rng, torch_gen, n_views, blur=(.1, 2.), jitter=0.4, jitter_p=0.5, grey_p=0.2):
    """The draws of `tf_augm` in the reference's call order: RandGaussianBlur (tf_target.py:341-343), MaskRandJitter
    (:382-385) and MaskRandGreyscale (:358-360) each loop over the views.  `rng`: random.Random; `torch_gen`: the
    torch.Generator ColorJitter.get_params draws from (torchvision >= 0.8: randperm(4), then uniform brightness, contrast,
    saturation in [max(0, 1-j), 1+j] and hue in [-min(.1, j), min(.1, j)]).  Defaults = configs/deeplabv2_resnet101_train.yaml.
    Returns one dict per view: blur radius or None, jitter (order, factors) or None, grey flag."""
    views = [dict(blur=None, jitter=None, grey=False) for _ in range(n_views)]
    if blur is not None:
        for v in views:
            v["blur"] = rng.uniform(blur[0], blur[1])
    if jitter > 0:
        lo, hi, hue = max(0., 1. - jitter), 1. + jitter, min(0.1, jitter)
        for v in views:
            if rng.random() < jitter_p:
                order = torch.randperm(4, generator=torch_gen).tolist()
                fac = [float(torch.empty(1).uniform_(a, b, generator=torch_gen)) for a, b in ((lo, hi), (lo, hi), (lo, hi), (-hue, hue))]
                v["jitter"] = (order, fac)
    if grey_p > 0:
        for v in views:
            v["grey"] = grey_p > rng.random()
    return views


def photometric_params(views):
    """float64 [L, DASAC_PHOTO_PARAMS] rows for dasac_view_photometric (layout: include/dasac_hip.h)."""
    out = np.zeros((len(views), 12), dtype=np.float64)
    for r, v in enumerate(views):
        out[r, 0] = v["blur"] if v["blur"] is not None else 0.0
        if v["jitter"] is not None:
            out[r, 1] = 1.0
            out[r, 2:6] = v["jitter"][0]
            out[r, 6:10] = v["jitter"][1]
        out[r, 10] = 1.0 if v["grey"] else 0.0
    return out


def view_tables(views, H, W):
    """int32 [L, dasac_make_views_table_ints(H, W)] rows for dasac_make_views (layout: include/dasac_hip.h)."""
    stride = L.load().dasac_make_views_table_ints(H, W)
    out = np.zeros((len(views), stride), dtype=np.int32)
    for r, v in enumerate(views):
        row = out[r]
        row[0] = 1 if v["flip"] < 0 else 0
        if v["window"] is None:
            row[1:6] = (0, 0, H, W, 1)
            continue
        ii, jj, h, w = v["window"]
        row[1:6] = (ii, jj, h, w, 0)
        bh, kh = _bilinear_tables(w, W)
        bv, kv = _bilinear_tables(h, H)
        o = 8
        for part in (bh, kh, bv, kv, _nearest_table(w, W), _nearest_table(h, H)):
            row[o:o + part.size] = part.reshape(-1)
            o += part.size
        assert o == stride
    return out


class TargetViews:
    """Device-side view generator for one target crop.  `make` returns what the reference's loader yields for one
    image (dataloader_target.py:306): (frames1, gt, frames2, affine, affine_inv) with frames1 is frames2."""

    def __init__(self, crop_hw, group_size, zoom_range=(0.5, 1.0), guided_hflip=True, seed=None, mean=MEAN, std=STD,
                 blur=None, jitter=0.0, jitter_p=0.5, grey_p=0.0):
        """blur = (r0, r1) / jitter / grey_p switch the photometric augmentations of frames1 on (cfg.DATASET.RND_BLUR,
        RND_JITTER, RND_GREYSCALE; all off by default like core/config.py:78-81 minus the blur)."""
        self.H, self.W = int(crop_hw[0]), int(crop_hw[1])
        self.L, self.zoom, self.guided_hflip = int(group_size), tuple(zoom_range), bool(guided_hflip)
        self.rng = random.Random(seed)
        self.torch_gen = torch.Generator()
        # seed=None: both generators start from OS entropy (the reference draws ColorJitter from the global torch RNG, which
        # DataLoader seeds per worker) -- a bare torch.Generator() would start every rank and every run on torch's fixed default
        self.torch_gen.manual_seed(seed if seed is not None else self.rng.getrandbits(63))
        self.mean = np.asarray(mean, dtype=np.float32)
        self.std = np.asarray(std, dtype=np.float32)
        self.blur, self.jitter, self.jitter_p, self.grey_p = blur, float(jitter), float(jitter_p), float(grey_p)

    @property
    def photometric(self):
        return self.blur is not None or self.jitter > 0 or self.grey_p > 0

    def sample(self):
        return sample_views(self.rng, self.L, self.H, self.W, self.zoom, self.guided_hflip)

    def sample_photometric(self):
        return sample_photometric(self.rng, self.torch_gen, self.L, self.blur, self.jitter, self.jitter_p, self.grey_p)

    def augment(self, views_u8, gt, photo, want_u8=False):
        """`tf_augm` + post transforms on the u8 views of `make(..., want_u8=True)`: frames1 f32 [L,3,H,W] (+ the bytes)."""
        L.require_gpu(views_u8, gt)
        lib = L.load()
        nv, H, W = views_u8.shape[0], self.H, self.W
        assert tuple(views_u8.shape) == (nv, 3, H, W) and views_u8.dtype == torch.uint8 and views_u8.is_contiguous()
        assert gt is None or (tuple(gt.shape) == (nv, H, W) and gt.dtype == torch.int64 and gt.is_contiguous())
        dev = views_u8.device
        params = np.ascontiguousarray(photometric_params(photo))
        frames = torch.empty((nv, 3, H, W), dtype=torch.float32, device=dev)
        u8 = torch.empty((nv, 3, H, W), dtype=torch.uint8, device=dev) if want_u8 else None
        nbytes = lib.dasac_view_photometric_workspace(H, W, nv)
        ws = L.workspace(nbytes, dev)
        L.check(lib.dasac_view_photometric(views_u8.data_ptr(), L.ptr(gt), H, W, nv, params.ctypes.data, self.mean.ctypes.data,
                                           self.std.ctypes.data, -1, frames.data_ptr(), L.ptr(u8), ws.data_ptr(), nbytes,
                                           L.stream_ptr()), "dasac_view_photometric")
        return (frames, u8) if want_u8 else frames

    def make(self, image_u8, label_u8, mask_u8=None, views=None, want_u8=False, photo=None):
        """image_u8 [3,H,W] uint8 cuda (planar), label_u8 [H,W] uint8, mask_u8 [H,W] uint8 or None (non-zero = padding).
        With photometric augmentations configured (or `photo` given) frames1 is the augmented student input, frames2 the
        clean teacher input, like `images1` / `images2` of dataloader_target.py:292-306."""
        L.require_gpu(image_u8, label_u8, mask_u8)
        lib = L.load()
        views = self.sample() if views is None else views
        H, W, nv = self.H, self.W, len(views)
        assert tuple(image_u8.shape) == (3, H, W) and image_u8.dtype == torch.uint8 and image_u8.is_contiguous()
        assert tuple(label_u8.shape) == (H, W) and label_u8.dtype == torch.uint8 and label_u8.is_contiguous()
        assert mask_u8 is None or (tuple(mask_u8.shape) == (H, W) and mask_u8.dtype == torch.uint8 and mask_u8.is_contiguous())
        dev = image_u8.device
        tables = torch.from_numpy(view_tables(views, H, W)).pin_memory().to(dev, non_blocking=True)
        frames = torch.empty((nv, 3, H, W), dtype=torch.float32, device=dev)
        gt = torch.empty((nv, H, W), dtype=torch.int64, device=dev)
        if photo is None and self.photometric:
            photo = self.sample_photometric()
        u8 = torch.empty((nv, 3, H, W), dtype=torch.uint8, device=dev) if (want_u8 or photo is not None) else None
        L.check(lib.dasac_make_views(image_u8.data_ptr(), label_u8.data_ptr(), L.ptr(mask_u8), H, W, nv, tables.data_ptr(),
                                     self.mean.ctypes.data, self.std.ctypes.data, -1, frames.data_ptr(), gt.data_ptr(), L.ptr(u8),
                                     L.stream_ptr()), "dasac_make_views")
        theta, theta_inv = driver.view_affines([tuple(v["affine"]) for v in views], H, W)
        theta, theta_inv = theta.to(dev, non_blocking=True), theta_inv.to(dev, non_blocking=True)
        frames1 = frames if photo is None else self.augment(u8, gt, photo)
        out = (frames1, gt, frames, theta, theta_inv)
        return out + (u8,) if want_u8 else out
