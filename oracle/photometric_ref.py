"""Oracle (CPU, test infrastructure) for the photometric augmentations of the student's views -- `tf_augm` of
`DataTarget` (/root/reference/datasets/dataloader_target.py:116-123,292-296), applied to `images1` only:

    RandGaussianBlur     datasets/tf_target.py:331-349   radius = random.uniform(.1, 2.) per view, PIL GaussianBlur
    MaskRandJitter       datasets/tf_target.py:365-390   with probability p: torchvision ColorJitter(b, c, s, hue=min(.1, j))
    MaskRandGreyscale    datasets/tf_target.py:351-363   with probability p: F.to_grayscale(img, 3)

The pixel arithmetic lives in third-party dependencies, restated here in numpy and pinned against the real thing:
  * Pillow (present in this image: 12.2.0; the reference pins no version) -- libImaging BoxBlur.c (GaussianBlur = three
    passes of an "extended box filter" per axis, 24-bit fixed point, every pass rounded to u8), Blend.c (ImageEnhance =
    blend(degenerate, image, factor) in float32, truncating), Convert.c (RGB -> L = (19595 R + 38470 G + 7471 B + 2^15) >> 16,
    RGB <-> HSV).  tests/test_photometric_cpu.py checks every function below against Pillow itself, the colour conversions
    exhaustively over all 2^24 colours.
  * torchvision (ABSENT from this image and unpinned by the reference): ColorJitter.forward / functional_pil -- the order of
    the four adjustments is a random permutation, brightness / contrast / saturation = PIL.ImageEnhance.{Brightness,Contrast,
    Color}(img).enhance(f), hue = shift of the H channel of img.convert("HSV") by uint8(f * 255) with wrap-around.  Restated
    from its published behaviour (torchvision >= 0.8: `get_params` draws permutation and factors from torch's RNG).
Byte work: the contract is BIT-EXACT.
"""
import math

import numpy as np

# ------------------------------------------------------------------------------------------------
# Gaussian blur (BoxBlur.c)
# ------------------------------------------------------------------------------------------------
PASSES = 3


def box_radius(radius, passes=PASSES):
    """BoxBlur.c _gaussian_blur_radius: the extended-box radius whose `passes`-fold convolution has standard deviation
    `radius` (float arithmetic with double intermediates, as the C expression evaluates)."""
    f = np.float32
    radius = f(radius)
    sigma2 = f(radius * radius / f(passes))
    L = f(math.sqrt(12.0 * float(sigma2) + 1.0))
    l = f(math.floor((float(L) - 1.0) / 2.0))
    a = f(f(f(2) * l + f(1)) * f(f(l * f(l + f(1))) - f(f(3) * sigma2)))
    a = f(a / f(f(6) * f(sigma2 - f(f(l + f(1)) * f(l + f(1))))))
    return f(l + a)


def box_weights(float_radius):
    """ImagingHorizontalBoxBlur: integer radius, weight of the 2r+1 inner pixels `ww` and of the two outer ones `fw`,
    both in 1/2^24."""
    fr = np.float32(float_radius)
    radius = int(fr)
    ww = int(np.float32(1 << 24) / np.float32(fr * np.float32(2) + np.float32(1)))
    fw = ((1 << 24) - (radius * 2 + 1) * ww) // 2
    return radius, ww, fw


def box_pass(img, radius, ww, fw, axis):
    """One extended-box pass along `axis` of a u8 array with edge replication; the result is rounded to u8."""
    a = np.moveaxis(np.asarray(img, dtype=np.uint8), axis, -1).astype(np.int64)
    n = a.shape[-1]
    idx = np.arange(n)
    acc = np.zeros_like(a)
    for d in range(-radius, radius + 1):
        acc += a[..., np.clip(idx + d, 0, n - 1)]
    far = a[..., np.clip(idx - radius - 1, 0, n - 1)] + a[..., np.clip(idx + radius + 1, 0, n - 1)]
    out = (((acc * ww + far * fw) & 0xFFFFFFFF) + (1 << 23) & 0xFFFFFFFF) >> 24
    return np.moveaxis(out.astype(np.uint8), -1, axis)


def gaussian_blur_u8(img, radius):
    """PIL `img.filter(ImageFilter.GaussianBlur(radius))` for an [H,W,C] / [H,W] u8 array: three horizontal passes, then
    three vertical ones (BoxBlur.c ImagingBoxBlur; the vertical passes run on the transposed image)."""
    img = np.asarray(img, dtype=np.uint8)
    r, ww, fw = box_weights(box_radius(radius))
    for _ in range(PASSES):
        img = box_pass(img, r, ww, fw, 1)
    for _ in range(PASSES):
        img = box_pass(img, r, ww, fw, 0)
    return img


# ------------------------------------------------------------------------------------------------
# colour conversions (Convert.c) and blend (Blend.c)
# ------------------------------------------------------------------------------------------------
def rgb_to_l(rgb):
    """`Image.convert("L")` of RGB: ITU-R 601-2 luma in 16-bit fixed point."""
    c = np.asarray(rgb, dtype=np.uint8).astype(np.int64)
    return ((c[..., 0] * 19595 + c[..., 1] * 38470 + c[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend_u8(a, b, alpha):
    """Blend.c ImagingBlend(a, b, alpha): a + alpha * (b - a) in float32, truncated; clipped when alpha is outside [0, 1]."""
    alpha = np.float32(alpha)
    a = np.asarray(a, dtype=np.uint8)
    b = np.asarray(b, dtype=np.uint8)
    if alpha == 0.0:
        return np.broadcast_to(a, np.broadcast(a, b).shape).copy()
    if alpha == 1.0:
        return np.broadcast_to(b, np.broadcast(a, b).shape).copy()
    ai = a.astype(np.int32)
    t = ai.astype(np.float32) + alpha * (b.astype(np.int32) - ai).astype(np.float32)
    if 0.0 <= alpha <= 1.0:
        return t.astype(np.int32).astype(np.uint8)
    return np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t.astype(np.int32))).astype(np.uint8)


def rgb_to_hsv(rgb):
    """Convert.c rgb2hsv_row (follows colorsys.py, float32 arithmetic with double constants, truncating)."""
    c = np.asarray(rgb, dtype=np.uint8)
    r, g, b = (c[..., k].astype(np.int32) for k in range(3))
    maxc = np.maximum(r, np.maximum(g, b))
    minc = np.minimum(r, np.minimum(g, b))
    f = np.float32
    cr = (maxc - minc).astype(f)
    safe = np.where(cr == 0, f(1), cr)
    s = cr / np.where(maxc == 0, 1, maxc).astype(f)
    rc = (maxc - r).astype(f) / safe
    gc = (maxc - g).astype(f) / safe
    bc = (maxc - b).astype(f) / safe
    h = np.where(r == maxc, bc - gc, np.where(g == maxc, (2.0 + rc.astype(np.float64) - bc).astype(f),
                                              (4.0 + gc.astype(np.float64) - rc).astype(f)))
    h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(f)
    uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    grey = minc == maxc
    return np.stack([np.where(grey, 0, uh), np.where(grey, 0, us), maxc], -1).astype(np.uint8)


def hsv_to_rgb(hsv):
    """Convert.c hsv2rgb_row."""
    c = np.asarray(hsv, dtype=np.uint8)
    h, s, v = (c[..., k].astype(np.int32) for k in range(3))
    f32 = np.float32
    hf = h.astype(f32)
    i = np.floor(hf.astype(np.float64) * 6.0 / 255.0).astype(np.int32)
    f = (hf.astype(np.float64) * 6.0 / 255.0 - i).astype(f32)
    fs = (s.astype(f32).astype(np.float64) / 255.0).astype(f32)
    vf = v.astype(f32).astype(np.float64)
    p = np.round(vf * (1.0 - fs.astype(np.float64)))
    q = np.round(vf * (1.0 - fs.astype(np.float64) * f.astype(np.float64)))
    t = np.round(vf * (1.0 - fs.astype(np.float64) * (1.0 - f.astype(np.float64))))
    p, q, t = (np.clip(x, 0, 255).astype(np.int32) for x in (p, q, t))
    k = i % 6
    r = np.choose(k, [v, q, p, p, t, v])
    g = np.choose(k, [t, v, v, q, p, p])
    b = np.choose(k, [p, p, t, v, v, q])
    grey = s == 0
    return np.stack([np.where(grey, v, r), np.where(grey, v, g), np.where(grey, v, b)], -1).astype(np.uint8)


# ------------------------------------------------------------------------------------------------
# ImageEnhance / torchvision functional_pil adjustments on an [H,W,3] u8 image
# ------------------------------------------------------------------------------------------------
def adjust_brightness(img, factor):
    return blend_u8(np.zeros(1, dtype=np.uint8), img, factor)


def contrast_mean(img):
    """int(ImageStat.Stat(img.convert("L")).mean[0] + 0.5): python float sum / count."""
    grey = rgb_to_l(img)
    return int(float(grey.astype(np.int64).sum()) / grey.size + 0.5)


def adjust_contrast(img, factor):
    return blend_u8(np.full(1, contrast_mean(img), dtype=np.uint8), img, factor)


def adjust_saturation(img, factor):
    return blend_u8(rgb_to_l(img)[..., None], img, factor)


def hue_shift_byte(factor):
    """torchvision functional_pil.adjust_hue: np.uint8(hue_factor * 255) -- truncation toward zero, modulo 256."""
    return int(factor * 255) & 0xFF


def adjust_hue(img, factor):
    hsv = rgb_to_hsv(img)
    hsv[..., 0] = (hsv[..., 0].astype(np.int32) + hue_shift_byte(factor)) & 0xFF
    return hsv_to_rgb(hsv)


def to_greyscale3(img):
    return np.repeat(rgb_to_l(img)[..., None], 3, axis=-1)


ADJUST = (adjust_brightness, adjust_contrast, adjust_saturation, adjust_hue)


def color_jitter(img, order, factors):
    """torchvision ColorJitter.forward: `order` = permutation of (0 brightness, 1 contrast, 2 saturation, 3 hue),
    `factors` = the four factors (None = that adjustment is disabled)."""
    for k in order:
        if factors[k] is not None:
            img = ADJUST[k](img, factors[k])
    return img


# ------------------------------------------------------------------------------------------------
# parameter draws, in the reference's call order
# ------------------------------------------------------------------------------------------------
def sample_photometric(rng, torch_gen, n_views, blur=(.1, 2.), jitter=0.4, jitter_p=0.5, grey_p=0.2):
    """tf_augm's draws for one group of views: RandGaussianBlur, then MaskRandJitter, then MaskRandGreyscale, each looping
    over the views (tf_target.py:341-343,382-385,358-360).  `rng`: random.Random (the reference uses the module-level
    generator); `torch_gen`: torch.Generator for ColorJitter.get_params (torchvision >= 0.8 draws from torch's RNG:
    randperm(4), then uniform brightness, contrast, saturation, hue)."""
    import torch
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


def photometric_u8(img, view):
    """One view's `tf_augm`: [H,W,3] u8 -> [H,W,3] u8."""
    if view["blur"] is not None:
        img = gaussian_blur_u8(img, view["blur"])
    if view["jitter"] is not None:
        img = color_jitter(img, *view["jitter"])
    if view["grey"]:
        img = to_greyscale3(img)
    return img
