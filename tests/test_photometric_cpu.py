"""Oracle of the photometric view augmentations (oracle/photometric_ref.py) against (a) Pillow itself -- the library whose
C code does the reference's pixel work (BoxBlur.c, Blend.c, Convert.c), the colour conversions exhaustively over all 2^24
colours -- and (b) golden g13, captured from the reference's own RandGaussianBlur / MaskRandJitter / MaskRandGreyscale
(tests/golden/make_goldens.py: g13_photometric).  Byte work: bit-exact."""
import random

import numpy as np
import pytest
import torch

from oracle import photometric_ref as P


def test_colour_conversions_equal_pillow_on_every_colour():
    Image = pytest.importorskip("PIL.Image")
    allc = np.arange(1 << 24, dtype=np.uint32)
    rgb = np.stack([(allc >> 16) & 255, (allc >> 8) & 255, allc & 255], -1).astype(np.uint8).reshape(4096, 4096, 3)
    im = Image.fromarray(rgb, "RGB")
    assert np.array_equal(np.asarray(im.convert("L")), P.rgb_to_l(rgb))
    assert np.array_equal(np.asarray(im.convert("HSV")), P.rgb_to_hsv(rgb))
    assert np.array_equal(np.asarray(Image.fromarray(rgb, "HSV").convert("RGB")), P.hsv_to_rgb(rgb))


def test_blend_enhance_hue_equal_pillow():
    Image = pytest.importorskip("PIL.Image")
    from PIL import ImageEnhance
    rs = np.random.RandomState(0)
    a, b = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    ia, ib = Image.fromarray(a, "L"), Image.fromarray(b, "L")
    for alpha in list(rs.uniform(-0.5, 2.0, 40)) + [0.0, 1.0, 0.5, 0.6, 1.4, 1.0000001, 0.3333333]:
        assert np.array_equal(np.asarray(Image.blend(ia, ib, float(alpha))), P.blend_u8(a, b, alpha)), alpha
    img = rs.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    img[:10, :10] = 200
    img[20:, 30:] //= 4
    pim = Image.fromarray(img, "RGB")
    for f in rs.uniform(0.5, 1.5, 12).tolist() + [0.0, 1.0]:
        assert np.array_equal(np.asarray(ImageEnhance.Brightness(pim).enhance(f)), P.adjust_brightness(img, f)), f
        assert np.array_equal(np.asarray(ImageEnhance.Contrast(pim).enhance(f)), P.adjust_contrast(img, f)), f
        assert np.array_equal(np.asarray(ImageEnhance.Color(pim).enhance(f)), P.adjust_saturation(img, f)), f
    for f in rs.uniform(-0.1, 0.1, 12).tolist() + [0.0, -0.5, 0.5]:
        # torchvision functional_pil.adjust_hue, statement by statement
        h, s, v = pim.convert("HSV").split()
        np_h = np.array(h, dtype=np.uint8)
        with np.errstate(over="ignore"):
            np_h += np.int32(f * 255).astype(np.uint8)
        ref = Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB")
        assert np.array_equal(np.asarray(ref), P.adjust_hue(img, f)), f
    assert np.array_equal(np.asarray(pim.convert("L")), P.to_greyscale3(img)[..., 1])


def test_gaussian_blur_equals_pillow():
    Image = pytest.importorskip("PIL.Image")
    from PIL import ImageFilter
    rs = np.random.RandomState(1)
    for trial in range(48):
        H, W = rs.randint(1, 40), rs.randint(1, 60)           # images narrower than the box radius included
        im = rs.randint(0, 256, (H, W, 3)).astype(np.uint8)
        r = float(rs.uniform(0.1, 2.0)) if trial < 36 else float(rs.uniform(2.0, 9.0))
        assert np.array_equal(np.asarray(Image.fromarray(im, "RGB").filter(ImageFilter.GaussianBlur(r))), P.gaussian_blur_u8(im, r)), (H, W, r)
    imL = rs.randint(0, 256, (23, 31)).astype(np.uint8)
    assert np.array_equal(np.asarray(Image.fromarray(imL, "L").filter(ImageFilter.GaussianBlur(1.3))), P.gaussian_blur_u8(imL, 1.3))


def _golden_views(g, case):
    t = "c%d_" % case
    L = len(g[t + "radii"])
    views, k = [], 0
    for v in range(L):
        jit = None
        if g[t + "jitter_on"][v]:
            jit = (g[t + "jitter_order"][k].tolist(), g[t + "jitter_factors"][k].tolist())
            k += 1
        views.append(dict(blur=float(g[t + "radii"][v]), jitter=jit, grey=bool(g[t + "grey_on"][v])))
    return views


def test_photometric_oracle_matches_reference_golden_g13(golden):
    g = golden("g13_photometric")
    seen = set()
    for case in range(int(g["n_cases"])):
        t = "c%d_" % case
        views = _golden_views(g, case)
        for v, view in enumerate(views):
            assert np.array_equal(P.photometric_u8(g[t + "image"], view), g[t + "out_u8"][v]), (case, v)
            seen.add((view["jitter"] is not None, view["grey"]))
        # the samplers (oracle and product) reproduce the reference's draws from the seeds
        seed, jitter, grey_p = int(g[t + "seed"]), float(g[t + "jitter"]), float(g[t + "grey_p"])
        import views as product
        for fn in (P.sample_photometric, product.sample_photometric):
            drawn = fn(random.Random(seed), torch.Generator().manual_seed(seed), len(views), (.1, 2.), jitter, 0.5, grey_p if grey_p > 0 else 1e-300)
            assert [d["blur"] for d in drawn] == [v["blur"] for v in views], case
            assert [d["jitter"] for d in drawn] == [v["jitter"] for v in views], case
            assert [d["grey"] for d in drawn] == [v["grey"] for v in views], case
    assert seen == {(False, False), (False, True), (True, False), (True, True)}


def test_product_parameter_rows():
    import views as product
    vs = [dict(blur=1.25, jitter=([2, 0, 3, 1], [0.9, 1.1, 1.3, -0.05]), grey=True), dict(blur=None, jitter=None, grey=False)]
    rows = product.photometric_params(vs)
    assert rows.shape == (2, 12) and rows.dtype == np.float64
    assert rows[0].tolist() == [1.25, 1.0, 2, 0, 3, 1, 0.9, 1.1, 1.3, -0.05, 1.0, 0.0] and not rows[1].any()


def test_color_jitter_oracle_equals_pillow_for_all_24_adjustment_orders():
    """torchvision's ColorJitter applies brightness / contrast / saturation / hue in a randomly permuted order
    (/root/reference/datasets/tf_target.py:365-390).  torchvision is absent here, so WHICH permutation its RNG stream yields for a
    seed is not pinned -- instead every one of the 24 orders is: the oracle's chain equals the real Pillow chain for each."""
    import itertools
    Image = pytest.importorskip("PIL.Image")
    from PIL import ImageEnhance
    rs = np.random.RandomState(7)
    img = rs.randint(0, 256, (29, 41, 3)).astype(np.uint8)
    img[:8, :12] = (250, 3, 128)

    def pillow_adjust(pim, which, f):
        if which == 0:
            return ImageEnhance.Brightness(pim).enhance(f)
        if which == 1:
            return ImageEnhance.Contrast(pim).enhance(f)
        if which == 2:
            return ImageEnhance.Color(pim).enhance(f)
        h, s, v = pim.convert("HSV").split()
        np_h = np.array(h, dtype=np.uint8)
        with np.errstate(over="ignore"):
            np_h += np.int32(f * 255).astype(np.uint8)
        return Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB")

    orders = list(itertools.permutations(range(4)))
    assert len(orders) == 24
    for order in orders:
        factors = [float(rs.uniform(0.6, 1.4)), float(rs.uniform(0.6, 1.4)), float(rs.uniform(0.6, 1.4)), float(rs.uniform(-0.1, 0.1))]
        pim = Image.fromarray(img, "RGB")
        for which in order:
            pim = pillow_adjust(pim, which, factors[which])
        assert np.array_equal(np.asarray(pim), P.color_jitter(img, list(order), factors)), order
