"""Oracle of the K-view generation (oracle/views_ref.py) against (a) Pillow itself -- the third-party library that does
the reference's pixel work -- and (b) golden g12, captured from the reference's own transform classes
(tests/golden/make_goldens.py: g12_views).  Everything here is integer/byte work: bit-exact."""
import random

import numpy as np
import pytest
import torch

from oracle import head_ref as H
from oracle import views_ref as V


def test_pillow_resize_restatement_is_bit_exact():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.RandomState(0)
    for (h, w, oh, ow) in [(33, 57, 64, 96), (64, 96, 33, 57), (48, 48, 48, 48), (100, 140, 64, 96), (64, 96, 27, 31), (32, 48, 64, 96)]:
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        lab = rng.randint(0, 20, (h, w)).astype(np.uint8)
        assert np.array_equal(np.array(Image.fromarray(img).resize((ow, oh), Image.BILINEAR)), V.resize_bilinear_u8(img, oh, ow))
        assert np.array_equal(np.array(Image.fromarray(lab, "L").resize((ow, oh), Image.NEAREST)), V.resize_nearest(lab, oh, ow))


def test_views_match_reference_golden_g12(golden):
    g = golden("g12_views")
    mean, std = g["mean"].tolist(), g["std"].tolist()
    for case in range(int(g["n_cases"])):
        t = "c%d_" % case
        img, lab, msk = g[t + "image"], g[t + "label"], g[t + "mask"]
        Hh, Ww = lab.shape
        views = V.sample_view_params(random.Random(int(g[t + "seed"])), 4, Hh, Ww, g[t + "zoom"].tolist(), guided_hflip=True)
        assert np.array_equal(np.array([v["affine"] for v in views]), g[t + "params"]), case          # same draws, same arithmetic
        out = V.make_views_u8(img, lab, msk, views)
        assert np.array_equal(np.stack([o[0] for o in out]), g[t + "views_u8"]), case
        valid = g[t + "masks_u8"] == 0
        assert np.array_equal(np.stack([o[2] for o in out]), g[t + "masks_u8"]), case
        assert np.array_equal(np.stack([o[1] for o in out])[valid], g[t + "labels_u8"][valid]), case
        frames, gts = V.post_transform(out, mean, std)
        assert torch.equal(gts, torch.from_numpy(g[t + "gt"].astype(np.int64))), case
        if (t + "frames") in g.files:
            assert torch.equal(frames, torch.from_numpy(g[t + "frames"])), case
        aff, inv = H.view_affines([tuple(v["affine"]) for v in views], Hh, Ww)
        assert torch.equal(aff, torch.from_numpy(g[t + "affine"])) and torch.equal(inv, torch.from_numpy(g[t + "affine_inv"])), case
    assert any(g["c%d_params" % c][:, 3].min() < 1.0 for c in range(3))      # a zoom-out (pad) view is covered


def test_product_host_tables_equal_the_oracle_tables():
    """da-sac_amd/views.py builds the resampling tables vectorised; they must equal the tap-by-tap restatement."""
    import views
    for (i, o) in [(33, 64), (57, 96), (64, 64), (140, 96), (250, 96), (511, 512), (1229, 1024), (512, 1024), (47, 96), (95, 96), (97, 96)]:
        b, k, ks = V.resample_coeffs(i, o)
        b2, k2 = views._bilinear_tables(i, o)
        assert np.array_equal(b, b2) and np.array_equal(k, k2[:, :ks]) and not k2[:, ks:].any(), (i, o)
        assert np.array_equal(V.nearest_index_table(i, o), views._nearest_table(i, o)), (i, o)
    with pytest.raises(NotImplementedError):
        views._bilinear_tables(300, 96)
    vs = views.sample_views(random.Random(3), 4, 64, 96, (0.5, 1.0))
    assert [v["affine"] for v in vs] == [v["affine"] for v in V.sample_view_params(random.Random(3), 4, 64, 96, (0.5, 1.0))]
