"""K-view generation on the device (da-sac_amd/views.py, dasac_make_views) against golden g12 -- the outputs of the
reference's own GuidedRandHFlip / MaskRandScaleCrop / ToTensorMask / Normalize / ApplyMask / _get_affine(_inv)
(tests/golden/make_goldens.py: g12_views) -- and against the oracle on fresh crops.  Byte work: bit-exact."""
import random

import numpy as np
import pytest
import torch

from oracle import views_ref as V

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def test_views_golden_g12_bit_exact(golden):
    import views
    g = golden("g12_views")
    for case in range(int(g["n_cases"])):
        t = "c%d_" % case
        img, lab, msk = g[t + "image"], g[t + "label"], g[t + "mask"]
        H, W = lab.shape
        tv = views.TargetViews((H, W), 4, zoom_range=g[t + "zoom"].tolist(), guided_hflip=True, seed=int(g[t + "seed"]))
        vs = tv.sample()
        assert np.array_equal(np.array([v["affine"] for v in vs]), g[t + "params"]), case      # the reference's draws
        f1, gt, f2, aff, inv, u8 = tv.make(T(np.ascontiguousarray(img.transpose(2, 0, 1))).cuda(), T(lab).cuda(), T(msk).cuda(),
                                           views=vs, want_u8=True)
        assert torch.equal(u8.cpu(), T(np.ascontiguousarray(g[t + "views_u8"].transpose(0, 3, 1, 2)))), case
        assert torch.equal(gt.cpu(), T(g[t + "gt"].astype(np.int64))), case
        if (t + "frames") in g.files:
            assert torch.equal(f1.cpu(), T(g[t + "frames"])), case
        assert f1 is f2 and gt.dtype == torch.int64 and f1.is_contiguous()
        assert torch.equal(aff.cpu(), T(g[t + "affine"])) and torch.equal(inv.cpu(), T(g[t + "affine_inv"])), case


@pytest.mark.parametrize("hw,zoom,seed", [((64, 96), (0.5, 1.0), 3), ((97, 131), (0.5, 1.2), 4), ((512, 1024), (0.5, 1.0), 5)])
def test_views_vs_oracle_fresh_crops(hw, zoom, seed):
    import views
    H, W = hw
    gen = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([(127 + 100 * np.sin(xx / (4.0 + c) + yy / 7.0) + gen.randint(-20, 21, (H, W))).clip(0, 255) for c in range(3)], -1).astype(np.uint8)
    lab = gen.randint(0, 19, ((H + 7) // 8, (W + 7) // 8)).repeat(8, 0).repeat(8, 1)[:H, :W].astype(np.uint8)
    msk = np.zeros((H, W), np.uint8)
    msk[:, :5] = 1
    msk[H - 4:] = 1
    tv = views.TargetViews(hw, 4, zoom_range=zoom, seed=seed)
    vs = tv.sample()
    ref_vs = V.sample_view_params(random.Random(seed), 4, H, W, zoom)
    assert [v["affine"] for v in vs] == [v["affine"] for v in ref_vs]
    ref_u8 = V.make_views_u8(img, lab, msk, ref_vs)
    ref_frames, ref_gt = V.post_transform(ref_u8, views.MEAN, views.STD)
    f1, gt, _, _, _, u8 = tv.make(T(np.ascontiguousarray(img.transpose(2, 0, 1))).cuda(), T(lab).cuda(), T(msk).cuda(), views=vs, want_u8=True)
    assert torch.equal(u8.cpu(), T(np.stack([o[0].transpose(2, 0, 1) for o in ref_u8])))
    assert torch.equal(gt.cpu(), ref_gt)
    assert torch.equal(f1.cpu(), ref_frames)
    # no mask given == an all-valid mask, padding of zoom-out windows included
    f3, gt3, _, _, _ = tv.make(T(np.ascontiguousarray(img.transpose(2, 0, 1))).cuda(), T(lab).cuda(), None, views=vs)
    r3, g3 = V.post_transform(V.make_views_u8(img, lab, np.zeros_like(msk), ref_vs), views.MEAN, views.STD)
    assert torch.equal(gt3.cpu(), g3) and torch.equal(f3.cpu(), r3)
