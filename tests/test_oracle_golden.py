"""CPU: the oracle (oracle/aug_oracle.{c,py}) against the golden vectors minted from the
reference's own functions (tests/golden/make_golden.py).  This is what pins the checker."""
import numpy as np
import pytest

import inputs as gi
from oracle import aug_oracle as orc

B = 2
CONV_ATOL = 1e-5 * 255.0  # 1e-5 of the 0..255 output scale (north_star: float ops within 1e-5)


@pytest.fixture(scope="module")
def x():
    return gi.obs_u8(11, B).astype(np.float32)


def test_shift_bit_exact(golden_aug, x):
    want = golden_aug["shift_out_u8"].astype(np.float32)
    oy, ox = golden_aug["shift_oy"], golden_aug["shift_ox"]
    assert np.array_equal(orc.random_shift_c(x, oy, ox), want)
    assert np.array_equal(orc.random_shift_np(x, oy, ox), want)
    # u8 variants: same index arithmetic
    xu = gi.obs_u8(11, B)
    assert np.array_equal(orc.random_shift_c(xu, oy, ox), golden_aug["shift_out_u8"])
    assert np.array_equal(orc.random_shift_c(xu, oy, ox, out_dtype=np.float32), want)


def test_conv_within_tolerance(golden_aug, x):
    w = golden_aug["conv_weights"]
    for got in (orc.random_conv_c(x, w), orc.random_conv_np(x, w)):
        err = np.abs(got - golden_aug["conv_out"]).max()
        assert err <= CONV_ATOL, err
    # observed: ~1e-4 absolute (4e-7 of scale): summation-order noise only
    assert np.abs(orc.random_conv_c(x, w) - golden_aug["conv_out"]).max() < 5e-4


def test_overlay_bit_exact(golden_aug, x):
    ov = gi.overlay_imgs(12, B)
    assert np.array_equal(orc.random_overlay_c(x, ov), golden_aug["overlay_out"])
    assert np.array_equal(orc.random_overlay_np(x, ov), golden_aug["overlay_out"])


def test_jitter_compat_matches_callsite_golden(golden_aug, x):
    # parity unpinned for the arithmetic (torchvision absent); this pins the reference's CALL SITE
    # (index range, in-place scatter, one parameter set) and C-vs-torch restatement agreement
    p = golden_aug["jitter_params"]
    got = orc.color_jitter_c(x, golden_aug["jitter_idx"], golden_aug["jitter_order"], p[0], p[1], p[2])
    assert np.array_equal(got, golden_aug["jitter_out"])
    untouched = sorted(set(range(B * 3)) - set(golden_aug["jitter_idx"].tolist()))
    v = got.reshape(B * 3, 3, 84, 84)
    assert np.array_equal(v[untouched], x.reshape(B * 3, 3, 84, 84)[untouched])


def test_jitter_unit_range(golden_aug):
    x01 = gi.smooth_obs_u8(13, B).astype(np.float32) / np.float32(255.0)
    p = golden_aug["jitter01_params"]
    got = orc.color_jitter_c(x01, golden_aug["jitter01_idx"], golden_aug["jitter01_order"], p[0], p[1], p[2])
    assert np.abs(got - golden_aug["jitter01_out"]).max() <= 1e-5


def test_jitter_sane_mode_is_scaled_compat():
    x = gi.smooth_obs_u8(5, 1).astype(np.float32)
    order, c, s, h = gi.jitter_params(3)
    idx = np.arange(3, dtype=np.int32)
    a = orc.color_jitter_c(x, idx, order, c, s, h, scale=255.0)
    b = orc.color_jitter_c(x / np.float32(255.0), idx, order, c, s, h) * np.float32(255.0)
    assert np.array_equal(a, b)


def test_edge_cases():
    # B = 0, extreme offsets, single-frame observations
    z = np.zeros((0, 9, 84, 84), np.float32)
    assert orc.random_shift_c(z, np.zeros(0, np.int32), np.zeros(0, np.int32)).shape == z.shape
    x = gi.obs_u8(3, 1, C=3).astype(np.float32)
    for oy, ox in [(0, 0), (8, 8), (0, 8), (4, 4)]:
        got = orc.random_shift_c(x, [oy], [ox])
        assert np.array_equal(got, orc.random_shift_np(x, [oy], [ox]))
    assert np.array_equal(orc.random_shift_c(x, [4], [4]), x)  # centre crop = identity


# ---------------------------------------------------------------- rest of the registry (aug_extra_golden.npz)
def test_crop_flip_rotate_bit_exact(golden_aug_extra):
    g = golden_aug_extra
    n = 4
    x = gi.obs_u8(21, n).astype(np.float32)
    xb = gi.obs_u8(22, n, H=100, W=100).astype(np.float32)
    want = g["crop_out_u8"].astype(np.float32)
    assert np.array_equal(orc.random_crop_np(xb, g["crop_w1"], g["crop_h1"]), want)
    coef = orc.affine_coef([5] * n, n, 100, 100, a=g["crop_w1"], b=g["crop_h1"])
    assert np.array_equal(orc.affine_gather_np(xb, coef, 84, 84), want)

    want = g["flip_out_u8"].astype(np.float32)
    assert np.array_equal(orc.random_flip_np(x, g["flip_mask"]), want)
    coef = orc.affine_coef(np.where(g["flip_mask"], 4, 0), n)
    assert np.array_equal(orc.affine_gather_np(x, coef, 84, 84), want)

    want = g["rot_out_u8"].astype(np.float32)
    assert np.array_equal(orc.random_rotation_np(x, g["rot_k"]), want)
    assert np.array_equal(orc.affine_gather_np(x, orc.affine_coef(g["rot_k"], n), 84, 84), want)


def test_cutout_and_grayscale_bit_exact(golden_aug_extra):
    g = golden_aug_extra
    x = gi.obs_u8(21, 4).astype(np.float32)
    got = orc.random_cutout_color_np(x, g["cutout_w1"], g["cutout_h1"], g["cutout_color"])
    assert np.array_equal(got, g["cutout_out"])
    assert np.array_equal(orc.random_grayscale_np(x / np.float32(255.0), g["gray01_mask"]), g["gray01_out"])
    assert np.array_equal(orc.random_grayscale_np(x, g["gray255_mask"]), g["gray255_out"])


def test_splice_matches_callsite_golden(golden_aug_extra):
    # parity unpinned for kornia's HSV arithmetic (kornia absent; restated): pins the call site and the mask logic
    g = golden_aug_extra
    x = gi.smooth_obs_u8(23, 4).astype(np.float32)
    ov = gi.overlay_imgs(24, 4)
    assert np.array_equal(orc.splice_np(x, ov), g["splice_out"])
    assert np.array_equal(orc.splice_np(x, ov, val_thres=0.4), g["splice_easy_out"])
    assert np.array_equal(orc.splice_np(x, ov, hue_thres=3.5, val_thres=0.0, cs=True), g["cs_splice_out"])


def test_green_screen_bit_exact(golden_eval):
    # wrappers.do_green_screen run on seeded frames (tests/golden/make_golden.py: make_eval)
    for i, (seed, H, W) in enumerate(((41, 100, 100), (42, 84, 84))):
        x = gi.green_frame_u8(seed, H, W)
        bg = gi.obs_u8(seed + 100, 1, C=3, H=H, W=W)[0]
        assert np.array_equal(orc.green_screen_np(x, bg), golden_eval[f"gs{i}_out"])


def test_interpolate_bg_and_greenscreen_bit_exact(golden_eval):
    # VideoWrapper._interpolate_bg / _greenscreen run on seeded frames (tests/golden/make_golden.py: make_eval)
    for i, (seed, Hb, Wb, H, W) in enumerate(((51, 240, 320, 84, 84), (52, 360, 640, 100, 100), (53, 100, 100, 84, 84),
                                              (54, 84, 84, 84, 84), (55, 128, 256, 100, 100))):
        bg = gi.obs_u8(seed, 1, C=3, H=Hb, W=Wb)[0]
        bg[:, :Hb // 4, :Wb // 4] = 77
        r = orc.interpolate_bg_np(bg, (H, W))
        assert np.array_equal(r, golden_eval[f"ib{i}_out"]), i
        x = gi.green_frame_u8(seed + 10, H, W)
        assert np.array_equal(orc.green_screen_np(x, r), golden_eval[f"gsr{i}_out"]), i
