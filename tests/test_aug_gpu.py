"""GPU parity: HIP augmentation kernels (through the C ABI) vs the CPU oracle and the golden
vectors minted from the reference.  Integer/index ops bit-exact; float ops within the stated
tolerance (1e-5 of the 0..255 scale for random_conv; bit-exact for random_overlay)."""
import numpy as np
import pytest
import torch

import inputs as gi
from augcl_amd import _lib
from augcl_amd import augmentations as aug
from oracle import aug_oracle as orc

pytestmark = pytest.mark.gpu

CONV_ATOL = 1e-5 * 255.0  # north_star tolerance on the 0..255 output scale
CONV_ATOL_VS_ORACLE = 5e-4  # what we actually hold the kernel to against the C oracle


def dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def host(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------ random_shift
def test_shift_golden(cuda, golden_aug):
    x = gi.obs_u8(11, 2)
    oy, ox = golden_aug["shift_oy"], golden_aug["shift_ox"]
    want = golden_aug["shift_out_u8"]
    got = aug.random_shift(dev(x.astype(np.float32), cuda), oy=oy, ox=ox)
    assert np.array_equal(host(got), want.astype(np.float32))
    got = aug.random_shift(dev(x, cuda), oy=oy, ox=ox)  # u8 -> f32 fused
    assert got.dtype == torch.float32 and np.array_equal(host(got), want.astype(np.float32))
    out = torch.empty(x.shape, dtype=torch.uint8, device=cuda)
    aug.random_shift(dev(x, cuda), oy=oy, ox=ox, out=out)  # u8 -> u8
    assert np.array_equal(host(out), want)


@pytest.mark.parametrize("shape,pad", [((81, 9, 84, 84), 4), ((5, 9, 100, 100), 4), ((3, 6, 50, 50), 3),
                                       ((4, 3, 33, 47), 5), ((2, 9, 84, 84), 0), ((7, 1, 84, 84), 12)])
def test_shift_all_offsets_bit_exact(cuda, shape, pad):
    B, C, H, W = shape
    rs = np.random.RandomState(1)
    x = rs.randint(0, 256, size=shape).astype(np.uint8)
    n = 2 * pad + 1
    # every (oy, ox) pair at least once when B allows, otherwise random
    oy = (np.arange(B) // n % n).astype(np.int32) if B >= n * n else rs.randint(0, n, B).astype(np.int32)
    ox = (np.arange(B) % n).astype(np.int32) if B >= n * n else rs.randint(0, n, B).astype(np.int32)
    xf = x.astype(np.float32) + rs.random_sample(shape).astype(np.float32)  # non-integer floats too
    got = aug.random_shift(dev(xf, cuda), pad, oy=oy, ox=ox)
    assert np.array_equal(host(got), orc.random_shift_c(xf, oy, ox, pad))
    got = aug.random_shift(dev(x, cuda), pad, oy=oy, ox=ox)
    assert np.array_equal(host(got), orc.random_shift_c(x, oy, ox, pad, out_dtype=np.float32))
    out = torch.empty(shape, dtype=torch.uint8, device=cuda)
    aug.random_shift(dev(x, cuda), pad, oy=oy, ox=ox, out=out)
    assert np.array_equal(host(out), orc.random_shift_c(x, oy, ox, pad))


def test_shift_unaligned_view_takes_generic_path(cuda):
    rs = np.random.RandomState(2)
    buf = torch.from_numpy(rs.random_sample(2 * 9 * 84 * 84 + 1).astype(np.float32)).to(cuda)
    x = buf[1:].view(2, 9, 84, 84)  # 4-byte aligned only
    oy, ox = np.array([0, 8], np.int32), np.array([7, 1], np.int32)
    got = aug.random_shift(x, oy=oy, ox=ox)
    assert np.array_equal(host(got), orc.random_shift_c(host(x), oy, ox))


def test_shift_empty_batch(cuda):
    x = torch.empty((0, 9, 84, 84), dtype=torch.float32, device=cuda)
    assert aug.random_shift(x).shape == x.shape


def test_shift_rng_internal_matches_exported_stream(cuda):
    B = 64
    x = gi.obs_u8(4, B)
    aug.manual_seed(1234)
    aug.set_sample_base(0)
    got = aug.random_shift(dev(x, cuda))
    oy = torch.empty(B, dtype=torch.int32, device=cuda)
    ox = torch.empty(B, dtype=torch.int32, device=cuda)
    _lib.call("augcl_philox_shift_offsets", 1234, 0, 0, B, 4, oy.data_ptr(), ox.data_ptr(), None)
    torch.cuda.synchronize()
    oyh, oxh = host(oy), host(ox)
    assert oyh.min() >= 0 and oyh.max() <= 8 and oxh.min() >= 0 and oxh.max() <= 8
    assert len(set(zip(oyh.tolist(), oxh.tolist()))) > 20  # it is actually random
    assert np.array_equal(host(got), orc.random_shift_c(x, oyh, oxh, out_dtype=np.float32))
    # sharding invariance: rank 1 of 2 reproduces the second half
    aug.manual_seed(1234)
    aug.set_sample_base(B // 2)
    got2 = aug.random_shift(dev(x[B // 2:], cuda))
    aug.set_sample_base(0)
    assert torch.equal(got2, got[B // 2:])


def test_shift_full_size_properties(cuda):
    """BASELINE configs' batch sizes, size-independent properties (no oracle loop)."""
    for B in (256, 512, 1024):
        x = torch.randint(0, 256, (B, 9, 84, 84), dtype=torch.uint8, device=cuda)
        c = torch.full((B,), 4, dtype=torch.int32, device=cuda)
        assert torch.equal(aug.random_shift(x, oy=c, ox=c), x.float())  # centred crop = identity
        oy = torch.randint(0, 9, (B,), dtype=torch.int32, device=cuda)
        ox = torch.randint(0, 9, (B,), dtype=torch.int32, device=cuda)
        y = aug.random_shift(x, oy=oy, ox=ox)
        xp = torch.nn.functional.pad(x.float(), (4, 4, 4, 4), mode="replicate")  # torch restatement on device
        iy = (oy.long()[:, None] + torch.arange(84, device=cuda)[None, :])
        ix = (ox.long()[:, None] + torch.arange(84, device=cuda)[None, :])
        want = xp[torch.arange(B, device=cuda)[:, None, None, None], torch.arange(9, device=cuda)[None, :, None, None],
                  iy[:, None, :, None], ix[:, None, None, :]]
        assert torch.equal(y, want)


# ------------------------------------------------------------------ random_conv
def test_conv_golden(cuda, golden_aug):
    x = gi.obs_u8(11, 2).astype(np.float32)
    got = host(aug.random_conv(dev(x, cuda), weights=golden_aug["conv_weights"]))
    err = np.abs(got - golden_aug["conv_out"]).max()
    assert err <= CONV_ATOL, err
    assert err <= CONV_ATOL_VS_ORACLE, err


def test_conv_reference_weight_stream_reproduces_the_reference_run(cuda, golden_aug):
    """src/augmentations.py:268-269 draws one CPU torch.randn(3,3,3,3) per sample, in order, on the default generator.
    Seeded as tests/golden/make_golden.py seeds the reference (torch.manual_seed(202)), the reference-stream mode reproduces
    the reference's OWN output without being handed its weights -- per call (weights="reference"), as the module default
    (set_conv_weight_stream), and through the fused shift + conv front end."""
    x = gi.obs_u8(11, 2).astype(np.float32)
    torch.manual_seed(202)
    got = host(aug.random_conv(dev(x, cuda), weights="reference"))
    assert np.abs(got - golden_aug["conv_out"]).max() <= CONV_ATOL
    aug.set_conv_weight_stream("reference")
    try:
        torch.manual_seed(202)
        got2 = host(aug.random_conv(dev(x, cuda)))
        assert np.array_equal(got, got2)
        # fused front end: random_conv(random_shift(x)) with a zero shift = pad 4 / offset 4 = the identity crop
        torch.manual_seed(202)
        z = np.full(2, 4, np.int32)
        _, o_a = aug.shift_conv(dev(x, cuda), oy=z, ox=z)
        assert np.abs(host(o_a) - golden_aug["conv_out"]).max() <= CONV_ATOL
    finally:
        aug.set_conv_weight_stream("philox")
    torch.manual_seed(202)
    assert not np.array_equal(host(aug.random_conv(dev(x, cuda))), got)  # back on the in-kernel Philox stream


@pytest.mark.parametrize("shape", [(16, 9, 84, 84), (3, 3, 84, 84), (2, 9, 100, 100), (2, 6, 40, 52)])
def test_conv_vs_oracle(cuda, shape):
    B = shape[0]
    rs = np.random.RandomState(7)
    x = (rs.randint(0, 256, size=shape) + rs.random_sample(shape)).astype(np.float32)
    w = gi.conv_weights(8, B) * np.float32(1.5)
    got = host(aug.random_conv(dev(x, cuda), weights=w))
    want = orc.random_conv_c(x, w)
    assert got.min() >= 0.0 and got.max() <= 255.0
    assert np.abs(got - want).max() <= CONV_ATOL_VS_ORACLE, np.abs(got - want).max()


def test_conv_does_not_mutate_input_and_rng_path(cuda):
    B = 8
    x = dev(gi.obs_u8(9, B).astype(np.float32), cuda)
    x0 = x.clone()
    aug.manual_seed(77)
    y = aug.random_conv(x)
    assert torch.equal(x, x0)
    w = torch.empty(B * 81, dtype=torch.float32, device=cuda)
    _lib.call("augcl_philox_conv_weights", 77, 0, 0, B, w.data_ptr(), None)
    torch.cuda.synchronize()
    wh = host(w)
    assert abs(wh.mean()) < 0.2 and 0.8 < wh.std() < 1.2  # N(0,1)-looking
    y2 = aug.random_conv(x, weights=w)
    assert torch.equal(y, y2)  # RNG-internal == explicit with the exported stream, bit for bit
    assert np.abs(host(y) - orc.random_conv_c(host(x), wh)).max() <= CONV_ATOL_VS_ORACLE


def test_conv_full_size_properties(cuda):
    B = 256
    x = torch.randint(0, 256, (B, 9, 84, 84), device=cuda).float()
    # zero weights -> sigmoid(0)*255 = 127.5 everywhere; delta kernel on channel k -> sigmoid(x/255)*255
    y = aug.random_conv(x, weights=torch.zeros(B, 81, device=cuda))
    assert torch.equal(y, torch.full_like(y, 127.5))
    w = torch.zeros(B, 3, 3, 3, 3, device=cuda)
    for o in range(3):
        w[:, o, o, 1, 1] = 1.0
    y = aug.random_conv(x, weights=w)
    want = torch.sigmoid(x / 255.0) * 255.0
    assert (y - want).abs().max().item() <= 1e-3
    # per-sample independence: permuting the batch permutes the output
    w = torch.randn(B, 81, device=cuda)
    perm = torch.randperm(B, device=cuda)
    assert torch.equal(aug.random_conv(x, weights=w)[perm], aug.random_conv(x[perm].contiguous(), weights=w[perm].contiguous()))


# ------------------------------------------------------------------ fused shift -> conv
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_shift_conv_fused_equals_chain(cuda, dtype):
    B = 12
    x = gi.obs_u8(21, B).astype(dtype)
    oy, ox = gi.shift_offsets(22, B)
    oy[:3], ox[:3] = [0, 8, 4], [8, 0, 4]
    w = gi.conv_weights(23, B)
    o_s, o_a = aug.shift_conv(dev(x, cuda), oy=oy, ox=ox, weights=w)
    want_s = orc.random_shift_c(x.astype(np.float32), oy, ox)
    assert np.array_equal(host(o_s), want_s)
    chain = aug.random_conv(aug.random_shift(dev(x, cuda), oy=oy, ox=ox), weights=w)
    assert torch.equal(o_a, chain)  # same staging + same compute => bit-identical to the unfused kernels
    assert np.abs(host(o_a) - orc.random_conv_c(want_s, w)).max() <= CONV_ATOL_VS_ORACLE
    only_a = aug.shift_conv(dev(x, cuda), oy=oy, ox=ox, weights=w, want_shift=False)[1]
    assert torch.equal(only_a, o_a)


# ------------------------------------------------------------------ random_overlay
def test_overlay_golden_bit_exact(cuda, golden_aug):
    x = gi.obs_u8(11, 2).astype(np.float32)
    ov = gi.overlay_imgs(12, 2)
    got = host(aug.random_overlay(dev(x, cuda), imgs=dev(ov, cuda)))
    assert np.array_equal(got, golden_aug["overlay_out"])


@pytest.mark.parametrize("shape", [(32, 9, 84, 84), (2, 3, 100, 100), (3, 6, 31, 45)])
def test_overlay_vs_oracle_bit_exact(cuda, shape):
    B, C, H, W = shape
    rs = np.random.RandomState(5)
    x = (rs.randint(0, 256, size=shape) + rs.random_sample(shape)).astype(np.float32)
    ov = rs.random_sample((B, 3, H, W)).astype(np.float32)
    got = host(aug.random_overlay(dev(x, cuda), imgs=dev(ov, cuda)))
    assert np.array_equal(got, orc.random_overlay_c(x, ov))


def test_overlay_source_and_errors(cuda):
    x = torch.zeros(2, 9, 84, 84, device=cuda)
    aug.set_overlay_source(None)
    with pytest.raises(FileNotFoundError):
        aug.random_overlay(x)
    with pytest.raises(NotImplementedError):
        aug.random_overlay(x, dataset="imagenet")
    aug.set_overlay_source(aug.SyntheticOverlaySource(seed=3))
    y = aug.random_overlay(x)
    assert y.shape == x.shape and float(y.min()) >= 0 and float(y.max()) <= 127.5
    assert torch.equal(y[:, 0:3], y[:, 3:6]) and torch.equal(y[:, 0:3], y[:, 6:9])  # same image on every frame
    aug.set_overlay_source(None)


# ------------------------------------------------------------------ color_jitter
def test_jitter_golden_compat(cuda, golden_aug):
    x = dev(gi.obs_u8(11, 2).astype(np.float32), cuda)
    p = golden_aug["jitter_params"]
    y = aug.color_jitter(x, idx=golden_aug["jitter_idx"], order=golden_aug["jitter_order"], contrast=p[0],
                         saturation=p[1], hue=p[2])
    assert y.data_ptr() == x.data_ptr()  # in place, like the reference
    assert np.abs(host(y) - golden_aug["jitter_out"]).max() <= 1e-5


def test_jitter_golden_unit_range(cuda, golden_aug):
    x = dev(gi.smooth_obs_u8(13, 2).astype(np.float32) / np.float32(255.0), cuda)
    p = golden_aug["jitter01_params"]
    y = aug.color_jitter(x, idx=golden_aug["jitter01_idx"], order=golden_aug["jitter01_order"], contrast=p[0],
                         saturation=p[1], hue=p[2])
    assert np.abs(host(y) - golden_aug["jitter01_out"]).max() <= 1e-5


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("mode", ["compat", "sane"])
def test_jitter_vs_oracle(cuda, seed, mode):
    B = 6
    x = gi.smooth_obs_u8(30 + seed, B).astype(np.float32)
    if mode == "compat":
        x = x / np.float32(255.0)  # exercise the real arithmetic, not the degenerate clamp
    order, c, s, h = gi.jitter_params(seed)
    idx = np.random.RandomState(seed).randint(0, B, B * 3).astype(np.int32)  # the reference's index range
    want = orc.color_jitter_c(x, idx, order, c, s, h, scale=1.0 if mode == "compat" else 255.0)
    xt = dev(x, cuda)
    got = host(aug.color_jitter(xt, idx=idx, order=order, contrast=c, saturation=s, hue=h, mode=mode))
    tol = 1e-5 if mode == "compat" else 1e-5 * 255.0
    assert np.abs(got - want).max() <= tol, np.abs(got - want).max()
    untouched = sorted(set(range(B * 3)) - set(idx.tolist()))
    assert np.array_equal(got.reshape(B * 3, -1)[untouched], x.reshape(B * 3, -1)[untouched])


def test_jitter_draws_its_own_params_like_the_reference(cuda):
    x = dev(gi.smooth_obs_u8(2, 4).astype(np.float32), cuda)
    x0 = x.clone()
    np.random.seed(0)
    torch.manual_seed(0)
    y = aug.color_jitter(x, p=1)
    assert y.shape == x0.shape and not torch.equal(y, x0)
    # only frame-rows 0..B-1 of the 3B rows can ever be selected (reference quirk, kept)
    assert torch.equal(y.view(12, -1)[4:], x0.view(12, -1)[4:])


# ---------------------------------------------------------------- rest of the registry
@pytest.mark.gpu
def test_extra_ops_golden_bit_exact(cuda, golden_aug_extra):
    g = golden_aug_extra
    x = gi.obs_u8(21, 4).astype(np.float32)
    xb = gi.obs_u8(22, 4, H=100, W=100).astype(np.float32)
    got, w1, h1 = aug.random_crop(dev(xb, cuda), size=84, w1=torch.from_numpy(g["crop_w1"].astype(np.int64)),
                                  h1=torch.from_numpy(g["crop_h1"].astype(np.int64)), return_w1_h1=True)
    assert np.array_equal(host(got), g["crop_out_u8"].astype(np.float32))
    assert np.array_equal(host(aug.random_flip(dev(x, cuda), mask=g["flip_mask"])), g["flip_out_u8"].astype(np.float32))
    assert np.array_equal(host(aug.random_rotation(dev(x, cuda), rot=g["rot_k"])), g["rot_out_u8"].astype(np.float32))
    got = aug.random_cutout_color(dev(x, cuda), w1=g["cutout_w1"], h1=g["cutout_h1"], color=g["cutout_color"])
    assert np.array_equal(host(got), g["cutout_out"])
    got = aug.random_grayscale(dev(x / np.float32(255.0), cuda), mask=g["gray01_mask"])
    assert np.array_equal(host(got), g["gray01_out"])
    got = aug.random_grayscale(dev(x, cuda), mask=g["gray255_mask"])
    assert np.array_equal(host(got), g["gray255_out"])


@pytest.mark.gpu
def test_extra_ops_draw_like_the_reference(cuda):
    """Without explicit parameters the shims consume np.random / torch's generator exactly as the reference does
    (same calls, same order), so a seeded reference run and a seeded run here pick the same samples."""
    B = 6
    x = gi.obs_u8(31, B).astype(np.float32)
    xd = dev(x, cuda)
    np.random.seed(5)
    got = host(aug.aug_to_func["flip"](xd, p=0.5))
    np.random.seed(5)
    assert np.array_equal(got, orc.random_flip_np(x, np.random.uniform(0.0, 1.0, size=(B,)) <= 0.5))

    np.random.seed(6)
    got = host(aug.aug_to_func["rotate"](xd, p=0.7))
    np.random.seed(6)
    rnd = np.random.uniform(0.0, 1.0, size=(B,))
    rot = np.random.randint(1, 4, size=(B,)) * (rnd <= 0.7)
    assert np.array_equal(got, orc.random_rotation_np(x, rot))

    np.random.seed(7)
    got = host(aug.aug_to_func["cutout_color"](xd))
    np.random.seed(7)
    w1, h1 = np.random.randint(10, 30, B), np.random.randint(10, 30, B)
    col = np.random.randint(0, 255, size=(B, 9)) / 255.0
    assert np.array_equal(got, orc.random_cutout_color_np(x, w1, h1, col))

    np.random.seed(8)
    got = host(aug.aug_to_func["grayscale"](xd / 255.0, p=0.5))
    np.random.seed(8)
    assert np.array_equal(got, orc.random_grayscale_np(x / np.float32(255.0), np.random.uniform(0, 1, size=(B,)) <= 0.5))

    xb = gi.obs_u8(32, B, H=100, W=100).astype(np.float32)
    torch.manual_seed(9)
    got = host(aug.aug_to_func["crop"](dev(xb, cuda)))
    torch.manual_seed(9)
    w1 = torch.LongTensor(B).random_(0, 16)
    h1 = torch.LongTensor(B).random_(0, 16)
    assert np.array_equal(got, orc.random_crop_np(xb, w1.numpy(), h1.numpy()))
    # crop_max <= 0: the input comes back untouched (reference L397-400)
    same, a, b = aug.random_crop(xd, size=84, return_w1_h1=True)
    assert same is xd and a is None and b is None


@pytest.mark.gpu
def test_extra_ops_edges_and_properties(cuda):
    x = gi.obs_u8(33, 3, C=3, H=20, W=20).astype(np.float32)
    xd = dev(x, cuda)
    # four quarter turns / two flips are the identity; rot180 = flip rows and columns
    y = xd
    for _ in range(4):
        y = aug.random_rotation(y, rot=[1, 1, 1])
    assert torch.equal(y, xd)
    assert torch.equal(aug.random_flip(aug.random_flip(xd, mask=[1, 0, 1]), mask=[1, 0, 1]), xd)
    assert torch.equal(aug.random_rotation(xd, rot=[2, 2, 2]), xd.flip([2, 3]))
    assert torch.equal(aug.random_rotation(xd, rot=[1, 3, 0]),
                       torch.stack([xd[0].rot90(1, [1, 2]), xd[1].rot90(3, [1, 2]), xd[2]]))
    # cutout box clipped by the frame edge (h1 + h1 > H), as numpy slicing clips
    got = host(aug.random_cutout_color(xd, w1=[15, 0, 19], h1=[12, 19, 0], color=np.full((3, 3), 0.5, np.float32)))
    assert np.array_equal(got, orc.random_cutout_color_np(x, [15, 0, 19], [12, 19, 0], np.full((3, 3), 0.5)))
    # out-of-frame crop window is refused on the host, before any launch
    with pytest.raises(ValueError):
        aug.random_crop(xd, size=16, w1=torch.tensor([0, 5, 0]), h1=torch.tensor([0, 0, 0]))
    # empty batch
    z = torch.empty((0, 9, 84, 84), device=cuda)
    assert aug.random_flip(z).shape == z.shape and aug.random_grayscale(z).shape == z.shape
    assert aug.random_cutout_color(z).shape == z.shape
    # full batch: flip is an involution; the uint8 quantisation is idempotent (the gray mix itself is not: its
    # coefficients sum to 0.9999, so gray(g,g,g) = g-1)
    big = torch.rand((256, 9, 84, 84), device=cuda)
    m = np.arange(256) % 2 == 0
    assert torch.equal(aug.random_flip(aug.random_flip(big, mask=m), mask=m), big)
    g1 = aug.random_grayscale(big, mask=m)
    assert torch.equal(aug.random_grayscale(g1, mask=np.zeros(256, bool)), g1)
    v = g1.view(256, 3, 3, 84, 84)[m]
    assert torch.equal(v[:, :, 0], v[:, :, 1]) and torch.equal(v[:, :, 0], v[:, :, 2])


@pytest.mark.gpu
def test_splice_golden_and_oracle(cuda, golden_aug_extra):
    g = golden_aug_extra
    x = gi.smooth_obs_u8(23, 4).astype(np.float32)
    ov = gi.overlay_imgs(24, 4)
    xd, ovd = dev(x, cuda), dev(ov, cuda)
    assert np.array_equal(host(aug.splice(xd, imgs=ovd)), g["splice_out"])
    assert np.array_equal(host(aug.splice(xd, val_thres=0.4, imgs=ovd)), g["splice_easy_out"])
    assert np.array_equal(host(aug.CS_splice(xd, imgs=ovd)), g["cs_splice_out"])
    assert np.array_equal(host(xd), x)  # input untouched
    # uniform-noise frames, full batch, registered image source, odd thresholds
    B = 256
    xn = gi.obs_u8(41, B).astype(np.float32)
    src = aug.SyntheticOverlaySource(seed=3)
    aug.set_overlay_source(src)
    try:
        got = aug.aug_to_func["splice"](dev(xn, cuda), 1.0, 0.2, 0.5)
        chk = aug.SyntheticOverlaySource(seed=3)  # replays the registered source's stream
        imgs = host(chk(B, 84, cuda))
        want = orc.splice_np(xn, imgs, 1.0, 0.2, 0.5)
        assert np.array_equal(host(got), want)
        got = aug.aug_to_func["CS_splice"](dev(xn, cuda))
        imgs2 = host(chk(B, 84, cuda))
        assert np.array_equal(host(got), orc.splice_np(xn, imgs2, 3.5, 0.0, 0.0, cs=True))
    finally:
        aug.set_overlay_source(None)
    with pytest.raises(FileNotFoundError):
        aug.splice(xd)


@pytest.mark.gpu
def test_green_screen_golden_and_all_colours(cuda, golden_eval):
    from augcl_amd import eval_ops
    for i, (seed, H, W) in enumerate(((41, 100, 100), (42, 84, 84))):
        x = gi.green_frame_u8(seed, H, W)
        bg = gi.obs_u8(seed + 100, 1, C=3, H=H, W=W)[0]
        assert np.array_equal(eval_ops.do_green_screen(x, bg), golden_eval[f"gs{i}_out"])
    # every one of the 2^24 colours once: (N,3,256,256) with r = frame index
    r = np.arange(256, dtype=np.uint8)
    x = np.empty((256, 3, 256, 256), np.uint8)
    x[:, 0] = r[:, None, None]
    x[:, 1] = r[None, :, None]
    x[:, 2] = r[None, None, :]
    bg = np.full_like(x, 7)
    got = host(eval_ops.green_screen(dev(x, cuda), dev(bg, cuda)))
    assert np.array_equal(got, orc.green_screen_np(x, bg))
    assert 0.05 < float((got[:, 0] == 7).mean()) < 0.5


def test_interpolate_bg_and_fused_greenscreen(cuda, golden_eval):
    """VideoWrapper._interpolate_bg / _greenscreen (src/env/wrappers.py:336-348): uint8-exact against goldens minted from
    the reference's own methods, against the oracle on further sizes, and the one-launch resize+key == the two steps."""
    from augcl_amd import eval_ops
    for i, (seed, Hb, Wb, H, W) in enumerate(((51, 240, 320, 84, 84), (52, 360, 640, 100, 100), (53, 100, 100, 84, 84),
                                              (54, 84, 84, 84, 84), (55, 128, 256, 100, 100))):
        bg = gi.obs_u8(seed, 1, C=3, H=Hb, W=Wb)[0]
        bg[:, :Hb // 4, :Wb // 4] = 77
        r = eval_ops.interpolate_bg(dev(bg, cuda), (H, W))
        assert np.array_equal(host(r), golden_eval[f"ib{i}_out"]), i
        x = gi.green_frame_u8(seed + 10, H, W)
        fused = eval_ops.greenscreen(dev(x, cuda), dev(bg, cuda))
        assert np.array_equal(host(fused), golden_eval[f"gsr{i}_out"]), i
        assert torch.equal(fused, eval_ops.green_screen(dev(x, cuda), r))
    rs = np.random.RandomState(9)
    for Hb, Wb, H, W in ((480, 854, 84, 84), (101, 203, 100, 100), (84, 100, 100, 84), (720, 1280, 84, 84)):
        bgs = rs.randint(0, 256, (4, 3, Hb, Wb)).astype(np.uint8)  # a batch: one background per frame
        bgs[0] = 200
        got = host(eval_ops.interpolate_bg(dev(bgs, cuda), (H, W)))
        assert got.shape == (4, 3, H, W) and np.array_equal(got, orc.interpolate_bg_np(bgs, (H, W))), (Hb, Wb, H, W)
