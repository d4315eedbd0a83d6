#!/usr/bin/env python3
"""Mint golden vectors from the UNMODIFIED reference (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python3 -O tests/golden/make_golden.py

The reference (/root/reference/src) is imported on CPU with empty placeholder
modules for the third-party packages that are absent here (torchvision, kornia,
termcolor, tensorboard) and `.cuda()` neutralised (recipe: SURVEY.md Appendix A).
`-O` strips `assert x.is_cuda`.  Nothing from the reference is copied: only the
numeric OUTPUTS of its functions on seeded inputs are stored, together with the
explicit random parameters that were drawn, as small .npz files next to this
script.  Inputs are regenerated on both sides by tests/golden/inputs.py.

Stand-ins for absent libraries (restatement-defined, "parity unpinned"):
  * kornia.augmentation.RandomCrop -> integer slice at recorded offsets.
  * torchvision.transforms.ColorJitter -> torch restatement of torchvision
    0.8.2's functional_tensor algorithm (see _ColorJitter below).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import inputs as gi  # noqa: E402

REF = "/root/reference/src"


def _install_stubs():
    sys.path.insert(0, REF)
    for n in ["torchvision", "torchvision.transforms", "torchvision.datasets", "kornia",
              "kornia.augmentation", "kornia.color", "termcolor", "torch.utils.tensorboard"]:
        sys.modules[n] = types.ModuleType(n)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
    sys.modules["kornia"].augmentation = sys.modules["kornia.augmentation"]
    sys.modules["kornia"].color = sys.modules["kornia.color"]
    sys.modules["termcolor"].colored = lambda s, *a, **k: s
    sys.modules["torch.utils.tensorboard"].SummaryWriter = object
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.Tensor.get_device = lambda self: "cpu"

    class RandomCrop:  # stand-in for the absent kornia op; records what it drew
        last = None

        def __init__(self, size):
            self.size = size

        def __call__(self, x):
            n, c, H, W = x.shape
            h, w = self.size
            oy = torch.randint(0, H - h + 1, (n,))
            ox = torch.randint(0, W - w + 1, (n,))
            RandomCrop.last = (oy.numpy().astype(np.int32), ox.numpy().astype(np.int32))
            return torch.stack([x[i, :, oy[i]:oy[i] + h, ox[i]:ox[i] + w] for i in range(n)])

    sys.modules["kornia.augmentation"].RandomCrop = RandomCrop

    class _ColorJitter:
        """torchvision 0.8.2 ColorJitter on float tensors, restated with torch ops."""
        last = None

        def __init__(self, brightness, contrast, saturation, hue):
            assert brightness == 0
            self.contrast, self.saturation, self.hue = contrast, saturation, hue

        def to(self, *_a, **_k):
            return self

        @staticmethod
        def _gray(img):
            r, g, b = img.unbind(dim=-3)
            return (0.2989 * r + 0.587 * g + 0.114 * b).unsqueeze(-3)

        @staticmethod
        def _blend(a, b, ratio):
            return (ratio * a + (1.0 - ratio) * b).clamp(0, 1.0)

        @staticmethod
        def _rgb2hsv(img):
            r, g, b = img.unbind(dim=-3)
            maxc = torch.max(img, dim=-3).values
            minc = torch.min(img, dim=-3).values
            eqc = maxc == minc
            cr = maxc - minc
            ones = torch.ones_like(maxc)
            s = cr / torch.where(eqc, ones, maxc)
            div = torch.where(eqc, ones, cr)
            rc, gc, bc = (maxc - r) / div, (maxc - g) / div, (maxc - b) / div
            hr = (maxc == r) * (bc - gc)
            hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
            hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
            h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
            return torch.stack((h, s, maxc), dim=-3)

        @staticmethod
        def _hsv2rgb(img):
            h, s, v = img.unbind(dim=-3)
            i = torch.floor(h * 6.0)
            f = (h * 6.0) - i
            i = i.to(torch.int32) % 6
            p = torch.clamp(v * (1.0 - s), 0.0, 1.0)
            q = torch.clamp(v * (1.0 - (s * f)), 0.0, 1.0)
            t = torch.clamp(v * (1.0 - (s * (1.0 - f))), 0.0, 1.0)
            mask = i.unsqueeze(-3) == torch.arange(6).view(-1, 1, 1)
            a1 = torch.stack((v, q, p, p, t, v), dim=-3)
            a2 = torch.stack((t, v, v, q, p, p), dim=-3)
            a3 = torch.stack((p, p, t, v, v, q), dim=-3)
            a4 = torch.stack((a1, a2, a3), dim=-4)
            return torch.einsum("...ijk,...xijk->...xjk", mask.to(img.dtype), a4)

        def __call__(self, img):
            order = torch.randperm(4)
            rec = dict(order=order.numpy().astype(np.int32), contrast=1.0, saturation=1.0, hue=0.0)
            for fn in order.tolist():
                if fn == 1:
                    c = float(torch.tensor(1.0).uniform_(self.contrast[0], self.contrast[1]).item())
                    rec["contrast"] = c
                    mean = torch.mean(self._gray(img), dim=(-3, -2, -1), keepdim=True)
                    img = self._blend(img, mean, c)
                elif fn == 2:
                    s = float(torch.tensor(1.0).uniform_(self.saturation[0], self.saturation[1]).item())
                    rec["saturation"] = s
                    img = self._blend(img, self._gray(img), s)
                elif fn == 3:
                    hf = float(torch.tensor(1.0).uniform_(self.hue[0], self.hue[1]).item())
                    rec["hue"] = hf
                    hsv = self._rgb2hsv(img)
                    h, s_, v = hsv.unbind(dim=-3)
                    h = (h + hf) % 1.0
                    img = self._hsv2rgb(torch.stack((h, s_, v), dim=-3))
            _ColorJitter.last = rec
            return img

    sys.modules["torchvision.transforms"].ColorJitter = _ColorJitter

    def rgb_to_hsv(image, eps=1e-6):
        """Stand-in for the absent kornia.color.rgb_to_hsv: torch restatement of kornia 0.5.x kornia/color/hsv.py
        (first-maximum channel selection, hue in radians)."""
        import math
        maxc, _ = image.max(-3)
        maxc_mask = image == maxc.unsqueeze(-3)
        _, max_indices = ((maxc_mask.cumsum(-3) == 1) & maxc_mask).max(-3)
        minc = image.min(-3)[0]
        v = maxc
        deltac = maxc - minc
        s = deltac / (v + eps)
        deltac = torch.where(deltac == 0, torch.ones_like(deltac), deltac)
        maxc_tmp = maxc.unsqueeze(-3) - image
        rc, gc, bc = maxc_tmp[..., 0, :, :], maxc_tmp[..., 1, :, :], maxc_tmp[..., 2, :, :]
        h = torch.stack([bc - gc, 2.0 * deltac + rc - bc, 4.0 * deltac + gc - rc], dim=-3)
        h = torch.gather(h, dim=-3, index=max_indices[..., None, :, :]).squeeze(-3)
        h = h / deltac
        h = (h / 6.0) % 1.0
        h = 2 * math.pi * h
        return torch.stack([h, s, v], dim=-3)

    sys.modules["kornia.color"].rgb_to_hsv = rgb_to_hsv
    return RandomCrop, _ColorJitter


def make_aug():
    RandomCrop, CJ = _install_stubs()
    import augmentations as ref_aug

    B = 2
    out = {}
    obs_u8 = gi.obs_u8(seed=11, B=B)
    x = torch.from_numpy(obs_u8.astype(np.float32))

    # random_shift (augmentations.py:378-382) -- F.pad half is the reference's; crop is the stand-in
    torch.manual_seed(101)
    y = ref_aug.random_shift(x.clone(), pad=4)
    oy, ox = RandomCrop.last
    out["shift_out_u8"] = y.numpy().astype(np.uint8)
    assert np.array_equal(out["shift_out_u8"].astype(np.float32), y.numpy())
    out["shift_oy"], out["shift_ox"] = oy, ox

    # random_conv (augmentations.py:265-274): per-sample torch.randn(3,3,3,3) on the default generator
    torch.manual_seed(202)
    y = ref_aug.random_conv(x.clone())
    torch.manual_seed(202)
    w = torch.stack([torch.randn(3, 3, 3, 3) for _ in range(B)])
    out["conv_out"] = y.numpy()
    out["conv_weights"] = w.numpy()

    # random_overlay (augmentations.py:253-262) with a seeded U[0,1) image batch in place of the dataset
    ov = torch.from_numpy(gi.overlay_imgs(seed=12, B=B))
    ref_aug.dataloader = object()
    ref_aug._get_data_batch = lambda batch_size: ov.clone()
    y = ref_aug.random_overlay(x.clone())
    out["overlay_out"] = y.numpy()

    # color_jitter (augmentations.py:225-237), literal ("compat") path on 0..255 floats
    torch.manual_seed(303)
    np.random.seed(303)
    xin = x.clone()
    y = ref_aug.color_jitter(xin, p=1)
    assert y.data_ptr() == xin.data_ptr()  # in place
    np.random.seed(303)
    idx = np.random.randint(0, B, int(1 * B * 3))
    rec = CJ.last
    out["jitter_out"] = y.numpy()
    out["jitter_idx"] = idx.astype(np.int32)
    out["jitter_order"] = rec["order"]
    out["jitter_params"] = np.array([rec["contrast"], rec["saturation"], rec["hue"]], dtype=np.float64)

    # same call on [0,1]-scaled input: exercises the non-degenerate arithmetic ("sane" mode, scale=255)
    # on image-like frames (gray patches, saturated primaries: every rgb2hsv branch)
    torch.manual_seed(808)
    np.random.seed(808)
    xin = torch.from_numpy(gi.smooth_obs_u8(seed=13, B=B).astype(np.float32)) / 255.0
    y = ref_aug.color_jitter(xin, p=1)
    np.random.seed(808)
    idx = np.random.randint(0, B, int(1 * B * 3))
    rec = CJ.last
    out["jitter01_out"] = y.numpy()
    out["jitter01_idx"] = idx.astype(np.int32)
    out["jitter01_order"] = rec["order"]
    out["jitter01_params"] = np.array([rec["contrast"], rec["saturation"], rec["hue"]], dtype=np.float64)

    np.savez_compressed(os.path.join(HERE, "aug_golden.npz"), **out)
    print("wrote aug_golden.npz", {k: v.shape for k, v in out.items()})


def make_extra():
    """Rest of the registry (augmentations.py:436-449): crop / flip / rotate / cutout_color / grayscale.
    The reference draws from np.random / torch's default generator; each call is seeded and the same draws are
    replayed afterwards to record the parameters.  Must run in the process that already installed the stubs."""
    import augmentations as ref_aug

    B = 4
    out = {}
    x = torch.from_numpy(gi.obs_u8(seed=21, B=B).astype(np.float32))

    # random_crop (L385-433) 100x100 -> 84x84, drawn and explicit offsets
    xb = torch.from_numpy(gi.obs_u8(seed=22, B=B, H=100, W=100).astype(np.float32))
    torch.manual_seed(404)
    y, w1, h1 = ref_aug.random_crop(xb.clone(), size=84, return_w1_h1=True)
    torch.manual_seed(404)
    w1r = torch.LongTensor(B).random_(0, 16)
    h1r = torch.LongTensor(B).random_(0, 16)
    assert torch.equal(w1, w1r) and torch.equal(h1, h1r)
    out["crop_out_u8"] = y.numpy().astype(np.uint8)
    out["crop_w1"], out["crop_h1"] = w1.numpy().astype(np.int32), h1.numpy().astype(np.int32)
    assert y.shape == (B, 9, 84, 84)

    # random_flip (L139-168); p raised so both branches occur in 4 samples
    np.random.seed(505)
    y = ref_aug.random_flip(x.clone(), p=0.5)
    np.random.seed(505)
    mask = np.random.uniform(0.0, 1.0, size=(B,)) <= 0.5
    assert 0 < mask.sum() < B
    out["flip_out_u8"], out["flip_mask"] = y.numpy().astype(np.uint8), mask

    # random_rotation (L171-213)
    np.random.seed(606)
    y = ref_aug.random_rotation(x.clone(), p=0.8)
    np.random.seed(606)
    rnd = np.random.uniform(0.0, 1.0, size=(B,))
    rot = np.random.randint(1, 4, size=(B,)) * (rnd <= 0.8)
    out["rot_out_u8"], out["rot_k"] = y.numpy().astype(np.uint8), rot.astype(np.int32)
    assert len(set(rot.tolist())) >= 3, rot

    # random_cutout_color (L327-350) -- numpy on the host in the reference
    np.random.seed(707)
    y = ref_aug.random_cutout_color(x.clone())
    np.random.seed(707)
    w1 = np.random.randint(10, 30, B)
    h1 = np.random.randint(10, 30, B)
    col = np.random.randint(0, 255, size=(B, 9)) / 255.0
    out["cutout_out"] = y.numpy()
    out["cutout_w1"], out["cutout_h1"] = w1.astype(np.int32), h1.astype(np.int32)
    out["cutout_color"] = col.astype(np.float32)

    # random_grayscale (L89-136): on [0,1] frames (what the arithmetic is written for) and, literally, on the
    # 0..255 floats the training code would hand it (uint8 wrap of x*255)
    for tag, xin, seed in (("gray01", x / 255.0, 808), ("gray255", x.clone(), 809)):
        np.random.seed(seed)
        y = ref_aug.random_grayscale(xin.clone(), p=0.5)
        np.random.seed(seed)
        mask = np.random.uniform(0.0, 1.0, size=(B,)) <= 0.5
        assert 0 < mask.sum() < B
        out[tag + "_out"], out[tag + "_mask"] = y.numpy(), mask

    # splice / CS_splice (L277-324): kornia's rgb_to_hsv is the stand-in above (parity unpinned for its
    # arithmetic); the goldens pin the reference's call site: batch-dim overlay tiling, mask logic, /255 ... *255.
    # Image-like frames (gray patches, saturated primaries) so every branch of the HSV code is hit.
    xs = torch.from_numpy(gi.smooth_obs_u8(seed=23, B=B).astype(np.float32))
    ov = torch.from_numpy(gi.overlay_imgs(seed=24, B=B))
    ref_aug.dataloader = object()
    ref_aug._get_data_batch = lambda batch_size: ov.clone()
    xin = xs.clone()
    out["splice_out"] = ref_aug.splice(xin).numpy()
    out["splice_easy_out"] = ref_aug.splice(xin, val_thres=0.4).numpy()
    out["cs_splice_out"] = ref_aug.CS_splice(xin).numpy()
    assert torch.equal(xin, xs)  # neither mutates its input
    for k in ("splice_out", "splice_easy_out", "cs_splice_out"):
        frac = float((np.abs(out[k] - xs.numpy()) > 0.5).mean())
        assert 0.05 < frac < 0.95, (k, frac)  # both branches well populated

    np.savez_compressed(os.path.join(HERE, "aug_extra_golden.npz"), **out)
    print("wrote aug_extra_golden.npz", {k: v.shape for k, v in out.items()})


def make_eval():
    """Eval-side pixel op: wrappers.do_green_screen (src/env/wrappers.py:231-260), the reference's own function run
    on seeded frames.  gym / dmc2gym are absent (empty placeholders, never called); torchvision's
    TF.to_pil_image is PIL.Image.fromarray on the HWC view, which is all that function does for a uint8 CHW tensor."""
    from PIL import Image
    for n in ["gym", "dmc2gym", "torchvision.transforms.functional"]:
        sys.modules[n] = types.ModuleType(n)
    sys.modules["gym"].Wrapper = object
    sys.modules["gym"].spaces = types.SimpleNamespace(Box=object)
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    sys.modules["torchvision.transforms.functional"].to_pil_image = \
        lambda t: Image.fromarray(np.ascontiguousarray(t.numpy().transpose(1, 2, 0)))
    sys.path.insert(0, os.path.join(REF, "env"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_wrappers", os.path.join(REF, "env", "wrappers.py"))
    wr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(wr)
    out = {}
    for i, (seed, H, W) in enumerate(((41, 100, 100), (42, 84, 84))):
        x = gi.green_frame_u8(seed, H, W)
        bg = gi.obs_u8(seed + 100, 1, C=3, H=H, W=W)[0]
        y = wr.do_green_screen(x.copy(), bg.copy())
        assert y.shape == x.shape and y.dtype == np.uint8
        frac = float((y != x).any(axis=0).mean())
        assert 0.1 < frac < 0.9, frac
        out[f"gs{i}_out"] = y
    # VideoWrapper._interpolate_bg (wrappers.py:336-340) and _greenscreen (L342-348) = resize + keying: the reference's
    # own methods (unbound: they do not touch `self`) on seeded video-frame-like backgrounds of several sizes
    for i, (seed, Hb, Wb, H, W) in enumerate(((51, 240, 320, 84, 84), (52, 360, 640, 100, 100), (53, 100, 100, 84, 84),
                                              (54, 84, 84, 84, 84), (55, 128, 256, 100, 100))):
        bg = gi.obs_u8(seed, 1, C=3, H=Hb, W=Wb)[0]
        bg[:, :Hb // 4, :Wb // 4] = 77  # flat areas: k/255*255 lands within an ulp of an integer before the truncating cast
        r = wr.VideoWrapper._interpolate_bg(None, bg.copy(), (H, W))
        assert r.shape == (3, H, W) and r.dtype == np.uint8
        out[f"ib{i}_out"] = r
        x = gi.green_frame_u8(seed + 10, H, W)
        out[f"gsr{i}_out"] = wr.do_green_screen(x.copy(), r.copy())
    np.savez_compressed(os.path.join(HERE, "eval_golden.npz"), **out)
    print("wrote eval_golden.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    make_aug()
    make_extra()
    make_eval()  # networks / update() goldens: make_golden_sac.py (separate process)
