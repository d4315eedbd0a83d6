"""CPU restatement of the f16-pair arithmetic of csrc/encoder_x2.hip (TEST INFRASTRUCTURE ONLY: imported by tests/, never by the
product).  Not a reference algorithm -- the reference convolves in float32 (src/algorithms/modules.py:145-165, torch Conv2d) --
but the numerical method the HIP kernels use to reproduce float32 products on the f16 matrix pipe, restated in numpy so that its
accuracy claims can be checked without a GPU:

    s      = 2^k with max|x| * 2^k in [2^14, 2^15)                       (scale_exp)
    hi     = f16(x s),  lo' = f16((x s - hi) 2^11)                        (split, round-to-nearest-even)
    x.y    = [ sum hi_x hi_y  +  2^-11 sum (hi_x lo'_y + lo'_x hi_y) ] / (s_x s_y)   (three products; lo' lo' dropped)

numpy's float16 conversions round to nearest even and keep subnormals, like v_cvt_pk_f16_f32; f16 x f16 products are exact in
float32, and the accumulation here is float64 (the MFMA accumulates in float32: its own rounding is measured on the device by
tools/x3probe.hip, not here)."""
import numpy as np


def scale_exp(amax):
    """k with amax * 2^k in [2^14, 2^15); clamped like the kernels (biased exponent in [67, 187])"""
    bits = np.float32(amax).view(np.uint32)
    e = int((bits >> 23) & 0xFF)
    e = min(max(e, 67), 187)
    return 141 - e


def split(x, k):
    """x (float32 array) -> (hi, lo') float16 arrays under the scale 2^k"""
    xs = np.ldexp(x.astype(np.float32), k).astype(np.float32)
    hi = xs.astype(np.float16)
    lo = ((xs - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    return hi, lo


def dot3(a, b, scaled=True):
    """rows of a (M,K) . columns of b (K,N) with the three-product scheme; returns float64 (M,N)"""
    ka = scale_exp(np.abs(a).max()) if scaled else 0
    kb = scale_exp(np.abs(b).max()) if scaled else 0
    ah, al = split(a, ka)
    bh, bl = split(b, kb)
    f = lambda t: t.astype(np.float64)  # noqa: E731  (f16 x f16 is exact in float32 and in float64)
    main = f(ah) @ f(bh)
    cross = f(ah) @ f(bl) + f(al) @ f(bh)
    return np.ldexp(main + cross / 2048.0, -(ka + kb))
