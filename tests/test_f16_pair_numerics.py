"""The f16-pair method of csrc/encoder_x2.hip (precision "f32" of the conv stack), restated in numpy (oracle/f16_pair_oracle.py):
its accuracy against float64 on the operand distributions the encoder sees, why it needs the per-tensor scale, and where its
range ends.  CPU only; the device's own accumulation rounding is measured by tools/x3probe.hip and held to 1e-5 by
tests/test_referee_gpu.py."""
import numpy as np

from oracle import f16_pair_oracle as fo


def _case(dist, rs, K=288, M=32, N=64):
    a = (rs.standard_normal((M, K)) * 0.06).astype(np.float32)  # weights
    g = rs.standard_normal((K, N))
    b = {"gauss": g, "relu": np.maximum(g, 0), "tiny": g * 1e-6, "wide": g * np.exp(4.0 * rs.standard_normal((K, N)))}[dist]
    return a, b.astype(np.float32)


def _rel_l2(got, a, b):
    want = a.astype(np.float64) @ b.astype(np.float64)
    return float(np.linalg.norm(got - want) / np.linalg.norm(want))


def test_three_products_reach_float32_accuracy():
    rs = np.random.RandomState(0)
    for dist in ("gauss", "relu", "tiny", "wide"):
        a, b = _case(dist, rs)
        e = _rel_l2(fo.dot3(a, b), a, b)
        # the split and the dropped lo' lo' product together cost ~2^-24 relative per product: 7-9e-8 in the norm, before any
        # accumulation rounding (the device adds its own: 1.4e-7 in all, tools/x3probe.hip; a float32 FMA chain has 3.0e-7)
        assert e <= 1e-7, (dist, e)


def test_scale_is_what_makes_it_work():
    rs = np.random.RandomState(1)
    a, b = _case("tiny", rs)  # gradients of 1e-6: f16's subnormal range without a scale
    assert _rel_l2(fo.dot3(a, b, scaled=False), a, b) >= 2e-6
    assert _rel_l2(fo.dot3(a, b, scaled=True), a, b) <= 1e-7
    big = (b * 1e12).astype(np.float32)  # activations of 1e6: overflow f16 without a scale
    with np.errstate(all="ignore"):
        assert not np.isfinite(fo.dot3(a, big, scaled=False)).all()
    assert _rel_l2(fo.dot3(a, big), a, big) <= 1e-7


def test_split_represents_23_bits_within_2_to_26_of_the_maximum():
    rs = np.random.RandomState(2)
    x = (rs.standard_normal(20000) * np.exp2(rs.uniform(-28, 0, 20000))).astype(np.float32)
    x[0] = 1.5  # the maximum
    k = fo.scale_exp(np.abs(x).max())
    assert 2.0 ** 14 <= np.abs(x).max() * 2.0 ** k < 2.0 ** 15
    hi, lo = fo.split(x, k)
    back = np.ldexp(hi.astype(np.float64) + lo.astype(np.float64) / 2048.0, -k)
    rel = np.abs(back - x.astype(np.float64)) / np.abs(x.astype(np.float64))
    near = np.abs(x) >= 2.0 ** -26 * np.abs(x).max()   # hi AND lo' normal: two full 11-bit parts
    inside = np.abs(x) >= 2.0 ** -29 * np.abs(x).max()  # hi (nearly) normal, lo' going subnormal
    assert near.sum() > 15000 and inside.sum() > near.sum()
    assert float(rel[near].max()) <= 2.0 ** -22.9
    assert float(rel[inside].max()) <= 2.0 ** -21
    assert float(np.abs(back - x.astype(np.float64))[~inside].max()) <= 2.0 ** -36 * 1.5  # below the range: absolutely small
    # far below the range the error stays ABSOLUTELY small: 2^-36 of the maximum
    tiny = np.float32(2.0 ** -40) * np.arange(1, 100, dtype=np.float32)
    hi, lo = fo.split(tiny, k)
    back = np.ldexp(hi.astype(np.float64) + lo.astype(np.float64) / 2048.0, -k)
    assert float(np.abs(back - tiny.astype(np.float64)).max()) <= 2.0 ** -36 * 1.5


def test_scale_exp_clamps():
    assert fo.scale_exp(0.0) == 141 - 67 and fo.scale_exp(np.inf) == 141 - 187
    for v in (1e-3, 0.7, 1.0, 255.0, 3e4):
        k = fo.scale_exp(v)
        assert 2.0 ** 14 <= v * 2.0 ** k < 2.0 ** 15
