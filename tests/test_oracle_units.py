"""Single-stage properties of the oracle's arithmetic (RTL stage G/H/J/K functions)."""
import ctypes

import numpy as np

from oracle import m2v_oracle_ctypes as orc

L = orc.lib()


def fdct(x):
    x = np.ascontiguousarray(x, np.int16)
    c = np.zeros(64, np.int32)
    L.m2v_oracle_fdct(x.ctypes.data, c.ctypes.data)
    return c


def idct(d):
    d = np.ascontiguousarray(d, np.int16)
    r = np.zeros(64, np.int16)
    L.m2v_oracle_idct(d.ctypes.data, r.ctypes.data)
    return r


def quant(c, inter, Q):
    c = np.ascontiguousarray(c, np.int32)
    q = np.zeros(64, np.int16)
    L.m2v_oracle_quant(c.ctypes.data, inter, Q, q.ctypes.data)
    return q


def dequant(q, inter, Q):
    q = np.ascontiguousarray(q, np.int16)
    d = np.zeros(64, np.int16)
    L.m2v_oracle_dequant(q.ctypes.data, inter, Q, d.ctypes.data)
    return d


def _dct_matrix():
    k = np.arange(8)
    m = np.cos((2 * k[None, :] + 1) * k[:, None] * np.pi / 16) * 0.5
    m[0] *= np.sqrt(0.5)
    return m


def test_fdct_is_8x_orthonormal_dct():
    """C = (DCTM X DCTM^T + 2048) >> 12 is 8x the orthonormal DCT up to the ~1% mismatch of the integer
    (HEVC-style) basis: +-2 on small residuals, <2% of the largest coefficient on full-range ones."""
    rng = np.random.default_rng(0)
    D = _dct_matrix()
    for _ in range(50):
        x = rng.integers(-255, 256, 64).astype(np.int16)
        ref = 8 * (D @ x.reshape(8, 8).astype(float) @ D.T)
        assert np.abs(fdct(x).reshape(8, 8) - ref).max() <= 2.0 + 0.02 * np.abs(ref).max()
        xs = (x // 32).astype(np.int16)
        refs = 8 * (D @ xs.reshape(8, 8).astype(float) @ D.T)
        assert np.abs(fdct(xs).reshape(8, 8) - refs).max() <= 2.0
    assert fdct(np.full(64, -128, np.int16))[0] == -8192 and not fdct(np.full(64, -128, np.int16))[1:].any()


def test_idct_close_to_float_and_clipped():
    rng = np.random.default_rng(1)
    D = _dct_matrix()
    for _ in range(50):
        d = np.zeros(64, np.int16)
        idx = rng.integers(0, 64, 6)
        d[idx] = rng.integers(-300, 300, 6)
        ref = D.T @ d.reshape(8, 8).astype(float) @ D
        got = idct(d).reshape(8, 8)
        assert np.abs(got - np.clip(ref, -255, 255)).max() <= 1.5
    big = np.zeros(64, np.int16)
    big[0] = 2047
    assert idct(big).max() == 255 and idct(-big).min() == -255        # +-255, not -256 (RTL:781)


def test_quant_formulas():
    c = np.arange(-40, 24, dtype=np.int32) * 37
    for Q in (1, 2, 3, 4):
        q = quant(c, 1, Q)
        a = np.abs(c)
        assert np.array_equal(q, np.sign(c) * np.minimum((a + 2) >> (4 + Q), 2047))
        qi = quant(c, 0, Q)
        assert qi[0] == np.sign(c[0]) * ((abs(c[0]) >> 4) + ((abs(c[0]) >> 3) & 1))
        w = 16                                                       # W[0][1]
        assert qi[1] == np.sign(c[1]) * (((abs(c[1]) + ((w * ((3 << Q) + 2)) >> 3)) >> Q) // w)


def test_dequant_formulas():
    q = np.array([5, -5, 0, 1, -1, 700, -700, 2047] + [0] * 56, np.int16)
    for Q in (1, 2, 3, 4):
        d = dequant(q, 1, Q)
        want = [max(-2047, min(2047, (2 * int(v) + int(np.sign(v))) << Q)) for v in q[:8]]
        assert list(d[:8]) == want
        di = dequant(q, 0, Q)
        assert di[0] == 10                                           # intra DC: 2q
        w = 16
        v = -5 * w
        assert di[1] == (v << (Q - 3) if Q >= 3 else v >> (3 - Q))   # floor on negatives (RTL:2143)


def test_subsample_two_stage_rounding():
    p = np.array([[0, 1], [1, 1]], np.uint8)                         # (0+1+1)>>1=1, (1+1+1)>>1=1 -> 1; (a+b+c+d+2)>>2 = 1 too
    o = np.zeros(1, np.uint8)
    L.m2v_oracle_subsample(p.ctypes.data, 2, 2, o.ctypes.data)
    assert o[0] == 1
    p = np.array([[0, 1], [0, 0]], np.uint8)                         # two-stage: mean2(1,0)=1 ; single stage (1+2)>>2 = 0
    L.m2v_oracle_subsample(p.ctypes.data, 2, 2, o.ctypes.data)
    assert o[0] == 1
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (16, 32), dtype=np.uint8)
    out = np.zeros((8, 16), np.uint8)
    L.m2v_oracle_subsample(a.ctypes.data, 32, 16, out.ctypes.data)
    h = (a[:, 0::2].astype(int) + a[:, 1::2] + 1) >> 1
    assert np.array_equal(out, (h[0::2] + h[1::2] + 1) >> 1)
