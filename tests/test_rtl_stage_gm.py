"""CPU: the oracle's stage G/H/J/K/M unit functions against the register-width emulation of the RTL in
tests/rtl_stage_gm.py, on random, structured and extreme inputs (full-range residuals, saturating coefficients, the
17-bit / 18-bit / 13-bit register limits)."""
import numpy as np
import pytest

import rtl_stage_gm as R
from test_oracle_units import dequant, fdct, idct, quant


def _tiles(rng, n):
    out = []
    for k in range(n):
        kind = k % 6
        if kind == 0:
            t = rng.integers(-255, 256, (8, 8))
        elif kind == 1:
            t = rng.integers(-20, 21, (8, 8))
        elif kind == 2:
            t = np.full((8, 8), int(rng.choice([-255, 255])))                      # DC at the register limit
        elif kind == 3:
            t = np.where((np.add.outer(np.arange(8), np.arange(8)) & 1) == 0, 255, -255)   # highest frequency, full swing
        elif kind == 4:
            t = np.outer(rng.choice([-255, 255], 8), rng.choice([-1, 1], 8))
        else:
            t = (rng.integers(0, 2, (8, 8)) * 510 - 255)
        out.append(t.astype(np.int64))
    return out


def test_forward_dct_and_quantiser_match_the_register_model():
    rng = np.random.default_rng(77)
    for t in _tiles(rng, 120):
        want = np.array(R.forward_dct(t.tolist())).reshape(-1)
        got = fdct(t.reshape(-1))
        assert np.array_equal(got, want)
        for inter in (0, 1):
            for Q in (1, 2, 3, 4):
                wq = np.array(R.quantise(want.reshape(8, 8).tolist(), inter, Q)).reshape(-1)
                assert np.array_equal(quant(want, inter, Q), wq), (inter, Q)


def test_quantiser_at_the_17_bit_input_limits():
    """coefficients the DCT cannot produce but the 17-bit register can hold: the 16-bit |x| and the +2 wrap"""
    vals = [0, 1, -1, 2047 * 16, -(2047 * 16), 32767, -32768, 65535, -65535, -65536, 65534, 40000, -40000]
    rng = np.random.default_rng(5)
    for _ in range(40):
        c = rng.choice(vals, 64).astype(np.int64)
        c[rng.integers(0, 64, 8)] = rng.integers(-65536, 65536, 8)
        for inter in (0, 1):
            for Q in (1, 2, 3, 4):
                want = np.array(R.quantise(c.reshape(8, 8).tolist(), inter, Q)).reshape(-1)
                assert np.array_equal(quant(c, inter, Q), want), (inter, Q)


def test_inverse_quantiser_matches_the_register_model():
    rng = np.random.default_rng(78)
    for k in range(200):
        q = rng.integers(-2047, 2048, 64) if k % 2 else rng.choice([-2047, -1024, -1, 0, 1, 700, 2047], 64)
        for inter in (0, 1):
            for Q in (1, 2, 3, 4):
                want = np.array(R.dequantise(np.asarray(q).reshape(8, 8).tolist(), inter, Q)).reshape(-1)
                assert np.array_equal(dequant(q, inter, Q), want), (inter, Q)


def test_inverse_dct_matches_the_register_model():
    rng = np.random.default_rng(79)
    for k in range(150):
        kind = k % 5
        if kind == 0:
            d = rng.integers(-2047, 2048, 64)
        elif kind == 1:
            d = np.zeros(64, np.int64)
            d[rng.integers(0, 64, 3)] = rng.choice([-2047, 2047, -4096, 4095], 3)      # 13-bit limits: the 18-bit row store wraps
        elif kind == 2:
            d = rng.choice([-4096, 4095], 64)
        elif kind == 3:
            d = rng.integers(-40, 41, 64)
        else:
            d = np.zeros(64, np.int64)
            d[0] = int(rng.integers(-4096, 4096))
        want = np.array(R.inverse_dct(np.asarray(d).reshape(8, 8).tolist())).reshape(-1)
        assert np.array_equal(idct(d), want), kind


def test_add_clip():
    for a in (0, 1, 127, 128, 254, 255):
        for b in (-256, -255, -128, -1, 0, 1, 127, 255):
            v = a + R.wrap(b, 9, True)
            assert R.add_clip_0_255(a, b) == min(255, max(0, v))
