"""Host-side arithmetic and scheduling facts the HIP kernels rely on (no GPU needed)."""
import numpy as np

import m2v_load

M = m2v_load.load()

INTRA_W = [8, 16, 19, 22, 24, 26, 27, 29, 32, 34, 35, 37, 38, 40, 46, 48, 56, 58, 69, 83]


def test_multiply_shift_division_is_exact():
    """k_mb replaces the RTL's `/ INTRA_Q` (RTL:2072) by (n * ceil(2^21/W)) >> 21 in 32-bit arithmetic.
    n = (|C| + offset) >> Q_LEVEL with |C| <= 16320 (8x8 DCT of 9-bit residuals) and Q_LEVEL >= 1."""
    n = np.arange(0, 8600, dtype=np.uint64)
    for w in range(8, 84):
        m = ((1 << 21) + w - 1) // w
        assert int(n.max()) * m < (1 << 32)
        assert np.array_equal((n * m) >> 21, n // w), w
    for w in INTRA_W:
        for Q in (1, 2, 3, 4):
            off = (w * ((3 << Q) + 2)) >> 3
            assert ((16320 + off) >> Q) < 8600


def test_dct_coefficient_bound():
    """|C| <= 16320: the quantiser input fits the 16-bit g_t3 (RTL:1949) and the bound used above."""
    D = np.array([M.lib().m2v_debug_table(0, i, j) for i in range(8) for j in range(8)]).reshape(8, 8)
    worst = 0
    for i in range(8):
        for j in range(8):
            x = 255 * np.sign(np.outer(D[i], D[j]))          # the residual that maximises |C[i][j]|
            t = D[i] @ x @ D[j]
            worst = max(worst, (abs(int(t)) + 2048) >> 12)
    assert worst == 16320


def test_synth_is_deterministic_and_exercises_range():
    a = M.synth.clip(96, 64, 4, clip_index=5)
    b = M.synth.clip(96, 64, 4, clip_index=5)
    assert np.array_equal(a, b) and a.dtype == np.uint8 and a.shape == (4, 3, 64, 96)
    assert a.min() < 16 and a.max() > 200
    assert not np.array_equal(a, M.synth.clip(96, 64, 4, clip_index=6))
