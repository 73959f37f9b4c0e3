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


def test_mean4_lerp_identity():
    """k_mb computes the RTL's mean4 = (a+b+c+d+1)>>2 (RTL:760-767) on packed bytes as
    lerp(floor_avg(a,b), floor_avg(c,d), round = (a^b)|(c^d)) with v_lerp_u8; exhaustive over all byte values
    (the formula depends on (a,b) and (c,d) only through their sums)."""
    a = np.arange(256, dtype=np.int64)
    s1 = (a[:, None] + a[None, :]).reshape(-1)
    combos = np.unique(np.stack([s1 >> 1, s1 & 1, s1], 1), axis=0)      # floor avg, parity (= (a^b)&1), sum
    for p1, e1, t1 in combos:
        lhs = (p1 + combos[:, 0] + (e1 | combos[:, 1])) >> 1
        assert np.array_equal(lhs, (t1 + combos[:, 2] + 1) >> 2)


def test_signed_inter_quantiser_identity():
    """csrc/m2v_kernels.hpp quantises non-intra coefficients on the signed value: sign(C)*min((|C|+2)>>s, 2047) (RTL:2070)
    == clamp((C + 2 + (sg & (2^s - 5))) >> s) with sg = C >> 31, and the inverse quantiser's (2q + sign q) << Q
    (RTL:2134-2137) == (2q + (q != 0 ? sg | 1 : 0)) << Q - over more than the 17-bit range of the DCT output."""
    C = np.arange(-70000, 70001, dtype=np.int64)
    sg = C >> 63
    for Q in (1, 2, 3, 4):
        s = 4 + Q
        a = np.minimum((np.abs(C) + 2) >> s, 2047)
        q_ref = np.where(C < 0, -a, a)
        q_new = np.clip((C + 2 + (sg & ((1 << s) - 5))) >> s, -2047, 2047)
        assert np.array_equal(q_ref, q_new)
        x_ref = np.clip((2 * q_ref + np.sign(q_ref)) << Q, -2047, 2047)
        x_new = np.clip((2 * q_new + np.where(q_new != 0, sg | 1, 0)) << Q, -2047, 2047)
        assert np.array_equal(x_ref, x_new)


def test_inter_quantiser_never_reaches_its_clamp():
    """The kernel's non-intra quantiser works on the accumulator acc = sum + 2048 directly,
    q = (acc + (2 << 12) + (acc < 0 ? (2^s - 5) << 12 : 0)) >> (12 + s), without the min(.., 2047) of RTL:2070: over the
    whole reachable range |C| <= 16320 (test_dct_coefficient_bound) it equals the RTL's expression, the clamp never binds
    (|q| <= 510), and sign(q) = med3(q, -1, 1) feeds the inverse quantiser (RTL:2134-2137).  The shipped form carries the
    (2 << 12) inside the accumulator (t = acc + (2 << 12)) and takes the sign of t instead of the sign of acc."""
    acc = np.arange(-16321 * 4096, 16321 * 4096 + 1, 997, dtype=np.int64)      # every 997th accumulator value ...
    acc = np.concatenate([acc, np.arange(-70000, 70001, dtype=np.int64), np.array([-16320 * 4096, 16320 * 4096 + 4095])])
    C = acc >> 12
    for Q in (1, 2, 3, 4):
        s = 4 + Q
        a = np.minimum((np.abs(C) + 2) >> s, 2047)
        q_ref = np.where(C < 0, -a, a)
        q_new = (acc + (2 << 12) + np.where(acc < 0, ((1 << s) - 5) << 12, 0)) >> (12 + s)
        assert np.array_equal(q_ref, q_new)
        t = acc + (2 << 12)
        q_t = (t + (t >> 63) * -(((1 << s) - 5) << 12)) >> (12 + s)          # sign mask times MINUS the bias, plus t
        assert np.array_equal(q_ref, q_t)
        assert np.abs(q_new).max() <= 510
        x_ref = np.clip((2 * q_ref + np.sign(q_ref)) << Q, -2047, 2047)
        x_new = np.clip((2 * q_new + np.clip(q_new, -1, 1)) << Q, -2047, 2047)
        assert np.array_equal(x_ref, x_new)


def test_inverse_quantisers_never_reach_their_clamps():
    """The kernels drop three saturations of the RTL that no real input can reach (the oracle keeps them: it is also fed arbitrary
    levels by the unit tests):
    * non-intra inverse quantiser, RTL:2134-2137: |C| <= 16320 gives |q| <= 16322 >> (4 + Q), and (2 |q| + 1) << Q <= 2047;
    * intra quantiser, RTL:2075, and intra inverse quantiser, RTL:2139-2144 (17-bit product, +-2047): an intra block is
      pixel - 128, so |C| <= 8192 (the bound below), and over EVERY weight of the matrix, every Q_LEVEL and every |C| up to it the
      level stays below 2047, the product level * W below 2^16 and the shifted product within +-2047."""
    for Q in (1, 2, 3, 4):
        qmax = (16320 + 2) >> (4 + Q)
        assert ((2 * qmax + 1) << Q) <= 2047, Q
    D = np.array([M.lib().m2v_debug_table(0, i, j) for i in range(8) for j in range(8)]).reshape(8, 8)
    worst = 0
    for i in range(8):
        for j in range(8):
            x = 128 * np.sign(np.outer(D[i], D[j]))          # pixel - 128 lies in [-128, 127]: 128 bounds it
            worst = max(worst, (abs(int(D[i] @ x @ D[j])) + 2048) >> 12)
    assert worst <= 8193
    C = np.arange(0, worst + 1, dtype=np.int64)
    assert (((C + 8) >> 4) <= 2047).all()                    # DC, RTL:2074
    for w in sorted(set(INTRA_W) | set(range(8, 84))):
        for Q in (1, 2, 3, 4):
            off = (w * ((3 << Q) + 2)) >> 3
            q = ((C + off) >> Q) // w                        # RTL:2072
            assert q.max() <= 2047
            prod = q * w
            assert prod.max() < (1 << 16)                    # the 17-bit signed temporary of RTL:2093 cannot wrap
            x = prod << (Q - 3) if Q >= 3 else prod >> (3 - Q)
            assert x.max() < (1 << 16) and x.max() <= 2047, (w, Q)
            xn = (-prod) << (Q - 3) if Q >= 3 else (-prod) >> (3 - Q)       # negative levels: arithmetic shift floors (RTL:2143)
            assert xn.min() >= -2047
