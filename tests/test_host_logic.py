"""Host-side arithmetic and scheduling facts the HIP kernels rely on (no GPU needed)."""
import os

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


def test_keyed_minimum_is_the_rtl_decision_tree():
    """k_mb's half-pel decision takes the minimum of ten KEYS (cost << 16 | rank) instead of walking find_min_in_10_values' tree
    (RTL:804-840): the tree compares pairs with "<", the halves 0-3 / 4-7 so that 4-7 win a tie, and 8-9 with "<=" against both - i.e. among
    equal costs candidate 8 wins, then 9, 4, 5, 6, 7, 0, 1, 2, 3 (csrc/m2v_kernels.hpp kHpRank).  Checked against the tree (the
    second reading's, tests/rtl_stage_f.py) over EVERY pattern of ties and orders: all 10-tuples over four cost levels (4^10 ~ 1 M),
    vectorised; plus the two facts the kernel leans on - a cost the RTL caps at 4096 never wins because the intra cost (index 9) is at
    most 4095, so leaving it uncapped / marking a dead candidate by bit 12 changes nothing - over random costs with caps and dead bits."""
    import itertools
    from rtl_stage_f import find_min_in_10_values
    rank = np.array([6, 7, 8, 9, 2, 3, 4, 5, 0, 1])
    # the tree, vectorised (same comparisons as rtl_stage_f.find_min_in_10_values; that one is the scalar cross-check below)
    def tree(v):
        wi1 = v[:, 1] < v[:, 0]; w01 = np.where(wi1, v[:, 1], v[:, 0])
        wi3 = v[:, 3] < v[:, 2]; w23 = np.where(wi3, v[:, 3], v[:, 2])
        wi5 = v[:, 5] < v[:, 4]; w45 = np.where(wi5, v[:, 5], v[:, 4])
        wi7 = v[:, 7] < v[:, 6]; w67 = np.where(wi7, v[:, 7], v[:, 6])
        wi9 = v[:, 9] < v[:, 8]; w89 = np.where(wi9, v[:, 9], v[:, 8])
        xi23 = w23 < w01; x0123 = np.where(xi23, w23, w01)
        xi67 = w67 < w45; x4567 = np.where(xi67, w67, w45)
        a = np.where(xi23, 2 + wi3, wi1.astype(np.int64))
        b = np.where(xi67, 6 + wi7, 4 + wi5)
        return np.where((w89 <= x0123) & (w89 <= x4567), 8 + wi9, np.where(x0123 < x4567, a, b))
    def keyed(v):
        return np.argmin((v.astype(np.int64) << 16) | rank[None, :], axis=1)
    levels = np.array(list(itertools.product(range(4), repeat=10)), dtype=np.int64)
    assert np.array_equal(tree(levels), keyed(levels))
    rng = np.random.default_rng(5)
    for row in levels[rng.integers(0, len(levels), 300)]:
        assert find_min_in_10_values(list(row)) == tree(row[None, :])[0]
    # the kernel's costs: nine totals < 2^16, capped at 4096 by the RTL, dead ones forced to 4096; the intra cost <= 4095
    tot = rng.integers(0, 1 << 16, size=(400000, 9))
    tot[:, :] = np.where(rng.random(tot.shape) < 0.5, rng.integers(0, 4200, size=tot.shape), tot)      # plenty of costs around the cap
    tot[:100000] = rng.integers(4090, 4100, size=(100000, 9))
    dead = rng.random(tot.shape) < 0.3
    intra = rng.integers(0, 4096, size=(tot.shape[0], 1))
    intra[:50000] = 4095
    rtl = np.concatenate([np.minimum(tot | (dead.astype(np.int64) << 12), 4096), intra], axis=1)
    mine = np.concatenate([tot | (dead.astype(np.int64) << 12), intra], axis=1)                          # uncapped, dead = bit 12 set
    assert np.array_equal(tree(rtl), keyed(mine))
    # (hy + 1, hx + 1) of the candidate of rank r, two bits each: the kernel's decode constants
    cand_of_rank = [8, 4, 4, 5, 6, 7, 0, 1, 2, 3]                                                        # rank 1 = intra: the centre, unused
    hy = sum((c // 3) << (2 * r) for r, c in enumerate(cand_of_rank))
    hx = sum((c % 3) << (2 * r) for r, c in enumerate(cand_of_rank))
    for k in range(9):
        r = int(rank[k])
        assert ((hy >> (2 * r)) & 3, (hx >> (2 * r)) & 3) == (k // 3, k % 3)
    src = open(os.path.join(os.path.dirname(__file__), "..", "fpga-mpeg2-encoder_amd", "csrc", "m2v_kernels.hpp")).read()
    assert "kHpRank[10] = {6, 7, 8, 9, 2, 3, 4, 5, 0, 1}" in src


def test_half_pel_dead_words_match_the_rtl_rule():
    """the table word per packed pair of half-pel totals (hp_dead_word, read by the scalar unit): bit 12 / bit 28 set exactly where the
    pair's low / high candidate reaches outside the frame or beyond the search range (RTL:1757-1760) - rebuilt here from the rule"""
    for F in range(16):
        no_l, no_r, no_u, no_d = F & 1, (F >> 1) & 1, (F >> 2) & 1, (F >> 3) & 1
        dead = []
        for k in range(9):
            hy, hx = k // 3 - 1, k % 3 - 1
            dead.append(int((hx < 0 and no_l) or (hx > 0 and no_r) or (hy < 0 and no_u) or (hy > 0 and no_d)))
        for pair in range(5):
            want = (dead[2 * pair] << 12) | ((dead[2 * pair + 1] << 28) if pair < 4 else 0)
            # hp_dead_bit / hp_dead_word of csrc/m2v_kernels.hpp, restated
            bit = lambda kk: int(bool(((kk % 3 == 0) and (F & 1)) or ((kk % 3 == 2) and (F & 2)) or ((kk // 3 == 0) and (F & 4)) or ((kk // 3 == 2) and (F & 8))))
            got = (bit(2 * pair) << 12) | ((bit(2 * pair + 1) << 28) if pair < 4 else 0)
            assert got == want
