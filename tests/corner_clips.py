"""Hand-built clips for corners of the RTL's arithmetic that seeded content does not reach (found by the mutation-kill matrix,
tests/test_oracle_mutants.py: each of these was needed to catch one deliberate mis-reading).  Used three ways: by the second
restatements of the RTL against the oracle (CPU), by the mutation test, and by the HIP path against the oracle (-m gpu,
tests/test_gpu_corner_content.py).  Every function returns (clip [n, 3, H, W] uint8, pframes, VECTOR_LEVEL, Q_LEVEL)."""
import numpy as np

import m2v_load
from oracle import m2v_oracle_ctypes as orc

M = m2v_load.load()


def lone_level_after_31_zeros():
    """A level of +-1 after exactly 31 zeros - the last run that still has a table code (RTL:2525-2547); one more zero and it
    is an escape.  An intra picture whose only AC energy is the basis function at zig-zag position 32, its amplitude swept
    until the quantiser (Q_LEVEL 4) leaves exactly that one level."""
    L = orc.lib()
    i32, j32 = [(i, j) for i in range(8) for j in range(8) if L.m2v_oracle_tab_zigzag(i, j) == 32][0]
    W = H = 64
    yy, xx = np.mgrid[0:H, 0:W]
    basis = np.cos((2 * (yy % 8) + 1) * i32 * np.pi / 16) * np.cos((2 * (xx % 8) + 1) * j32 * np.pi / 16)
    for amp in range(4, 127):
        clip = np.zeros((1, 3, H, W), np.uint8)
        clip[0, 0] = np.clip(128 + amp * basis, 0, 255).astype(np.uint8)
        clip[0, 1:] = 128
        _, d = orc.encode(clip, 4, 4, 0, 7, 7, 1, 4, dump=True)
        lv = d["coef"][0][0][0]
        if abs(int(lv[32])) == 1 and not lv[1:32].any():
            return clip, 0, 1, 4
    raise AssertionError("no amplitude leaves a lone +-1 at zig-zag position 32")


def intra_between_inter_macroblocks():
    """A P picture whose slices alternate intra and inter macroblocks: the DC predictors an intra macroblock leaves behind are
    reset by the inter macroblock after it (RTL:2786-2792), the motion vector predictors an inter macroblock leaves behind by
    the intra macroblock after it (RTL:2771-2774).  Frame 0 is a soft ramp; in frame 1 every other macroblock is replaced by
    strong noise (nothing in the reference predicts it: intra) and the ones between move with the ramp (inter, non-zero vectors)."""
    rng = np.random.default_rng(14)
    W = H = 64
    yy, xx = np.mgrid[0:H, 0:W]
    clip = np.zeros((2, 3, H, W), np.uint8)
    clip[0, 0] = (60 + 2 * xx + yy).astype(np.uint8)
    clip[1, 0] = (60 + 2 * (xx + 3) + (yy + 1)).clip(0, 255).astype(np.uint8)       # the same ramp, displaced
    clip[:, 1] = 100
    clip[:, 2] = 160
    for by in range(4):
        for bx in range(by & 1, 4, 2):
            clip[1, 0, 16 * by:16 * by + 16, 16 * bx:16 * bx + 16] = rng.choice([10, 245], (16, 16))
            clip[1, 1, 16 * by:16 * by + 16, 16 * bx:16 * bx + 16] = rng.integers(0, 256, (16, 16))
    _, d = orc.encode(clip, 4, 4, 1, 7, 7, 3, 2, dump=True)
    inter = d["mb_inter"][1]
    assert inter.any() and not inter.all(), "the picture must mix the two kinds"
    return clip, 1, 3, 2


def vector_delta_of_sixteen():
    """Motion vector differences wrap into [-16, 15] (f_code 1, RTL:2736-2748): a difference of exactly +16 is sent as -16.  Two
    neighbouring macroblocks that move 4 pixels in opposite directions (vectors -8 and +8 half samples) make that difference;
    seeds are tried until the oracle's own vectors contain it."""
    W, H = 96, 64
    for seed in range(40):
        rng = np.random.default_rng(2100 + seed)
        coarse = rng.integers(30, 226, (H // 4 + 4, W // 4 + 8))
        tex = np.kron(coarse, np.ones((4, 4), np.int64))                       # 4x4 patches: full-pel matches are unambiguous
        clip = np.zeros((2, 3, H, W), np.uint8)
        clip[:, 1:] = 128
        clip[0, 0] = tex[8:8 + H, 16:16 + W]
        for bx in range(W // 16):
            sh = 4 if bx & 1 else -4
            clip[1, 0, :, 16 * bx:16 * bx + 16] = tex[8:8 + H, 16 + 16 * bx + sh:16 + 16 * bx + sh + 16]
        _, d = orc.encode(clip, W // 16, H // 16, 1, 7, 7, 3, 1, dump=True)
        inter = d["mb_inter"][1].reshape(H // 16, W // 16).astype(bool)
        mvx = d["mb_mvx"][1].reshape(H // 16, W // 16).astype(int)
        pairs = inter[:, 1:] & inter[:, :-1]
        if (np.abs(mvx[:, 1:] - mvx[:, :-1])[pairs] == 16).any():
            return clip, 1, 3, 1
    raise AssertionError("no seed produces a vector difference of 16")


def stream_ending_on_a_word_boundary():
    """The final-word rule (RTL:2932-2937): the residual ALWAYS leaves as one more zero-padded 32-byte word, also when there is
    none - a stream whose sequence_end_code ends exactly on a 256-bit boundary gets a whole word of zeros.  One seeded clip in
    32 ends like that; the first one found."""
    for ci in range(400):
        clip = M.synth.clip(64, 64, 1, clip_index=1000 + ci)
        data = orc.encode(clip, 4, 4, 0, 7, 7, 1, 2)
        if (data.rfind(b"\x00\x00\x01\xb7") + 4) % 32 == 0:
            return clip, 0, 1, 2
    raise AssertionError("no clip ends on a word boundary")


def sad_threshold_flat(W, H, total):
    """Two frames; the first is black (its reconstruction is exactly 0 / 128 / 128: tests/golden/kat_black), the second has
    `total` as the pixel sum of every macroblock, the last of it in the macroblock's last column: EVERY full-pel candidate of
    every macroblock then has a SAD of exactly `total`, reached only with the last column of the 16-clock accumulation.
    total = 4095 is the largest SAD that keeps a candidate alive, 4096 the smallest that kills it (RTL:1669-1670)."""
    clip = np.zeros((2, 3, H, W), np.uint8)
    clip[:, 1:] = 128
    mb = np.zeros((16, 16), np.int64)
    flat = mb.reshape(-1)
    left = total - 15                       # 15 sits in the last column, the rest is spread from the top left
    k = 0
    while left > 0:
        if k % 16 != 15:
            flat[k] = min(255, left)
            left -= flat[k]
        k += 1
    mb[15, 15] = 15
    assert mb.sum() == total and mb[:, :15].sum() < total
    clip[1, 0] = np.tile(mb, (H // 16, W // 16)).astype(np.uint8)
    return clip


def sad_exact(target):
    """Two frames 64x64, VECTOR_LEVEL 1 (+-2).  The interior macroblock (1, 1) of the second frame is built against the oracle's
    own reconstruction of the first so that its BEST full-pel candidate - not the one at the origin - has a SAD of exactly
    `target`, while the half-pel position next to it matches clearly better: a candidate that survives (target <= 4095) wins
    with its half-pel refinement, one that is killed (target >= 4096, RTL:1669-1670) leaves the vector at the origin.
    The one construction where the kill threshold itself decides the stream.
    How: the first frame is flat (60) in rows 8..23 and a ramp (5 per column, 3 per row) below; the macroblock's lower half is
    the half-pel sample between two reference blocks displaced by (-1, 0) and (-1, 1) (small SAD there, large elsewhere), its
    upper half is flat like the reference except for `nl` pixels lifted to 255: under those every candidate sees the same flat
    60, so they add the same mass to EVERY full- and half-pel SAD; one of them is then lowered until the minimum is the target."""
    W = H = 64
    by = bx = 1
    dy0, dx0 = -1, 0
    yy, xx = np.mgrid[0:H, 0:W]
    f0 = np.zeros((1, 3, H, W), np.uint8)
    f0[0, 0] = np.where((yy >= 8) & (yy < 24), 60, np.clip(20 + 5 * (xx - 10) + 3 * (yy - 10), 0, 200)).astype(np.uint8)
    f0[0, 1:] = 128
    _, d = orc.encode(f0, 4, 4, 0, 7, 7, 1, 2, dump=True)
    rec = d["recon"][0][:W * H].reshape(H, W).astype(np.int64)
    assert (rec[8:24] == 60).all(), "flat blocks reconstruct exactly"
    blk = lambda dy, dx: rec[16 * by + dy:16 * by + dy + 16, 16 * bx + dx:16 * bx + dx + 16]      # noqa: E731
    cands = [(dy, dx) for dy in range(-2, 3) for dx in range(-2, 3)]
    for nl in range(8, 64):
        cur = ((blk(dy0, dx0) + blk(dy0, dx0 + 1) + 1) >> 1).copy()
        cur.reshape(-1)[:nl] = 255                            # macroblock rows 0..3: reference rows -2..5 of any candidate, all flat
        sad = {c: int(np.abs(cur - blk(*c)).sum()) for c in cands}
        best = min(sad, key=lambda c: sad[c])
        lower = sad[best] - target                            # pixel (0, 0) comes down by this much
        if best[0] != dy0 or lower < 0 or 255 - lower < 61:
            continue
        cur[0, 0] = 255 - lower
        sad = {c: int(np.abs(cur - blk(*c)).sum()) for c in cands}
        hp = min(int(np.abs(cur - ((blk(*best) + blk(best[0], best[1] + sx) + 1) >> 1)).sum()) for sx in (-1, 1))
        assert min(sad.values()) == target and sorted(sad.values())[1] > target + 50 and hp < target - 100, (sorted(sad.values())[:3], hp)
        clip = np.concatenate([f0, f0])
        clip[1, 0, 16 * by:16 * by + 16, 16 * bx:16 * bx + 16] = cur.astype(np.uint8)
        return clip
    raise AssertionError("no feasible construction")
