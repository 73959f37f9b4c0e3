"""-m gpu: seeded randomized differential test, HIP path vs oracle, over geometry x VECTOR_LEVEL x Q_LEVEL x pframes x
chunking x content (smooth, noisy, flat, full-range, dark, mixtures).  Byte-level comparison; on a mismatch the stage-level
comparison of tests/gpu_util.py says where it starts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_content(rng, M, W, H, n):
    kind = rng.integers(0, 7)
    if kind == 0:
        return M.synth.clip(W, H, n, clip_index=int(rng.integers(0, 1 << 20)), scene_len=int(rng.integers(2, 9)))
    if kind == 1:
        return rng.integers(0, 256, (n, 3, H, W), dtype=np.uint8)                       # white noise: escapes, SAD overflow
    if kind == 2:
        return np.full((n, 3, H, W), int(rng.integers(0, 256)), np.uint8)               # flat: everything zero / skipped
    if kind == 3:                                                                       # full-range blocks moving by odd offsets
        base = rng.integers(0, 2, (H // 4 + 8, W // 4 + 8)).astype(np.uint8) * 255
        out = np.empty((n, 3, H, W), np.uint8)
        for f in range(n):
            oy, ox = rng.integers(0, 8, 2)
            big = np.kron(base, np.ones((4, 4), np.uint8))
            out[f, :] = big[oy:oy + H, ox:ox + W]
        return out
    if kind == 4:                                                                       # dark, low contrast (un-saturated intra cost)
        return rng.integers(0, 12, (n, 3, H, W), dtype=np.uint8)
    if kind == 5:                                                                       # smooth gradients translating by half pels
        y, x = np.mgrid[0:H, 0:W]
        out = np.empty((n, 3, H, W), np.uint8)
        for f in range(n):
            out[f, 0] = ((x * 3 + y * 2 + f * 3) // 2) & 255
            out[f, 1] = ((x + f) * 2) & 255
            out[f, 2] = ((y * 5 + f * 7) // 3) & 255
        return out
    a = M.synth.clip(W, H, n, clip_index=int(rng.integers(0, 1 << 20)))                 # mixture: noise in one quadrant
    a[:, :, :H // 2, :W // 2] = rng.integers(0, 256, (n, 3, H // 2, W // 2), dtype=np.uint8)
    return a


import os


@pytest.mark.parametrize("seed", range(int(os.environ.get("M2V_FUZZ_SEEDS", "6"))))
def test_fuzz(seed):
    import gpu_util as G
    from oracle import m2v_oracle_ctypes as orc
    rng = np.random.default_rng(1000 + seed)
    for case in range(7):
        W, H = 16 * int(rng.integers(4, 17)), 16 * int(rng.integers(4, 13))
        VL, Q = int(rng.integers(1, 4)), int(rng.integers(1, 5))
        pf = int(rng.choice([0, 1, 2, 3, 5, 8, 255]))
        n = int(rng.integers(1, 8))
        bf = int(rng.choice([1, 2, 3, 96]))
        clip = make_content(rng, G.M, W, H, n)
        nbeats = None
        if rng.integers(0, 3) == 0:                                                    # stop somewhere inside the last frame
            nbeats = (n - 1) * (W * H // 4) + int(rng.integers(1, W * H // 4))
        want = orc.encode(clip, W // 16, H // 16, pf, 7, 7, VL, Q, nbeats=nbeats)
        tag = "seed %d case %d: %dx%d n=%d pf=%d VL=%d Q=%d batch=%d nbeats=%s" % (seed, case, W, H, n, pf, VL, Q, bf, nbeats)
        if nbeats is None:
            got = G.resident_encode(clip, W // 16, H // 16, pf, 7, 7, VL, Q, batch_frames=bf)
            if got != want:
                pytest.fail(tag + "\n" + "\n".join(G.compare_stages(clip, W // 16, H // 16, pf, 7, 7, VL, Q, batch_frames=bf)))
        enc = G.M.Mpeg2Encoder(7, 7, VL, Q)
        try:
            enc.set_option("batch_frames", bf)
            assert enc.encode(clip, W // 16, H // 16, pf, nbeats=nbeats) == want, tag + " (port interface)"
            if case % 3 == 2:                                                          # the non-reference ISO reconstruction loop
                enc.set_option("conformant", 1)
                want_c = orc.encode(clip, W // 16, H // 16, pf, 7, 7, VL, Q, nbeats=nbeats, conformant=True)
                assert enc.encode(clip, W // 16, H // 16, pf, nbeats=nbeats) == want_c, tag + " (conformant)"
        finally:
            enc.close()
