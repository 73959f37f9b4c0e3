"""-m gpu: content built to drive every transform coefficient to its largest possible magnitude - each 8x8 tile is the sign pattern
of one DCT basis product at full swing (0 / 255), as an intra picture (pixel - 128) and as a P picture over the complementary
pattern (residual +-255) - for every Q_LEVEL.  The kernels drop three saturations of the RTL that their own quantiser cannot
reach (tests/test_host_logic.py proves the bounds); this is the same statement made with pixels: levels, reconstruction and
bytes equal the oracle's, which keeps every clamp."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def basis_sign_frames(W, H, seed):
    """[frame 0, frame 1]: tile (ty, tx) of frame 0 carries sign(outer(D[i], D[j])) with (i, j) cycling over all 64 products;
    frame 1 is its complement (plus a few tiles that repeat frame 0, so that some macroblocks stay inter with a zero residual)."""
    import m2v_load
    M = m2v_load.load()
    D = np.array([M.lib().m2v_debug_table(0, i, j) for i in range(8) for j in range(8)]).reshape(8, 8)
    rng = np.random.default_rng(seed)
    f0 = np.zeros((H, W), np.uint8)
    k = int(rng.integers(0, 64))
    for ty in range(H // 8):
        for tx in range(W // 8):
            i, j = divmod(k % 64, 8)
            k += 1
            s = np.sign(np.outer(D[i], D[j]))
            s[s == 0] = 1
            f0[8 * ty:8 * ty + 8, 8 * tx:8 * tx + 8] = np.where(s > 0, 255, 0)
    f1 = 255 - f0
    keep = rng.random((H // 16, W // 16)) < 0.2
    for by, bx in zip(*np.nonzero(keep)):
        f1[16 * by:16 * by + 16, 16 * bx:16 * bx + 16] = f0[16 * by:16 * by + 16, 16 * bx:16 * bx + 16]
    clip = np.zeros((2, 3, H, W), np.uint8)
    clip[0, 0], clip[1, 0] = f0, f1
    # chroma: the same idea at half resolution, through the 4:4:4 planes (every 2x2 block constant so that the 4:2:0 samples keep the swing)
    c0 = np.kron(f0[:H // 2, :W // 2], np.ones((2, 2), np.uint8))
    clip[0, 1], clip[0, 2] = c0, 255 - c0
    clip[1, 1], clip[1, 2] = 255 - c0, c0
    return clip


@pytest.mark.parametrize("Q", [1, 2, 3, 4])
@pytest.mark.parametrize("VL", [1, 3])
def test_full_swing_basis_patterns(Q, VL):
    import gpu_util as G
    clip = basis_sign_frames(128, 96, 10 * Q + VL)
    assert G.compare_stages(clip, 8, 6, 1, XL=7, YL=7, VL=VL, Q=Q) == []      # I + P
    assert G.compare_stages(clip, 8, 6, 0, XL=7, YL=7, VL=VL, Q=Q) == []      # both intra


def test_full_swing_through_the_integer_transform_path():
    """the same with the luma transform on the integer path (option dct_mfma = 0)"""
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    import gpu_util as G
    M = m2v_load.load()
    clip = basis_sign_frames(128, 96, 77)
    for Q in (1, 4):
        want = orc.encode(clip, 8, 6, 1, 7, 7, 3, Q)
        enc = M.Mpeg2Encoder(7, 7, 3, Q)
        try:
            enc.set_option("dct_mfma", 0)
            assert G.resident_encode(clip, 8, 6, 1, 7, 7, 3, Q, enc=enc) == want
        finally:
            enc.close()
