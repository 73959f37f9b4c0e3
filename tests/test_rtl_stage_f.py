"""The oracle's motion estimation / prediction against a register-level emulation of RTL stage F
(tests/rtl_stage_f.py): windows are cut out of the oracle's previous reconstruction with RANDOM GARBAGE wherever the
frame ends (the RTL holds stale / uninitialised data there), so the oracle's "never selectable" derivations are tested,
not assumed."""
import numpy as np
import pytest

import m2v_load
from oracle import m2v_oracle_ctypes as orc
import corner_clips as C
from rtl_stage_f import stage_f

M = m2v_load.load()
L = orc.lib()


def _planes(rec, W, H):
    Y = rec[:W * H].reshape(H, W)
    U = rec[W * H:W * H + W * H // 4].reshape(H // 2, W // 2)
    V = rec[W * H + W * H // 4:].reshape(H // 2, W // 2)
    return Y, U, V


def _window(plane, y0, x0, h, w, rng):
    out = rng.integers(0, 256, (h, w)).astype(np.int64)          # garbage outside the frame
    H, W = plane.shape
    ys, xs = np.arange(y0, y0 + h), np.arange(x0, x0 + w)
    vy, vx = (ys >= 0) & (ys < H), (xs >= 0) & (xs < W)
    out[np.ix_(vy, vx)] = plane[np.ix_(ys[vy], xs[vx])]
    return out


def _residual(levels_zz, inter, Q):
    """dequantise + IDCT the dumped levels with the oracle's unit functions -> 8x8 residual"""
    zz = np.array([L.m2v_oracle_tab_zigzag(i, j) for i in range(8) for j in range(8)])
    q = np.ascontiguousarray(levels_zz[zz], np.int16)
    d = np.zeros(64, np.int16)
    r = np.zeros(64, np.int16)
    L.m2v_oracle_dequant(q.ctypes.data, int(inter), Q, d.ctypes.data)
    L.m2v_oracle_idct(d.ctypes.data, r.ctypes.data)
    return r.reshape(8, 8).astype(np.int64)


@pytest.mark.parametrize("W,H,n,pf,VL,Q,ci,kind", [(96, 80, 3, 2, 3, 2, 96, "clip"), (64, 96, 3, 2, 2, 1, 97, "clip"),
                                                    (80, 64, 3, 2, 1, 3, 98, "clip"), (64, 64, 2, 1, 3, 2, 0, "noise"),
                                                    (64, 64, 3, 2, 3, 2, 0, "checker"),
                                                    (64, 64, 2, 1, 3, 2, 4095, "sad"), (64, 64, 2, 1, 2, 2, 4096, "sad"), (64, 64, 2, 1, 1, 2, 4094, "sad"),
                                                    (64, 64, 2, 1, 1, 2, 4095, "exact"), (64, 64, 2, 1, 1, 2, 4096, "exact")])
def test_stage_f_emulation_equals_oracle(W, H, n, pf, VL, Q, ci, kind):
    clip = M.synth.clip(W, H, n, clip_index=ci, scene_len=7) if kind == "clip" else \
        C.sad_threshold_flat(W, H, ci) if kind == "sad" else C.sad_exact(ci) if kind == "exact" else M.synth.degenerate(kind, W, H, n)
    _, d = orc.encode(clip, W // 16, H // 16, pf, 7, 7, VL, Q, dump=True)
    rng = np.random.default_rng(ci + 1)
    UR, YR = VL, 2 * VL
    mbw, mbh = W // 16, H // 16
    checked_inter = 0
    for f in range(1, n):
        if f % (pf + 1) == 0:
            continue
        cY, cU, cV = _planes(d["yuv420"][f], W, H)
        rY, rU, rV = _planes(d["recon"][f - 1], W, H)
        oY, oU, oV = _planes(d["recon"][f], W, H)
        for by in range(mbh):
            for bx in range(mbw):
                mb = by * mbw + bx
                Yref = _window(rY, 16 * by - YR, 16 * bx - YR, 16 + 2 * YR, 32 + YR, rng)
                Uref = _window(rU, 8 * by - UR, 8 * bx - UR, 8 + 2 * UR, 16 + UR, rng)
                Vref = _window(rV, 8 * by - UR, 8 * bx - UR, 8 + 2 * UR, 16 + UR, rng)
                inter, mvx, mvy, Yp, Up, Vp, _ = stage_f(cY[16 * by:16 * by + 16, 16 * bx:16 * bx + 16], Yref, Uref, Vref,
                                                         bx, by, mbw - 1, mbh - 1, f % (pf + 1), VL)
                assert inter == bool(d["mb_inter"][f][mb]), (f, bx, by)
                if inter:
                    assert (mvx, mvy) == (d["mb_mvx"][f][mb], d["mb_mvy"][f][mb]), (f, bx, by)
                    checked_inter += 1
                # prediction, through the reconstruction: recon = clip(pred + residual)
                lv = d["coef"][f][mb].astype(np.int16)
                for t in range(4):
                    oy, ox = 8 * (t >> 1), 8 * (t & 1)
                    want = oY[16 * by + oy:16 * by + oy + 8, 16 * bx + ox:16 * bx + ox + 8]
                    got = np.clip(Yp[oy:oy + 8, ox:ox + 8] + _residual(lv[t], inter, Q), 0, 255)
                    assert np.array_equal(got, want), (f, bx, by, t)
                assert np.array_equal(np.clip(Up + _residual(lv[4], inter, Q), 0, 255), oU[8 * by:8 * by + 8, 8 * bx:8 * bx + 8])
                assert np.array_equal(np.clip(Vp + _residual(lv[5], inter, Q), 0, 255), oV[8 * by:8 * by + 8, 8 * bx:8 * bx + 8])
    if kind == "clip":
        assert checked_inter > 10
