"""Clock-by-clock emulation of stages A, B, C of RTL/mpeg2encoder.v (RTL:1027-1171): the sequence FSM with its beat
counters, i_sequence_stop and black fill, the horizontal chroma mean, the line buffer (written and read in the same
clock: the read returns the OLD entry = the row above), and the vertical mean that is kept on odd rows only.
Checked against the oracle's 4:2:0 planes, including a stop inside a frame and bubbles between beats."""
import numpy as np
import pytest

import m2v_load
from oracle import m2v_oracle_ctypes as orc

M = m2v_load.load()
IDLE, DURING, ENDING, ENDED = range(4)


def mean2(a, b):
    return (a + b + 1) >> 1


def emulate(beats, stop_after, xsize16, ysize16, XL, YL, bubbles):
    """beats: list of (Y[4], U[4], V[4]); the stop pulse comes with i_en = 0 after beat index `stop_after`.
    Returns per frame the c-stage outputs: Y rows and the U/V samples stored on odd rows."""
    lim_x, lim_y = 1 << XL, 1 << YL
    i_max_x16 = lim_x - 1 if xsize16 > lim_x else 3 if xsize16 < 4 else xsize16 - 1      # RTL:985-987
    i_max_y16 = lim_y - 1 if ysize16 > lim_y else 3 if ysize16 < 4 else ysize16 - 1
    state, max_x4, max_y = IDLE, 0, 0
    a_x4 = a_y = 0
    a_en, a_Y, a_U, a_V = 0, [0] * 4, [0x80] * 2, [0x80] * 2
    lbuf_U, lbuf_V = {}, {}
    b = dict(en=0, x4=0, y=0, Y=[0] * 4, U=[0, 0], V=[0, 0], Uu=[0, 0], Vu=[0, 0])
    frames, cur = [], None
    rng = np.random.default_rng(0)
    k = 0
    pending_stop = False
    clocks = 0
    while True:
        clocks += 1
        assert clocks < 10_000_000
        # inputs of this clock
        i_en = i_stop = 0
        if k < len(beats):
            if not (bubbles and rng.integers(0, 3) == 0):
                i_en = 1
                iY, iU, iV = beats[k]
                k += 1
                if k - 1 == stop_after:
                    pending_stop = True
        elif pending_stop:
            i_stop, pending_stop = 1, False
        # ---- stage C from stage B (RTL:1152-1171) ----
        if b["en"]:
            if b["x4"] == 0 and b["y"] == 0:
                cur = dict(Y={}, U={}, V={})
                frames.append(cur)
            for p in range(4):
                cur["Y"][(b["y"], 4 * b["x4"] + p)] = b["Y"][p]
            if b["y"] & 1:                                   # only valid / stored when c_y is odd (RTL:1150, 1211)
                for p in range(2):
                    cur["U"][(b["y"] >> 1, 2 * b["x4"] + p)] = mean2(b["U"][p], b["Uu"][p])
                    cur["V"][(b["y"] >> 1, 2 * b["x4"] + p)] = mean2(b["V"][p], b["Vu"][p])
        # ---- stage B from stage A; line buffer read (old entry) then write (RTL:1116-1143) ----
        nb = dict(en=a_en, x4=a_x4, y=a_y, Y=list(a_Y), U=list(a_U), V=list(a_V),
                  Uu=list(lbuf_U.get(a_x4, [0, 0])), Vu=list(lbuf_V.get(a_x4, [0, 0])))
        if a_en:
            lbuf_U[a_x4], lbuf_V[a_x4] = list(a_U), list(a_V)
        b = nb
        # ---- stage A (RTL:1040-1093) ----
        n_en, nY, nU, nV = 0, [0] * 4, [0x80] * 2, [0x80] * 2
        if state == ENDED:
            break                                            # (the RTL waits for o_last here)
        elif state == ENDING:
            if a_x4 < max_x4:
                a_x4 += 1
                n_en = 1
            elif a_y < max_y:
                a_x4 = 0
                a_y += 1
                n_en = 1
            else:
                state = ENDED
        elif i_en:
            if state == IDLE:
                state = DURING
                max_x4, max_y = 4 * (i_max_x16 + 1) - 1, 16 * (i_max_y16 + 1) - 1
                a_x4 = a_y = 0
            else:
                if a_x4 < max_x4:
                    a_x4 += 1
                else:
                    a_x4 = 0
                    a_y = a_y + 1 if a_y < max_y else 0
            if i_stop:
                state = ENDING
            n_en = 1
            nY = list(iY)
            nU = [mean2(iU[0], iU[1]), mean2(iU[2], iU[3])]
            nV = [mean2(iV[0], iV[1]), mean2(iV[2], iV[3])]
        elif i_stop and state == DURING:
            state = ENDING
        a_en, a_Y, a_U, a_V = n_en, nY, nU, nV
    # drain the two pipeline stages
    return frames, (4 * (max_x4 + 1), max_y + 1)


@pytest.mark.parametrize("W,H,nbeats,bubbles", [(64, 64, 64 * 64 // 4 * 2, False), (96, 64, 96 * 64 // 4 + 777, True),
                                                (64, 80, 64 * 80 // 4 * 2 + 1, False)])
def test_stage_abc(W, H, nbeats, bubbles):
    bpf = W * H // 4
    nframes = (nbeats + bpf - 1) // bpf
    clip = M.synth.clip(W, H, nframes, clip_index=99)
    beats = []
    for f in range(nframes):
        y, u, v = clip[f, 0].reshape(-1, 4), clip[f, 1].reshape(-1, 4), clip[f, 2].reshape(-1, 4)
        beats += [(y[i].tolist(), u[i].tolist(), v[i].tolist()) for i in range(bpf)]
    beats = beats[:nbeats]
    frames, (Wc, Hc) = emulate(beats, nbeats - 1, W // 16, H // 16, 7, 7, bubbles)
    assert (Wc, Hc) == (W, H)
    _, d = orc.encode(clip, W // 16, H // 16, 1, 7, 7, 1, 2, nbeats=nbeats, dump=True)
    assert len(frames) == nframes
    for f in range(nframes):
        want = d["yuv420"][f]
        Y = want[:W * H].reshape(H, W)
        U = want[W * H:W * H + W * H // 4].reshape(H // 2, W // 2)
        V = want[W * H + W * H // 4:].reshape(H // 2, W // 2)
        got = frames[f]
        assert len(got["Y"]) == W * H and len(got["U"]) == W * H // 4
        assert all(Y[k] == v for k, v in got["Y"].items())
        assert all(U[k] == v for k, v in got["U"].items())
        assert all(V[k] == v for k, v in got["V"].items())
