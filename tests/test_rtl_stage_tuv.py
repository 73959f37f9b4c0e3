"""The oracle's bytes against a clock-by-clock model of RTL stages T/U/V (tests/rtl_stage_tuv.py) fed with the
oracle's own per-macroblock decisions: two structurally different restatements of the entropy coder must agree."""
import numpy as np
import pytest

import m2v_load
from oracle import m2v_oracle_ctypes as orc
from rtl_stage_tuv import StageTUV

M = m2v_load.load()


def run_model(W, H, Q, pf, d, n):
    mbw, mbh = W // 16, H // 16
    m = StageTUV(W, H, Q)
    m.sequence_start()
    for f in range(n):
        i_frame = f % (pf + 1)
        for y16 in range(mbh):
            for x16 in range(mbw):
                mb = y16 * mbw + x16
                m.macroblock(i_frame, x16, y16, bool(d["mb_inter"][f][mb]), int(d["mb_mvx"][f][mb]), int(d["mb_mvy"][f][mb]),
                             int(d["mb_cbp"][f][mb]), d["coef"][f][mb].astype(int).tolist())
    return m.sequence_end()


@pytest.mark.parametrize("kind,W,H,n,pf,VL,Q", [("clip", 64, 64, 4, 3, 3, 2), ("clip", 96, 64, 3, 1, 1, 1), ("clip", 64, 80, 3, 2, 2, 4),
                                                 ("noise", 64, 64, 2, 1, 3, 3), ("checker", 64, 64, 2, 1, 3, 2), ("gray", 64, 64, 26, 24, 1, 2)])
def test_clocked_model_equals_oracle(kind, W, H, n, pf, VL, Q):
    clip = M.synth.clip(W, H, n, clip_index=95, scene_len=2) if kind == "clip" else M.synth.degenerate(kind, W, H, n)
    data, d = orc.encode(clip, W // 16, H // 16, pf, 7, 7, VL, Q, dump=True)
    assert run_model(W, H, Q, pf, d, n) == data


@pytest.mark.parametrize("make", ["lone_level_after_31_zeros", "stream_ending_on_a_word_boundary", "intra_between_inter_macroblocks",
                                  "vector_delta_of_sixteen"])
def test_clocked_model_equals_oracle_on_the_corner_clips(make):
    """content built for one corner of the entropy coder each (tests/corner_clips.py): the last run with a table code, a stream
    that ends on a word boundary, DC / vector predictors across alternating intra and inter macroblocks, a vector difference
    at the wrap"""
    import corner_clips as C
    clip, pf, VL, Q = getattr(C, make)()
    n, _, H, W = clip.shape
    data, d = orc.encode(clip, W // 16, H // 16, pf, 7, 7, VL, Q, dump=True)
    assert run_model(W, H, Q, pf, d, n) == data
