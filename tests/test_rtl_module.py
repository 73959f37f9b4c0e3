"""CPU: the clock-by-clock model of the whole module (tests/rtl_module.py: every stage from the beat interface to the
256-bit words, real mem_lbuf / mem_dbuf / mem_ref / mem_delay arrays that start as random garbage, the prefetch and the
one-slice-late write-back at the RTL's own clocks) against oracle/m2v_oracle.c.

The oracle does not model those memories: it derives "the reference of frame f + 1 is the reconstruction of frame f"
(SURVEY.md 3.4).  Here that derivation is tested instead of trusted: P GOPs of three and more frames, a frame of exactly
four slices (the bottom-right block prefetches the NEXT frame's top-left block, RTL:1351-1353), a geometry narrower than
the memories' row stride, bubbles, a stop in the middle of a frame - and four deliberate deviations from the RTL's
timing that must each make the bytes differ, so that agreement means something.
Parity stays "unpinned by the reference": this is a second, structurally different READING of the RTL, not a run of it."""
import numpy as np
import pytest

import m2v_load
from oracle import m2v_oracle_ctypes as orc
import rtl_module

M = m2v_load.load()


def run(W, H, n, pf, XL, YL, VL, Q, ci, nbeats=None, bubbles=0, scene_len=4, seed=1, sabotage=None):
    clip = M.synth.clip(W, H, n, clip_index=ci, scene_len=scene_len)
    want = orc.encode(clip, W // 16, H // 16, pf, XL, YL, VL, Q, nbeats=nbeats)
    got, m = rtl_module.encode(clip, W // 16, H // 16, pf, XL, YL, VL, Q, nbeats=nbeats, bubbles=bubbles, seed=seed, sabotage=sabotage)
    return got, want, m


@pytest.mark.parametrize("W,H,n,pf,XL,YL,VL,Q,ci,kw", [
    (64, 64, 3, 2, 4, 4, 3, 2, 150, {}),                           # exactly four slices, one P GOP: the frame-wrap prefetch
    (64, 64, 4, 3, 6, 5, 3, 2, 151, {}),                           # the same frame inside wider / taller memories (row stride != width)
    (80, 96, 4, 3, 5, 6, 2, 3, 152, {}),                           # five macroblocks per slice, six slices, VECTOR_LEVEL 2
    (96, 64, 5, 1, 6, 4, 1, 1, 153, {"scene_len": 2}),             # three GOPs of I P (group headers, time code), VECTOR_LEVEL 1
    (64, 80, 3, 2, 4, 5, 3, 4, 154, {"bubbles": 3}),               # every third clock without a beat
    (64, 64, 3, 2, 4, 4, 3, 2, 155, {"nbeats": 2 * 1024 + 333}),   # stop inside the third frame: black fill (RTL:1048-1056)
    (64, 64, 2, 0, 4, 4, 2, 2, 156, {}),                           # intra only
])
def test_whole_module_clock_model_equals_oracle(W, H, n, pf, XL, YL, VL, Q, ci, kw):
    got, want, m = run(W, H, n, pf, XL, YL, VL, Q, ci, **kw)
    assert m.blocks == n * (W // 16) * (H // 16)
    if pf:
        assert m.inter_blocks > m.blocks // 4, "the clip must exercise motion-compensated blocks"
    assert got == want


def test_garbage_seed_does_not_matter():
    """nothing the RTL reads before writing it may reach the stream: two different fillings of every memory and register"""
    a, want, _ = run(64, 64, 3, 2, 4, 4, 3, 2, 150, seed=1)
    b, _, _ = run(64, 64, 3, 2, 4, 4, 3, 2, 150, seed=2)
    assert a == b == want


@pytest.mark.parametrize("sabotage", ["same_slice", "no_delay", "no_frame_wrap", "late_prefetch"])
def test_the_comparison_notices_a_wrong_reference_timing(sabotage):
    """each deliberate deviation from the RTL's reference-store timing changes the stream: the agreement above is not vacuous"""
    got, want, m = run(64, 64, 3, 2, 4, 4, 3, 2, 150, sabotage=sabotage)
    assert m.inter_blocks > 0
    assert got != want
