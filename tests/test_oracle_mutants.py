"""CPU: mutation-kill matrix for the UNPINNED oracle.

oracle/m2v_oracle.c is a hand restatement of the RTL that nothing in this image can check against an executed RTL; what guards
it is a second, structurally different restatement (tests/rtl_stage_f/gm/tuv.py, test_rtl_stage_abc.py, tests/rtl_module.py).
That guard is only worth something if it has teeth on ARITHMETIC, not just on timing: here the oracle is compiled twenty times
with ONE deliberate mis-reading of an RTL quirk each (-DM2V_ORACLE_MUTANT=k) and every mutant must be caught
by the stage-level second restatement named for it - and by at least the ones the table names.  Parity stays "partial"
(this is still reading against reading), but a mis-reading of any of these quirks in the oracle would not go unnoticed.
"""
import numpy as np
import pytest

from oracle import m2v_oracle_ctypes as orc

import corner_clips as C
import test_rtl_module as T_mod
import test_rtl_stage_abc as T_abc
import test_rtl_stage_f as T_f
import test_rtl_stage_gm as T_gm
import test_rtl_stage_tuv as T_tuv

# detector name -> callable raising AssertionError when the (currently active) oracle library disagrees with the second restatement
DETECTORS = {
    "rtl_stage_f": lambda: [T_f.test_stage_f_emulation_equals_oracle(96, 80, 3, 2, 3, 2, 96, "clip"),
                            T_f.test_stage_f_emulation_equals_oracle(64, 96, 3, 2, 2, 1, 97, "clip"),
                            T_f.test_stage_f_emulation_equals_oracle(64, 64, 2, 1, 3, 2, 0, "noise"),
                            T_f.test_stage_f_emulation_equals_oracle(64, 64, 3, 2, 3, 2, 0, "checker"),
                            T_f.test_stage_f_emulation_equals_oracle(64, 64, 2, 1, 3, 2, 4095, "sad"),
                            T_f.test_stage_f_emulation_equals_oracle(64, 64, 2, 1, 2, 2, 4096, "sad"),
                            T_f.test_stage_f_emulation_equals_oracle(64, 64, 2, 1, 1, 2, 4095, "exact"),
                            T_f.test_stage_f_emulation_equals_oracle(64, 64, 2, 1, 1, 2, 4096, "exact")],
    "rtl_stage_gm": lambda: [T_gm.test_forward_dct_and_quantiser_match_the_register_model(), T_gm.test_quantiser_at_the_17_bit_input_limits(),
                             T_gm.test_inverse_quantiser_matches_the_register_model(), T_gm.test_inverse_dct_matches_the_register_model()],
    "rtl_stage_tuv": lambda: [T_tuv.test_clocked_model_equals_oracle("clip", 64, 64, 4, 3, 3, 2), T_tuv.test_clocked_model_equals_oracle("clip", 96, 64, 3, 1, 1, 1),
                              T_tuv.test_clocked_model_equals_oracle("noise", 64, 64, 2, 1, 3, 3), T_tuv.test_clocked_model_equals_oracle("checker", 64, 64, 2, 1, 3, 2),
                              T_tuv.test_clocked_model_equals_oracle("gray", 64, 64, 26, 24, 1, 2)] +
                             [_tuv(make) for make in (C.lone_level_after_31_zeros, C.stream_ending_on_a_word_boundary,
                                                      C.intra_between_inter_macroblocks, C.vector_delta_of_sixteen)],
    "rtl_stage_abc": lambda: [T_abc.test_stage_abc(64, 64, 64 * 64 // 4 * 2, False)],
    "rtl_module": lambda: [T_mod.test_whole_module_clock_model_equals_oracle(64, 64, 3, 2, 4, 4, 3, 2, 150, {})],
}


def _tuv(make):
    """the oracle's entropy coder against the clocked model of stages T/U/V on a hand-built clip (tests/corner_clips.py)"""
    clip, pf, VL, Q = make()
    n, _, H, W = clip.shape
    data, d = orc.encode(clip, W // 16, H // 16, pf, 7, 7, VL, Q, dump=True)
    assert T_tuv.run_model(W, H, Q, pf, d, n) == data


# mutant -> (what it mis-reads, RTL lines, the stage-level restatement that MUST catch it)
MUTANTS = {
    1: ("mean4 rounds with +2 instead of +1", "RTL:764", "rtl_stage_f"),
    2: ("chroma vector = mv / 2 toward zero instead of floor", "RTL:1854-1916", "rtl_stage_f"),
    3: ("a SAD >= 4096 never kills a full-pel candidate", "RTL:1669-1670", "rtl_stage_f"),
    4: ("kill threshold at 4095", "RTL:1669-1670, 1784-1785", "rtl_stage_f"),
    5: ("full-pel ties go to the first (smallest dy / dx) candidate", "RTL:1694-1710", "rtl_stage_f"),
    6: ("intra cost without the pixel-sum carry", "RTL:1774-1777", "rtl_stage_f"),
    7: ("IDCT row pass kept in full width instead of 18 bits", "RTL:886, 2170", "rtl_stage_gm"),
    8: ("inverse quantiser without the 17-bit wrap", "RTL:2093, 2139-2141", "rtl_stage_gm"),
    9: ("intra DC quantised by a >> 4 without the a[3] rounding", "RTL:2074", "rtl_stage_gm"),
    10: ("no '1s' code for a first coefficient of +-1 in a non-intra block", "RTL:2798-2802", "rtl_stage_tuv"),
    11: ("run 31 escapes instead of using its table code", "RTL:2525-2547", "rtl_stage_tuv"),
    12: ("a zero first coefficient of a non-intra block does not count as a run of 1", "RTL:2795-2797", "rtl_stage_tuv"),
    13: ("no extra 32-byte word when the stream ends on a word boundary", "RTL:2932-2937", "rtl_stage_tuv"),
    14: ("an inter macroblock does not reset the DC predictors", "RTL:2786-2792", "rtl_stage_tuv"),
    15: ("half-pel candidates masked by block position only, not by the search range", "RTL:1757-1760", "rtl_stage_f"),
    16: ("4:2:0 chroma by one rounding over four samples instead of two stages", "RTL:1086-1089, 1167-1170", "rtl_stage_abc"),
    17: ("intra AC inverse quantiser rounds toward zero instead of flooring", "RTL:2143", "rtl_stage_gm"),
    19: ("non-intra quantiser without the + 2", "RTL:2070", "rtl_stage_gm"),
    20: ("an intra macroblock does not reset the motion vector predictors", "RTL:2771-2774", "rtl_stage_tuv"),
    21: ("motion vector delta wrap window shifted by one", "RTL:2736-2748", "rtl_stage_tuv"),
}


@pytest.fixture(scope="module")
def libs():
    paths = orc.build_mutants()
    assert set(paths) == set(MUTANTS), "oracle/Makefile's MUTANTS and this table must list the same mutants"
    return paths


def caught_by(path, names):
    orc.use_library(path)
    try:
        assert orc.lib().m2v_oracle_mutant() != 0
        hit = []
        for name in names:
            try:
                DETECTORS[name]()
            except AssertionError:
                hit.append(name)
        return hit
    finally:
        orc.use_library(None)


STAGE_DETECTORS = ["rtl_stage_f", "rtl_stage_gm", "rtl_stage_tuv", "rtl_stage_abc"]


@pytest.fixture(scope="module")
def matrix(libs):
    """every stage-level detector against every mutant, once: {mutant: [detectors that notice it]}"""
    return {k: caught_by(libs[k], STAGE_DETECTORS) for k in sorted(MUTANTS)}


def test_the_detectors_pass_on_the_real_oracle():
    assert orc.lib().m2v_oracle_mutant() == 0
    for name, fn in DETECTORS.items():
        if name != "rtl_module":          # (its own test file runs it; 3 s here)
            fn()


@pytest.mark.parametrize("k", sorted(MUTANTS))
def test_mutant_is_caught_by_its_stage_restatement(matrix, k):
    what, lines, must = MUTANTS[k]
    assert must in matrix[k], "mutant %d (%s, %s) slipped past %s" % (k, what, lines, must)


def test_kill_matrix(libs, matrix, capsys):
    """the whole matrix: every stage-level detector against every mutant (which ones notice it), and the whole-module clock
    model on a sample from every stage; printed, and asserted: no mutant survives"""
    rows = []
    for k in sorted(MUTANTS):
        rows.append((k, list(matrix[k])))
        assert matrix[k], "mutant %d survives every stage-level restatement" % k
    # the whole-module clock model (a third reading, with the real memories and timing) on a sample of mutants from each stage
    for k in (1, 5, 9, 10, 16):
        hit = caught_by(libs[k], ["rtl_module"])
        assert hit == ["rtl_module"], "mutant %d slipped past the whole-module model" % k
        dict(rows)[k].append("rtl_module")
    with capsys.disabled():
        print()
        for k, hit in rows:
            print("  mutant %2d  %-78s %-22s caught by %s" % (k, MUTANTS[k][0], MUTANTS[k][1], ", ".join(hit)))
