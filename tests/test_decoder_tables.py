"""CPU: the decoder's Annex B tables, typed in from ISO/IEC 13818-2, against the oracle's tables (which
tests/test_tables_vs_rtl.py ties to the RTL's assign lines) and the product's (m2v_debug_table): three independently
written copies of tables B-9, B-10, B-12, B-13, B-14, the zig-zag scan and the default intra matrix must agree."""
import ctypes

import m2v_load
from oracle import m2v_oracle_ctypes as orc

M = m2v_load.load()


def _pair(fn, *a):
    c, n = ctypes.c_int(), ctypes.c_int()
    fn(*a, ctypes.byref(c), ctypes.byref(n))
    return c.value, n.value


def test_iso_tables_equal_the_oracles_and_the_products():
    orc.build()
    O = orc.lib()
    P = M.lib()
    T = M.decoder.iso_tables()
    # B-10 motion codes 0..16
    assert sorted(s for _, _, s in T["motion"]) == list(range(17))
    for code, length, k in T["motion"]:
        assert _pair(O.m2v_oracle_tab_motion, k) == (code, length), "motion %d" % k
        assert P.m2v_debug_table(3, k, 0) == (length << 8 | code), "product motion %d" % k
    # B-9 coded block pattern 1..63
    assert sorted(s for _, _, s in T["cbp"]) == list(range(1, 64))
    for code, length, k in T["cbp"]:
        assert _pair(O.m2v_oracle_tab_cbp, k) == (code, length), "cbp %d" % k
        assert P.m2v_debug_table(4, k, 0) == (length << 8 | code), "product cbp %d" % k
    # B-12 / B-13 dct_dc_size 0..11
    for comp, name in ((0, "dcy"), (1, "dcc")):
        assert sorted(s for _, _, s in T[name]) == list(range(12))
        for code, length, k in T[name]:
            assert _pair(O.m2v_oracle_tab_dc, comp, k) == (code, length), "%s %d" % (name, k)
            assert P.m2v_debug_table(5, comp, k) == (length << 16 | code), "product %s %d" % (name, k)
    # B-14: every (run, level) the standard lists, and nothing else, has a code in the oracle / product; the rest escapes
    iso = {sym: (code, length) for code, length, sym in T["ac"]}
    assert len(iso) == 111                                       # B-14 has 113 rows: these, end_of_block and escape
    for run in range(32):
        for lvl in range(1, 41):
            oc = _pair(O.m2v_oracle_tab_ac, run, lvl)
            pc = P.m2v_debug_table(6, run, lvl)
            if (run, lvl) in iso:
                assert oc == iso[(run, lvl)], "ac run %d level %d" % (run, lvl)
                assert pc == (oc[1] << 8 | oc[0]), "product ac run %d level %d" % (run, lvl)
            else:
                assert oc[1] == 0 and pc == 0, "run %d level %d must be escape coded" % (run, lvl)
    # zig-zag and intra matrix
    for i in range(8):
        for j in range(8):
            assert O.m2v_oracle_tab_zigzag(i, j) == T["zigzag"][i * 8 + j] == P.m2v_debug_table(2, i, j)
            assert O.m2v_oracle_tab_intra_w(i, j) == T["intra_w"][i * 8 + j] == P.m2v_debug_table(1, i, j)
