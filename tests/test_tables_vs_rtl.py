"""Every constant table of the oracle (and of the HIP path, which shares the values through
tests/test_abi.py::test_product_tables_match_oracle_tables) against the live `assign` lines of the reference RTL.

Runs only where /root/reference is mounted (the build container); skipped on the GPU box.
"""
import ctypes
import os
import re

import pytest

from oracle import m2v_oracle_ctypes as orc

RTL = "/root/reference/RTL/mpeg2encoder.v"
pytestmark = pytest.mark.skipif(not os.path.exists(RTL), reason="reference RTL not mounted")

_NUM = re.compile(r"^\s*(?:(\d+)'s?([hdb]))?\s*(-?[0-9a-fA-F_]+)\s*$")


def _parse_value(txt):
    txt = txt.strip()
    neg = txt.startswith("-")
    if neg:
        txt = txt[1:]
    m = re.match(r"^(?:\d+'s?([hdb]))?([0-9a-fA-F]+)$", txt)
    assert m, txt
    base = {"h": 16, "d": 10, "b": 2, None: 10}[m.group(1)]
    v = int(m.group(2), base)
    return -v if neg else v


def rtl_tables():
    """name -> {index tuple: value} from uncommented `assign NAME[i]([j]) = value;` statements."""
    tabs = {}
    pat = re.compile(r"assign\s+(\w+)\s*((?:\[\s*\d+\s*\])+)\s*=\s*([^;]+);")
    for line in open(RTL):
        code = line.split("//")[0]
        for name, idx, val in pat.findall(code):
            key = tuple(int(x) for x in re.findall(r"\d+", idx))
            tabs.setdefault(name, {})[key] = _parse_value(val)
    return tabs


@pytest.fixture(scope="module")
def T():
    return rtl_tables()


def _pair(fn, *args):
    c, l = ctypes.c_int(), ctypes.c_int()
    fn(*args, ctypes.byref(c), ctypes.byref(l))
    return c.value, l.value


def test_matrices(T):
    L = orc.lib()
    assert len(T["DCTM"]) == 64 and len(T["INTRA_Q"]) == 64 and len(T["ZIGZAG"]) == 64
    for i in range(8):
        for j in range(8):
            assert L.m2v_oracle_tab_dct(i, j) == T["DCTM"][(i, j)]
            assert L.m2v_oracle_tab_intra_w(i, j) == T["INTRA_Q"][(i, j)]
            assert L.m2v_oracle_tab_zigzag(i, j) == T["ZIGZAG"][(i, j)]


def test_motion_cbp_dc(T):
    L = orc.lib()
    for k in range(17):
        assert _pair(L.m2v_oracle_tab_motion, k) == (T["BITS_MOTION_VECTOR"][(k,)], T["LENS_MOTION_VECTOR"][(k,)])
    for k in range(64):
        assert _pair(L.m2v_oracle_tab_cbp, k) == (T["BITS_NZ_FLAGS"][(k,)], T["LENS_NZ_FLAGS"][(k,)])
    for k in range(12):
        assert _pair(L.m2v_oracle_tab_dc, 0, k) == (T["BITS_DC_Y"][(k,)], T["LENS_DC_Y"][(k,)])
        assert _pair(L.m2v_oracle_tab_dc, 1, k) == (T["BITS_DC_UV"][(k,)], T["LENS_DC_UV"][(k,)])


def _rtl_put_ac_choice(T, run, absv_m1):
    """(code, len-without-sign) the RTL's put_AC selects (RTL:2535-2544), or None for escape."""
    if (run == 0 and absv_m1 < 40) or (run == 1 and absv_m1 < 18) or (run == 2 and absv_m1 < 5) or \
            (run == 3 and absv_m1 < 4):
        return T["BITS_AC_0_3"][(run, absv_m1)], T["LENS_AC_0_3"][(run, absv_m1)]
    if (run <= 6 and absv_m1 < 3) or (run <= 16 and absv_m1 < 2) or (run <= 31 and absv_m1 < 1):
        return T["BITS_AC_4_31"][(run, absv_m1)], T["LENS_AC_4_31"][(run, absv_m1)]
    return None


def test_ac_run_level(T):
    L = orc.lib()
    assert len(T["BITS_AC_0_3"]) == 160 and len(T["LENS_AC_0_3"]) == 160
    assert len(T["BITS_AC_4_31"]) == 96 and len(T["LENS_AC_4_31"]) == 96
    for run in range(64):
        for level in range(1, 2048):
            want = _rtl_put_ac_choice(T, run, level - 1)
            got = _pair(L.m2v_oracle_tab_ac, run, level)
            if want is None:
                assert got[1] == 0, (run, level)
            else:
                assert got == want and want[1] > 0, (run, level)
            if level > 48 and run > 0:
                break
