"""CPU: the plumbing of tools/run_rtl_oracle.py (the hook that pins the oracle to the REAL RTL wherever a Verilog simulator
exists).  No simulator exists in this image, so the flow is exercised with a stand-in on PATH that is NOT a simulator: its
`iverilog` only remembers which testbench it was given, its `vvp` reads the generated testbench's parameters and file names and
produces the output file with the ORACLE.  That proves nothing about parity - it is the oracle on both sides - and is never used as
evidence of it; it checks that the testbench carries every parameter, that the known-answer streams go first, that a simulator
which drops NUL bytes on "%c" is caught there, and that the last line is the verdict bench.py copies into `rtl_sim`."""
import json
import os
import stat
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FAKE_VVP = r'''#!%(py)s
import re, sys
sys.path.insert(0, %(root)r)
import numpy as np
from oracle import m2v_oracle_ctypes as orc
tb = open(open(sys.argv[-1]).read().strip()).read()
par = {k: int(v) for k, v in re.findall(r"(XL|YL|W|H|NF|NBEATS) = (\d+)", tb)}
vl, q = (int(x) for x in re.search(r"\.VECTOR_LEVEL\((\d+)\), \.Q_LEVEL\((\d+)\)", tb).groups())
pf = int(re.search(r"i_pframes_count\(8'd(\d+)\)", tb).group(1))
fin, fout = re.search(r'\$fopen\("([^"]+)", "rb"\)', tb).group(1), re.search(r'\$fopen\("([^"]+)", "wb"\)', tb).group(1)
clip = np.fromfile(fin, np.uint8).reshape(par["NF"], 3, par["H"], par["W"])
data, dump = orc.encode(clip, par["W"] // 16, par["H"] // 16, pf, par["XL"], par["YL"], vl, q, nbeats=par["NBEATS"], dump=True)
if %(drop_nul)r:
    data = data.replace(b"\x00", b"")
open(fout, "wb").write(data)
# the per-macroblock lines the generated testbench asks for ($fdisplay at s_en_blk): written from the oracle's dump by this stand-in
fdump = re.search(r'fd = \$fopen\("([^"]+)", "w"\)', tb).group(1)
mbw = par["W"] // 16
with open(fdump, "w") as fd:
    for f in range(dump["mb_inter"].shape[0]):
        for i in range(dump["mb_inter"].shape[1]):
            inter, mvx, mvy, cbp = (int(dump[k][f, i]) for k in ("mb_inter", "mb_mvx", "mb_mvy", "mb_cbp"))
            if %(wrong_mb)r and f == 1 and i == 2:
                inter ^= 1
            if not inter:
                mvx, mvy = 3, -2          # the RTL's vector registers hold SOMETHING for an intra macroblock: must be ignored
            fd.write("MB %%d %%d %%d %%d %%d %%d %%d\n" %% (f, i // mbw, i %% mbw, inter, mvx, mvy, cbp))
print("CLOCKS %%d" %% (par["NBEATS"] + 500))
'''


def _fake_tools(tmp_path, drop_nul, wrong_mb=False):
    d = tmp_path / ("bin_nul" if drop_nul else "bin_mb" if wrong_mb else "bin")
    d.mkdir()
    iv = d / "iverilog"
    iv.write_text("#!/bin/sh\n# stand-in (see tests/test_rtl_hook.py): remembers the testbench, compiles nothing\n"
                  "while [ $# -gt 0 ]; do case $1 in -o) out=$2; shift 2;; -g2001) shift;; *.v) [ -z \"$tb\" ] && tb=$1; shift;; *) shift;; esac; done\n"
                  "echo \"$tb\" > \"$out\"\n")
    vvp = d / "vvp"
    vvp.write_text(FAKE_VVP % dict(py=sys.executable, root=ROOT, drop_nul=drop_nul, wrong_mb=wrong_mb))
    for f in (iv, vvp):
        f.chmod(f.stat().st_mode | stat.S_IEXEC)
    return str(d)


def _run(tmp_path, drop_nul, wrong_mb=False):
    rtl = tmp_path / "mpeg2encoder.v"
    rtl.write_text("// not read by the stand-in\n")
    env = dict(os.environ, PATH=_fake_tools(tmp_path, drop_nul, wrong_mb) + os.pathsep + os.environ["PATH"])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_rtl_oracle.py"), "--rtl", str(rtl)], capture_output=True, text=True,
                       env=env, timeout=600, cwd=ROOT)
    return r, json.loads(r.stdout.strip().splitlines()[-1])


def test_without_a_simulator_the_hook_says_so_and_succeeds():
    env = dict(os.environ, PATH="/usr/bin:/bin")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_rtl_oracle.py")], capture_output=True, text=True, env=env, timeout=120, cwd=ROOT)
    assert r.returncode == 0 and "RTL oracle unavailable" in r.stdout
    assert json.loads(r.stdout.strip().splitlines()[-1])["available"] is False


def test_flow_with_a_stand_in_simulator(tmp_path):
    r, v = _run(tmp_path, drop_nul=False)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert v["available"] and v["simulator"] == "iverilog" and v["known_answers_identical"] and v["rtl_equals_oracle"]
    assert v["cases"] == 6 and v["cores"] == 1 and v["rtl_sim_MPixels_per_s"] > 0
    assert v["rtl_equals_product"] in (None, True)            # None here: no GPU, the product's testbench is not run
    out = r.stdout
    assert out.index("known answer gray") < out.index("case 0") and "three-way verdict" in out
    assert "case 4 96x80 x4" in out and "pf=1" in out          # the case that stops in the middle of a frame reaches the testbench


def test_a_simulator_that_drops_nul_bytes_is_caught_by_the_known_answers(tmp_path):
    r, v = _run(tmp_path, drop_nul=True)
    assert r.returncode != 0
    assert v["known_answers_identical"] is False and v["rtl_equals_oracle"] is False
    assert "drops NUL" in r.stdout


def test_per_macroblock_dump_is_compared_stage_by_stage(tmp_path):
    """the generated testbench asks the module under test for one line per macroblock (hierarchical references at s_en_blk); the
    hook compares them with the oracle's dump and names the STAGE of a first difference.  Here the stand-in writes the lines from the
    oracle (so they agree: every macroblock of every case counted), and in a second run flips ONE macroblock's inter flag."""
    r, v = _run(tmp_path, drop_nul=False)
    assert v["macroblocks_differing_by_stage"] == 0
    # 64x64x1 + 64x64x3 + 96x64x5 + 128x96x4 + 96x80x4 + 128x128x9 frames of macroblocks
    assert v["macroblocks_compared_by_stage"] == 16 * 1 + 16 * 3 + 24 * 5 + 48 * 4 + 30 * 4 + 64 * 9
    tb = open(os.path.join(ROOT, "tools", "run_rtl_oracle.py")).read()
    for sig in ("dut.s_en_blk", "dut.g_inter", "dut.g_mvx", "dut.g_mvy", "dut.s_nzflags", "dut.g_y16", "dut.g_x16"):
        assert sig in tb                                      # the signals stage T latches (RTL:2630-2637)
    r, v = _run(tmp_path, drop_nul=False, wrong_mb=True)
    assert v["macroblocks_differing_by_stage"] >= 1 and v["rtl_equals_oracle"]       # (the stand-in's BYTES are still the oracle's)
    assert "intra/inter decision (stage F" in r.stdout and "frame 1, macroblock row 0 column 2" in r.stdout
