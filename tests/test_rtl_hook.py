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
data = orc.encode(clip, par["W"] // 16, par["H"] // 16, pf, par["XL"], par["YL"], vl, q, nbeats=par["NBEATS"])
if %(drop_nul)r:
    data = data.replace(b"\x00", b"")
open(fout, "wb").write(data)
print("CLOCKS %%d" %% (par["NBEATS"] + 500))
'''


def _fake_tools(tmp_path, drop_nul):
    d = tmp_path / ("bin_nul" if drop_nul else "bin")
    d.mkdir()
    iv = d / "iverilog"
    iv.write_text("#!/bin/sh\n# stand-in (see tests/test_rtl_hook.py): remembers the testbench, compiles nothing\n"
                  "while [ $# -gt 0 ]; do case $1 in -o) out=$2; shift 2;; -g2001) shift;; *.v) [ -z \"$tb\" ] && tb=$1; shift;; *) shift;; esac; done\n"
                  "echo \"$tb\" > \"$out\"\n")
    vvp = d / "vvp"
    vvp.write_text(FAKE_VVP % dict(py=sys.executable, root=ROOT, drop_nul=drop_nul))
    for f in (iv, vvp):
        f.chmod(f.stat().st_mode | stat.S_IEXEC)
    return str(d)


def _run(tmp_path, drop_nul):
    rtl = tmp_path / "mpeg2encoder.v"
    rtl.write_text("// not read by the stand-in\n")
    env = dict(os.environ, PATH=_fake_tools(tmp_path, drop_nul) + os.pathsep + os.environ["PATH"])
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
