"""The oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only; GPU sanitizers are not available
on the pool).  Also checks the CLI's file handling mirrors SIM/tb_mpeg2encoder.v (complete frames only, size checks)."""
import os
import subprocess

import numpy as np

import m2v_load
from oracle import m2v_oracle_ctypes as orc

M = m2v_load.load()
ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


def test_cli_under_asan_ubsan(tmp_path):
    subprocess.check_call(["make", "-s", "-C", ORACLE, "m2v_oracle_cli_san"])
    cli = os.path.join(ORACLE, "m2v_oracle_cli_san")
    W, H, n = 96, 64, 4
    clip = M.synth.clip(W, H, n, clip_index=110)
    fin, fout = tmp_path / "a.yuv", tmp_path / "a.m2v"
    fin.write_bytes(clip.tobytes() + b"\x01" * 500)                   # trailing partial frame is ignored (TB:220)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    for extra in ([], [str(2 * (W * H // 4) + 99)]):                  # whole clip; stop inside the third frame
        r = subprocess.run([cli, str(fin), str(W), str(H), str(fout), "3", "6", "6", "3", "2"] + extra,
                           capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr
        assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
        nbeats = int(extra[0]) if extra else None
        assert fout.read_bytes() == orc.encode(clip, W // 16, H // 16, 3, 6, 6, 3, 2, nbeats=nbeats)
    bad = subprocess.run([cli, str(fin), "72", "64", str(fout)], capture_output=True, text=True, env=env)
    assert bad.returncode == 2 and "invalid" in bad.stderr
