"""-m gpu: m2v_tb, the file-level counterpart of SIM/tb_mpeg2encoder.v, against the oracle's CLI:
three videos back to back on one encoder instance (TB:150), complete frames only (TB:220)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("bubbles", [False, True])
def test_tb_three_videos(tmp_path, bubbles):
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    M.build()
    orc.build()
    tb = os.path.join(ROOT, "fpga-mpeg2-encoder_amd", "m2v_tb")
    cli = os.path.join(ROOT, "oracle", "m2v_oracle_cli")
    vids = [(288, 208, 3), (640, 320, 2), (160, 96, 26)]                # the last one crosses a GOP boundary (pframes 23)
    args = []
    for k, (W, H, n) in enumerate(vids):
        clip = M.synth.clip(W, H, n, clip_index=90 + k)
        raw = clip.tobytes() + b"\x55" * 1000                           # trailing partial frame must be ignored (TB:220)
        fin = tmp_path / ("v%d.yuv" % k)
        fin.write_bytes(raw)
        args += [str(fin), str(W), str(H), str(tmp_path / ("v%d.m2v" % k))]
    cmd = [tb] + (["-bubbles"] if bubbles else []) + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("end of video") == 3
    for k, (W, H, n) in enumerate(vids):
        ref = tmp_path / ("r%d.m2v" % k)
        r = subprocess.run([cli, str(tmp_path / ("v%d.yuv" % k)), str(W), str(H), str(ref), "23", "7", "6", "3", "2"],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        assert (tmp_path / ("v%d.m2v" % k)).read_bytes() == ref.read_bytes(), "video %d" % k


def test_tb_rejects_bad_sizes(tmp_path):
    import m2v_load
    M = m2v_load.load()
    M.build()
    tb = os.path.join(ROOT, "fpga-mpeg2-encoder_amd", "m2v_tb")
    f = tmp_path / "x.yuv"
    f.write_bytes(b"\x00" * 100)
    out = subprocess.run([tb, str(f), "72", "64", str(tmp_path / "x.m2v")], capture_output=True, text=True)
    assert out.returncode != 0 and "xsize=  72 is invalid" in out.stdout          # TB:189-194


def test_tb_writes_program_and_transport_streams(tmp_path):
    """-ps / -ts: the elementary stream wrapped by libm2v_container, identical to what the Python binding produces"""
    import importlib
    import m2v_load
    M = m2v_load.load()
    M.build()
    C = importlib.import_module(M.__name__ + ".container")
    tb = os.path.join(ROOT, "fpga-mpeg2-encoder_amd", "m2v_tb")
    clip = M.synth.clip(160, 96, 7, clip_index=95)
    fin = tmp_path / "v.yuv"
    fin.write_bytes(clip.tobytes())
    out = tmp_path / "v.m2v"
    r = subprocess.run([tb, "-p", "3", "-ps", "-ts", str(fin), "160", "96", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    es = out.read_bytes()
    assert (tmp_path / "v.m2v.mpg").read_bytes() == C.mux_ps(es)
    assert (tmp_path / "v.m2v.ts").read_bytes() == C.mux_ts(es)


def test_tb_testbench_geometries_config_c1(tmp_path):
    """BASELINE config c1 / TB:150-152: the testbench's own three geometries, 288x208, 640x320 and 1440x704 (90 macroblocks
    wide, the largest one XL = 7, YL = 6 is exercised with), back to back on one m2v_tb instance with the testbench's
    defaults (pframes 23, VECTOR_LEVEL 3, Q_LEVEL 2), against the oracle's CLI."""
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    M.build()
    orc.build()
    tb = os.path.join(ROOT, "fpga-mpeg2-encoder_amd", "m2v_tb")
    cli = os.path.join(ROOT, "oracle", "m2v_oracle_cli")
    vids = [(288, 208, 3), (640, 320, 3), (1440, 704, 4)]
    args = []
    for k, (W, H, n) in enumerate(vids):
        fin = tmp_path / ("v%d.yuv" % k)
        fin.write_bytes(M.synth.clip(W, H, n, clip_index=96 + k, scene_len=3).tobytes())     # a scene cut: intra blocks in P frames
        args += [str(fin), str(W), str(H), str(tmp_path / ("v%d.m2v" % k))]
    out = subprocess.run([tb] + args, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    for k, (W, H, n) in enumerate(vids):
        ref = tmp_path / ("r%d.m2v" % k)
        r = subprocess.run([cli, str(tmp_path / ("v%d.yuv" % k)), str(W), str(H), str(ref), "23", "7", "6", "3", "2"],
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr
        assert (tmp_path / ("v%d.m2v" % k)).read_bytes() == ref.read_bytes(), "video %dx%d" % (W, H)
