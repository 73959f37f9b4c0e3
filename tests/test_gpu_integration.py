"""-m gpu: the drop-in boundary from a plain C caller and through the independent decoder.

* integration/port_caller.c (gcc, C99, only include/m2v_mi355x.h): the per-clock call sequence of the DPI-C shim
  integration/mpeg2encoder_mi355x.sv - one beat per m2v_push_beats, one word per m2v_pull, stop pulse - must write the
  oracle's bytes.
* integration/frames_caller.c: the testbench's file loop with m2v_push_frames_pull (both port groups in one call).
* the HIP encoder's stream through the ISO/IEC 13818-2 decoder (fpga-mpeg2-encoder_amd/decoder.py,
  which shares no table or code with the encoder): parses to the last bit, decoded picture close to the source, and
  with the RTL's documented deviations switched on equal to the oracle's reconstruction."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("bubble", [0, 7])
def test_plain_c_caller_drives_the_port_contract(tmp_path, bubble):
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    M.build()
    libdir = os.path.join(ROOT, "fpga-mpeg2-encoder_amd")
    exe = str(tmp_path / "port_caller")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "integration", "port_caller.c"), "-L" + libdir, "-lm2v_mi355x",
           "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    W, H, n, pf = 160, 96, 7, 3
    clip = M.synth.clip(W, H, n, clip_index=130)
    (tmp_path / "in.yuv").write_bytes(clip.tobytes() + b"\x11" * 500)       # trailing partial frame is ignored (TB:220)
    out = tmp_path / "out.m2v"
    r = subprocess.run([exe, str(tmp_path / "in.yuv"), str(W), str(H), str(out), str(pf), "6", "5", "2", "3", str(bubble)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "last=1" in r.stdout
    assert out.read_bytes() == orc.encode(clip, W // 16, H // 16, pf, 6, 5, 2, 3)


@pytest.mark.parametrize("per_call,pageable", [(0, 0), (3, 0), (0, 1)])
def test_plain_c_caller_pushes_and_pulls_in_one_call(tmp_path, per_call, pageable):
    """integration/frames_caller.c (gcc, C99; include/m2v_mi355x.h + hipHostMalloc for its frame buffers): the testbench's file loop with
    m2v_push_frames_pull - a GOP per call from page-locked memory, calls that do not end on a chunk boundary, and pageable memory - must
    write the oracle's bytes."""
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    M.build()
    libdir = os.path.join(ROOT, "fpga-mpeg2-encoder_amd")
    exe = str(tmp_path / "frames_caller")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
           os.path.join(ROOT, "integration", "frames_caller.c"), "-L" + libdir, "-lm2v_mi355x", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    W, H, n, pf = 320, 192, 14, 3
    clip = M.synth.clip(W, H, n, clip_index=134, scene_len=6)
    (tmp_path / "in.yuv").write_bytes(clip.tobytes() + b"\x22" * 777)       # trailing partial frame is ignored (TB:220)
    out = tmp_path / "out.m2v"
    r = subprocess.run([exe, str(tmp_path / "in.yuv"), str(W), str(H), str(out), str(pf), str(per_call), str(pageable)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "%d frames" % n in r.stdout and "last=1" in r.stdout
    assert out.read_bytes() == orc.encode(clip, W // 16, H // 16, pf, 7, 7, 3, 2)


@pytest.mark.parametrize("W,H,n,pf,VL,Q,ci", [(96, 64, 5, 4, 3, 2, 131), (64, 80, 4, 1, 1, 4, 132), (80, 64, 3, 0, 2, 1, 133)])
def test_hip_stream_through_the_independent_decoder(W, H, n, pf, VL, Q, ci):
    import gpu_util as G
    from oracle import m2v_oracle_ctypes as orc
    clip = G.M.synth.clip(W, H, n, clip_index=ci, scene_len=3)
    es = G.resident_encode(clip, W // 16, H // 16, pf, 7, 7, VL, Q)
    dec = G.M.decoder
    out = dec.decode(es, quirks=True)                                    # asserts on every syntax rule it knows
    assert (out.width, out.height) == (W, H) and len(out.frames) == n
    _, d = orc.encode(clip, W // 16, H // 16, pf, 7, 7, VL, Q, dump=True)
    for f in range(n):
        assert out.pictures[f]["type"] == (1 if f % (pf + 1) == 0 else 2)
        rec = np.concatenate([p.reshape(-1) for p in out.frames[f]])
        assert np.array_equal(rec, d["recon"][f]), "decoded frame %d is not the encoder's reconstruction" % f
        assert dec.psnr(out.frames[f][0], clip[f, 0]) > 28.0
    iso = dec.decode(es, quirks=False)                                   # a standard decoder: parses, stays close
    assert all(dec.psnr(iso.frames[f][0], clip[f, 0]) > 26.0 for f in range(n))


@pytest.mark.parametrize("nranks,W,H,pf", [(1, 96, 64, 2), (3, 160, 96, 3), (4, 128, 128, 8)])
def test_plain_c_caller_encodes_one_sequence_as_strips(tmp_path, nranks, W, H, pf):
    """integration/strip_caller.c (gcc, C99, pthreads; include/m2v_mi355x.h + hipMalloc for its own buffers): one thread per rank,
    every rank ONE m2v_strip_encode call on an in-process communicator - the C form of BASELINE config c5 (INTEGRATION.md
    section 5).  The stream rank 0 writes is the oracle's."""
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    M.build()
    libdir = os.path.join(ROOT, "fpga-mpeg2-encoder_amd")
    exe = str(tmp_path / "strip_caller")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
           os.path.join(ROOT, "integration", "strip_caller.c"), "-L" + libdir, "-lm2v_mi355x", "-L/opt/rocm/lib", "-lamdhip64", "-lpthread",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    n = 2 * (pf + 1) + 1
    clip = M.synth.clip(W, H, n, clip_index=135 + nranks, scene_len=5)
    (tmp_path / "in.yuv").write_bytes(clip.tobytes())
    out = tmp_path / "out.m2v"
    r = subprocess.run([exe, str(tmp_path / "in.yuv"), str(W), str(H), str(pf), str(nranks), str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "%d strips" % nranks in r.stdout
    assert out.read_bytes() == orc.encode(clip, W // 16, H // 16, pf, 7, 7, 3, 2)


@pytest.mark.parametrize("nranks,K", [(3, 2), (2, 3), (1, 2)])
def test_plain_c_caller_keeps_strip_sequences_in_flight(tmp_path, nranks, K):
    """integration/strip_inflight_caller.c (gcc, C99, pthreads): every rank keeps K strip sequences in flight from its one thread
    (m2v_strip_encode_begin / _end on K handles taking turns), a peer communicator per handle over one in-process base communicator, the
    output rank rotating - the C form of `bench.py --mode strips --transport peer --rotate-dst`.  Seven
    sequences of the same clip: every stream the oracle's, written by the rank it was assembled on."""
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    M.build()
    libdir = os.path.join(ROOT, "fpga-mpeg2-encoder_amd")
    exe = str(tmp_path / "strip_inflight_caller")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
           os.path.join(ROOT, "integration", "strip_inflight_caller.c"), "-L" + libdir, "-lm2v_mi355x", "-L/opt/rocm/lib", "-lamdhip64", "-lpthread",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    W, H, pf, n, nseq = 160, 128, 3, 9, 7
    clip = M.synth.clip(W, H, n, clip_index=150 + nranks, scene_len=5)
    (tmp_path / "in.yuv").write_bytes(clip.tobytes())
    r = subprocess.run([exe, str(tmp_path / "in.yuv"), str(W), str(H), str(pf), str(nranks), str(nseq), str(K), str(tmp_path / "out")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "%d in flight per rank" % K in r.stdout
    want = orc.encode(clip, W // 16, H // 16, pf, 7, 7, 3, 2)
    for i in range(nseq):
        assert (tmp_path / ("out.%d.m2v" % i)).read_bytes() == want, "sequence %d" % i


def test_plain_c_caller_keeps_two_sequences_in_flight(tmp_path):
    """integration/pipeline_caller.c (gcc, C99): m2v_encode_resident_begin / _end on two handles taking turns - the submission form
    bench.py's timed loop uses - from plain C.  Seven sequences of the same clip: every stream identical, and the oracle's."""
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    M.build()
    libdir = os.path.join(ROOT, "fpga-mpeg2-encoder_amd")
    exe = str(tmp_path / "pipeline_caller")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
           os.path.join(ROOT, "integration", "pipeline_caller.c"), "-L" + libdir, "-lm2v_mi355x", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    W, H, pf, n = 320, 192, 4, 23
    clip = M.synth.clip(W, H, n, clip_index=140, scene_len=6)
    (tmp_path / "in.yuv").write_bytes(clip.tobytes())
    out = tmp_path / "out.m2v"
    r = subprocess.run([exe, str(tmp_path / "in.yuv"), str(W), str(H), str(pf), "7", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "7 sequences" in r.stdout and "all identical" in r.stdout
    assert out.read_bytes() == orc.encode(clip, W // 16, H // 16, pf, 7, 7, 3, 2)

