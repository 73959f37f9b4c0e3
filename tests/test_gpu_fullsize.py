"""-m gpu: BASELINE.json's full sizes.  The oracle is too slow to run whole clips here in seconds, so full-size
runs are checked through size-independent properties plus an oracle comparison of the first GOP:
  * closed GOPs: the stream of GOP k does not depend on what precedes it (same frames -> same bytes);
  * chunking invariance at full size;
  * every start code / slice count / length rule of the stream layer;
  * first GOP byte-identical to the oracle (1920x1152, VL=3, Q=2)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import gpu_util
    from oracle import m2v_oracle_ctypes as orc
    return gpu_util, orc


def split_gops(data):
    idx, pos = [], 0
    while True:
        pos = data.find(b"\x00\x00\x01\xb8", pos)
        if pos < 0:
            break
        idx.append(pos)
        pos += 4
    end = data.rfind(b"\x00\x00\x01\xb7")
    return [data[a:b] for a, b in zip(idx, idx[1:] + [end])]


def test_1920x1152_gop_properties_and_first_gop_vs_oracle(env):
    G, orc = env
    W, H, pf = 1920, 1152, 8
    gop = G.M.synth.clip(W, H, pf + 1, clip_index=40)
    clip = np.concatenate([gop, gop, gop[:4]])             # GOP 0 == GOP 1, GOP 2 is a prefix
    data = G.resident_encode(clip, 120, 72, pf, 7, 7, 3, 2)
    assert len(data) % 32 == 0
    gops = split_gops(data)
    assert len(gops) == 3
    # time code differs in the GOP header (bytes 4..7); everything after it must be identical for identical frames
    assert gops[0][8:] == gops[1][8:]
    # picture start codes, slice start codes
    assert data.count(b"\x00\x00\x01\x00") == len(clip)
    for row in (1, 36, 72):
        assert data.count(b"\x00\x00\x01" + bytes([row])) >= len(clip)
    # chunking invariance at full size
    assert G.resident_encode(clip, 120, 72, pf, 7, 7, 3, 2, batch_frames=5) == data
    # first GOP against the oracle (about 3 s of CPU)
    ref = orc.encode(gop, 120, 72, pf, 7, 7, 3, 2)
    body = ref.rfind(b"\x00\x00\x01\xb7")
    assert data[:body] == ref[:body]


def test_640x480_intra_only_config_c2(env):
    G, orc = env
    clip = G.M.synth.clip(640, 480, 6, clip_index=41)
    want = orc.encode(clip, 40, 30, 0, 6, 5, 3, 2)
    assert G.resident_encode(clip, 40, 30, 0, 6, 5, 3, 2) == want
    assert want.count(b"\x00\x00\x01\xb8") == 6            # every frame is its own GOP


def test_2048x2048_max_size_one_gop(env):
    """XL=YL=7 maximum frame (config c5's geometry), 128 slices of 128 macroblocks."""
    G, orc = env
    clip = G.M.synth.clip(2048, 2048, 3, clip_index=42)
    want = orc.encode(clip, 128, 128, 2, 7, 7, 3, 2)
    got = G.resident_encode(clip, 128, 128, 2, 7, 7, 3, 2)
    assert got == want
