"""-m gpu: the HIP path against the oracle on the hand-built corner clips (tests/corner_clips.py): the SAD kill threshold at
exactly 4095 / 4096, a level after 31 zeros, predictors across alternating intra / inter macroblocks, a vector difference at
the wrap, a stream that ends on a word boundary.  Stage by stage (mode, vectors, levels, reconstruction, bits, bytes)."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("make,arg", [("sad_exact", 4094), ("sad_exact", 4095), ("sad_exact", 4096), ("sad_threshold_flat", 4095),
                                      ("sad_threshold_flat", 4096), ("lone_level_after_31_zeros", None), ("intra_between_inter_macroblocks", None),
                                      ("vector_delta_of_sixteen", None), ("stream_ending_on_a_word_boundary", None)])
def test_corner_clip_stage_parity(make, arg):
    import corner_clips as C
    import gpu_util as G
    if make == "sad_exact":
        clip, pf, VL, Q = C.sad_exact(arg), 1, 1, 2
    elif make == "sad_threshold_flat":
        clip, pf, VL, Q = C.sad_threshold_flat(64, 64, arg), 1, 3, 2
    else:
        clip, pf, VL, Q = getattr(C, make)()
    n, _, H, W = clip.shape
    assert G.compare_stages(clip, W // 16, H // 16, pf, XL=7, YL=7, VL=VL, Q=Q) == []
