"""Ad-hoc robustness run: 300 sequences of varying geometry / GOP length / interface on ONE handle, each checked against the
oracle (buffer re-allocation on geometry growth, stage reuse, option changes between sequences).  usage: python tools/stress_handle.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, m2v_load, torch
from oracle import m2v_oracle_ctypes as orc
M = m2v_load.load()
enc = M.Mpeg2Encoder(7, 7, 3, 2)
rng = np.random.default_rng(9)
cache = {}
t0 = time.time(); bad = 0
for it in range(300):
    W, H = 16 * int(rng.integers(4, 20)), 16 * int(rng.integers(4, 14))
    n, pf = int(rng.integers(1, 6)), int(rng.choice([0, 1, 3, 8]))
    key = (W, H, n, pf, it % 7)
    clip = M.synth.clip(W, H, n, clip_index=it % 7)
    want = orc.encode(clip, W // 16, H // 16, pf, 7, 7, 3, 2)
    mode = it % 3
    if mode == 0:
        got = enc.encode(clip, W // 16, H // 16, pf)
    elif mode == 1:
        d = torch.from_numpy(np.ascontiguousarray(clip)).cuda()
        out = torch.empty(n * W * H * 3 + 65536, dtype=torch.uint8, device="cuda")
        nb = enc.encode_resident(d.data_ptr(), n, out.data_ptr(), out.numel(), W // 16, H // 16, pf)
        got = out[:nb].cpu().numpy().tobytes()
    else:
        enc.set_option("batch_frames", int(rng.integers(1, 5)))
        got = enc.encode(clip, W // 16, H // 16, pf)
        enc.set_option("batch_frames", 96)
    bad += got != want
print("300 sequences on one handle, mismatches:", bad, "%.1f s" % (time.time() - t0), "GPU mem MB", torch.cuda.memory_allocated() >> 20)
enc.close()
