import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import m2v_load
M = m2v_load.load()
W, H, n, pf = 1920, 1152, 18, 8
clip = M.synth.clip(W, H, n, clip_index=3)
pk = torch.from_numpy(np.ascontiguousarray(np.moveaxis(clip, 1, -1))).pin_memory().numpy().reshape(-1)
enc = M.Mpeg2Encoder(7, 7, 3, 2)
enc.set_option("batch_frames", 9)
for rep in range(6):
    for k in range(0, n, 9):
        enc.push_packed(W // 16, H // 16, pf, pk[k * W * H * 3:(k + 9) * W * H * 3], "yuv24")
        enc.pull(1 << 24)
    enc.sequence_stop()
    enc.pull_all()
enc.close()
print("done")
