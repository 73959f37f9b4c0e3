import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G
from oracle import m2v_oracle_ctypes as orc
M = G.M
f = M.synth.clip(128, 96, 9, clip_index=2)
ref_bytes, ref = orc.encode(f, 8, 6, 8, 7, 7, 3, 2, dump=True)
got_bytes, got = G.resident_encode(f, 8, 6, 8, 7, 7, 3, 2, None, debug=True)
mbw = 8
gb = got["mb_bits"].astype(np.int64).copy()
gb.reshape(gb.shape[0], -1, mbw)[:, :, 0] -= 38
rb = ref["mb_bits"].reshape(-1); gbf = gb.reshape(-1)
bad = np.nonzero(rb != gbf)[0]
print("n bad", len(bad), "of", len(rb))
nmb = ref["mb_inter"].shape[1]
for d in bad[:12]:
    fr, mb = divmod(int(d), nmb)
    c = ref["coef"].reshape(-1, 6, 64)[d]
    print("frame", fr, "mb", mb % mbw, mb // mbw, "inter", ref["mb_inter"].reshape(-1)[d], "cbp", ref["mb_cbp"].reshape(-1)[d], "bits", rb[d], gbf[d])
    for t in range(6):
        nz = [(int(i), int(c[t][i])) for i in np.nonzero(c[t])[0]]
        if nz: print("   tile", t, nz)
