"""Quick GPU bring-up script: python tools/gpu_debug.py  (prints stage mismatches vs the oracle)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G  # noqa: E402

cases = [
    ("gray64 I", G.M.synth.degenerate("gray", 64, 64, 1), 4, 4, 0, dict(XL=6, YL=6)),
    ("clip 128x96 I-only", G.M.synth.clip(128, 96, 2, 1), 8, 6, 0, dict()),
    ("clip 128x96 IPPP", G.M.synth.clip(128, 96, 5, 2), 8, 6, 4, dict()),
    ("clip 160x128 VL1 Q1", G.M.synth.clip(160, 128, 4, 3), 10, 8, 3, dict(VL=1, Q=1)),
    ("clip 160x128 VL2 Q3", G.M.synth.clip(160, 128, 4, 4), 10, 8, 3, dict(VL=2, Q=3)),
    ("noise 96x64", G.M.synth.degenerate("noise", 96, 64, 3), 6, 4, 2, dict()),
]
for name, f, xs, ys, pf, kw in cases:
    t = time.time()
    try:
        p = G.compare_stages(f, xs, ys, pf, **kw)
    except Exception as ex:  # noqa: BLE001
        p = ["EXCEPTION %r" % ex]
    print("%-28s %s (%.2fs)" % (name, "PARITY" if not p else "MISMATCH", time.time() - t))
    for line in p:
        print("     ", line)

# coverage statistics of the last P case, from the oracle dumps
import numpy as np  # noqa: E402
from oracle import m2v_oracle_ctypes as orc  # noqa: E402
for name, f, xs, ys, pf, kw in cases[2:]:
    b, d = orc.encode(f, xs, ys, pf, dump=True, **{k: v for k, v in kw.items()})
    inter = d["mb_inter"][1:]
    mvx, mvy = d["mb_mvx"][1:], d["mb_mvy"][1:]
    print("%-24s bytes %6d  inter %.2f  mv!=0 %.2f  halfpel %.2f  |lvl|>40 %d  cbp0 %d" % (
        name, len(b), inter.mean(), ((mvx != 0) | (mvy != 0)).mean(), ((mvx & 1) | (mvy & 1)).astype(bool).mean(),
        int((np.abs(d["coef"]) > 40).sum()), int(((d["mb_cbp"][1:] == 0) & (inter == 1)).sum())))
