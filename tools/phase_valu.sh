#!/bin/sh
# VALU / SALU / LDS instructions per macroblock of k_mb by phase: the -DM2V_DEBUG library's "ablate" option switches
# phases off one at a time (output invalid, counters still meaningful); difference to the full kernel = that phase.
#   sh tools/phase_valu.sh <outdir>
export TMPDIR=/tmp
OUT=${1:-gpurun_out/phase}
mkdir -p $OUT
for A in 0 1 2 4 8 16 31; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $OUT/a$A -o p -- python3 bench.py --inflight 1 --split 1 --steps 1 --warmup 1 --prewarm 0 --no-cpu-baseline --ablate $A > $OUT/a$A.log 2>&1
  python3 tools/summarize_pmc.py $OUT/a$A > $OUT/a$A.json
  rm -rf $OUT/a$A
done
python3 - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
names = {0: "full kernel", 1: "without full-pel search", 2: "without half-pel SADs", 4: "without VLC", 8: "without IDCT/recon", 16: "without DCT/quant", 31: "without all five"}
base = None
rows = []
for a in (0, 1, 2, 4, 8, 16, 31):
    d = json.load(open("%s/a%d.json" % (out, a)))
    k = max((v for n, v in d.items() if "k_mb<3, true" in n), key=lambda v: v["SQ_WAVES"])
    w = k["SQ_WAVES"]
    cur = dict(valu=k["SQ_INSTS_VALU"] / w, salu=k["SQ_INSTS_SALU"] / w, lds=k["SQ_INSTS_LDS"] / w, act=4 * k["SQ_ACTIVE_INST_VALU"] / w,
               ldsc=k["SQ_LDS_IDX_ACTIVE"] / w, conf=k["SQ_LDS_BANK_CONFLICT"] / w)
    if base is None:
        base = cur
    rows.append("%-26s VALU %7.1f (phase %6.1f)  SALU %6.1f (%6.1f)  LDS %6.1f (%5.1f)  VALU-active cycles %7.0f (%6.0f)  LDS-array cycles %6.0f (%5.0f) of which bank conflicts %5.0f (%4.0f)" % (
        names[a], cur["valu"], base["valu"] - cur["valu"], cur["salu"], base["salu"] - cur["salu"], cur["lds"], base["lds"] - cur["lds"], cur["act"], base["act"] - cur["act"],
        cur["ldsc"], base["ldsc"] - cur["ldsc"], cur["conf"], base["conf"] - cur["conf"]))
open(out + "/phases.txt", "w").write("\n".join(rows) + "\n")
print("\n".join(rows))
PY
