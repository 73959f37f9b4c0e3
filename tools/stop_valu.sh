#!/bin/sh
# VALU / SALU / LDS instructions per macroblock of k_mb by phase, by truncation: the -DM2V_DEBUG library's "ablate" option
# n << 8 ends the kernel at stop point n (M2V_STOP in m2v_kernels.hpp; output invalid, counters meaningful); the difference
# between consecutive stop points is that phase.   sh tools/stop_valu.sh <outdir>
export TMPDIR=/tmp
OUT=${1:-gpurun_out/stops}
mkdir -p $OUT
for N in 1 2 3 4 5 6 0; do
  A=$((N * 256))
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $OUT/s$N -o p -- python3 bench.py --inflight 1 --split 1 --steps 1 --warmup 1 --prewarm 0 --no-cpu-baseline --ablate $A > $OUT/s$N.log 2>&1
  python3 tools/summarize_pmc.py $OUT/s$N > $OUT/s$N.json
  rm -rf $OUT/s$N
done
python3 - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
names = {1: "loads, 4:2:0, window staging", 2: "full-pel search", 3: "half-pel, decision, prediction", 4: "forward transform", 5: "quantiser + inverse quantiser",
         6: "VLC pass 1 + look-up, IDCT", 0: "VLC pass 2, reconstruction and slot stores"}
prev = dict(valu=0, salu=0, lds=0, act=0, ldsc=0, conf=0)
rows = []
for n in (1, 2, 3, 4, 5, 6, 0):
    d = json.load(open("%s/s%d.json" % (out, n)))
    k = max((v for kk, v in d.items() if "k_mb<3, true" in kk), key=lambda v: v["SQ_WAVES"])   # not the one-wave FILL instantiation
    w = k["SQ_WAVES"]
    cur = dict(valu=k["SQ_INSTS_VALU"] / w, salu=k["SQ_INSTS_SALU"] / w, lds=k["SQ_INSTS_LDS"] / w, act=4 * k["SQ_ACTIVE_INST_VALU"] / w,
               ldsc=k["SQ_LDS_IDX_ACTIVE"] / w, conf=k["SQ_LDS_BANK_CONFLICT"] / w)
    rows.append("%-32s VALU %6.1f  SALU %6.1f  LDS instr %5.1f  VALU-active cycles %6.0f  LDS-array cycles %5.0f (bank conflicts %4.0f)   | cumulative VALU %6.1f" % (
        names[n], cur["valu"] - prev["valu"], cur["salu"] - prev["salu"], cur["lds"] - prev["lds"], cur["act"] - prev["act"],
        cur["ldsc"] - prev["ldsc"], cur["conf"] - prev["conf"], cur["valu"]))
    prev = cur
open(out + "/stops.txt", "w").write("\n".join(rows) + "\n")
print("\n".join(rows))
PY
