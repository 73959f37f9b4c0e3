#!/bin/sh
# The north star's "MFMA tried for a batched 8x8x8 DCT-as-GEMM, kept only if rocprof shows it beating the integer path":
# same box, same build, option dct_mfma off / on (M2V_DCT_MFMA): parity suite with it on, then rocprofv3 kernel stats
# and SQ counters of the P-frame macroblock kernel for both variants, and alternating bench runs.
#   sh tools/mfma_trial.sh <outdir>
export TMPDIR=/tmp
OUT=${1:-gpurun_out/mfma}
mkdir -p $OUT
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
M2V_DCT_MFMA=1 timeout 1800 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_mfma.log 2>&1
echo "pytest(mfma on) rc=$?"; tail -3 $OUT/pytest_mfma.log
for V in 0 1; do
  M2V_DCT_MFMA=$V rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw$V -o s -- python3 bench.py --split 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_rocprof_$V.log 2>&1
  python3 tools/summarize_rocprof.py $OUT/raw$V/s_kernel_stats.csv $OUT/kernel_stats_$V.csv; rm -rf $OUT/raw$V
  M2V_DCT_MFMA=$V rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_I8 --output-format csv -d $OUT/p$V -o p -- python3 bench.py --inflight 1 --split 1 --steps 1 --warmup 1 --prewarm 0 --no-cpu-baseline > /dev/null 2>&1
  python3 tools/summarize_pmc.py $OUT/p$V > $OUT/pmc_$V.json; rm -rf $OUT/p$V
done
for i in 1 2 3; do
  for V in 0 1; do
    M2V_DCT_MFMA=$V python3 bench.py --no-cpu-baseline --steps 200 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('dct_mfma=$V  value %.0f MPixels/s  profiled-pass k_mb_P %.4f ms/step (%.2f us per launch)  k_mb_I %.4f' % (d['value'], d['kernel_ms_per_step']['k_mb_P'], d['roofline']['avg_launch_ms']*1e3, d['kernel_ms_per_step']['k_mb_I']))" | tee -a $OUT/ab.txt
  done
done
python3 - "$OUT" <<'PY'
import json, sys, csv
out = sys.argv[1]
for v in (0, 1):
    d = json.load(open("%s/pmc_%d.json" % (out, v)))
    for n, k in d.items():
        if "k_mb<3, true" in n:
            w = k["SQ_WAVES"]
            print("dct_mfma=%d %s: VALU/MB %.1f  LDS instr/MB %.1f  VALU-active cycles/MB %.0f  LDS-array cycles/MB %.0f (conflicts %.0f)  MFMA i8/MB %.1f" % (
                v, n, k["SQ_INSTS_VALU"] / w, k["SQ_INSTS_LDS"] / w, 4 * k["SQ_ACTIVE_INST_VALU"] / w, k["SQ_LDS_IDX_ACTIVE"] / w, k["SQ_LDS_BANK_CONFLICT"] / w, k.get("SQ_INSTS_VALU_MFMA_I8", 0) / w))
    for r in csv.reader(open("%s/kernel_stats_%d.csv" % (out, v))):
        if r and "k_mb" in r[0]:
            print("   rocprofv3:", ", ".join(r[:6]))
PY
