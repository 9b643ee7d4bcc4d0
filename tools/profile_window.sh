#!/bin/bash
# rocprofv3 evidence for window_lm_kernel on the GPU box:  tools/profile_window.sh SHAPE BATCH OUTDIR [extra bench args]
# (KERNEL=chain_lm_kernel in the environment for batches that take the one-lane-per-window kernel)
#   1. plain run (JSON line)  2. --kernel-trace --stats  3. four --pmc passes (SQ twice, FETCH_SIZE, WRITE_SIZE: separate
#   passes, never combined with a trace domain).  The batch is generated once and cached in /tmp between the passes.
# Summaries land in OUTDIR (under gpurun_out/); copy what should be judged into profiles/.
set -eo pipefail
SHAPE=$1; BATCH=$2; OUT=$3; shift 3
mkdir -p "$OUT"
export TMPDIR=/tmp
CACHE=/tmp/wb_${SHAPE}_${BATCH}.npz
BENCH="python3 tests/perf/bench_window.py --shape $SHAPE --batch $BATCH --cache $CACHE --reps 3 --no-latency $*"
$BENCH --cpu-n 64 > "$OUT/bench_${SHAPE}_${BATCH}.json"
echo "[profile_window] plain run done: $(cat "$OUT/bench_${SHAPE}_${BATCH}.json" | cut -c1-300)"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_${SHAPE}" -o t -- $BENCH > /dev/null
f=$(find "$OUT/trace_${SHAPE}" -name '*kernel_stats.csv' | head -1)
head -5 "$f" > "$OUT/kernel_stats_${SHAPE}_${BATCH}.csv"
echo "[profile_window] kernel-trace done"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P2="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
P5="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM TCC_HIT_sum TCC_MISS_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d "$OUT/pmc${i}_${SHAPE}" -o p -- $BENCH > /dev/null || echo "[profile_window] pmc pass $i failed"
  echo "[profile_window] pmc pass $i done"
done
python3 tests/perf/pmc_summary.py ${KERNEL:-window_lm_kernel} "$OUT"/pmc*_${SHAPE} > "$OUT/pmc_${SHAPE}_${BATCH}.json"
cat "$OUT/pmc_${SHAPE}_${BATCH}.json"
# the same run once more with the counters folded into its roofline record
$BENCH --cpu-n 0 --pmc-json "$OUT/pmc_${SHAPE}_${BATCH}.json" > "$OUT/bench_roofline_${SHAPE}_${BATCH}.json"
cat "$OUT/bench_roofline_${SHAPE}_${BATCH}.json"
rm -rf "$OUT"/trace_${SHAPE} "$OUT"/pmc*_${SHAPE}
