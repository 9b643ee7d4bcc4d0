#!/bin/bash
# rocprofv3 evidence for the headline kernel (snapshot_lm_kernel) and the fusion kernel on the GPU box:
#   tools/profile_bench.sh OUTDIR
# 1. bench.py default run (JSON line)  2. the same under --kernel-trace --stats  3. four --pmc passes of a short numpy-fed run
# (separate passes, never combined with a trace domain)  4. the same three for tests/perf/bench_fusion.py.
set -eo pipefail
OUT=$1
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "[profile_bench] default run done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_bench" -o t -- python3 bench.py --no-cpu-baseline --legs none > "$OUT/bench_under_rocprof.json" 2>/dev/null
head -6 "$(find "$OUT/trace_bench" -name '*kernel_stats.csv' | head -1)" > "$OUT/kernel_stats_head.csv"
SHORT="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --legs none --datagen numpy"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P2="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d "$OUT/pmc${i}_bench" -o p -- $SHORT > /dev/null 2>&1 || echo "[profile_bench] pmc pass $i failed"
done
python3 tests/perf/pmc_summary.py snapshot_lm_kernel "$OUT"/pmc*_bench > "$OUT/pmc_snapshot.json"
cat "$OUT/pmc_snapshot.json"
FUS="python3 tests/perf/bench_fusion.py --steps 2 --cpu-tags 64 --cpu-epochs 4"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_fusion" -o t -- $FUS > "$OUT/bench_fusion_under_rocprof.json" 2>/dev/null
head -4 "$(find "$OUT/trace_fusion" -name '*kernel_stats.csv' | head -1)" > "$OUT/kernel_stats_fusion_head.csv"
i=0
for P in "$P1" "$P2" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d "$OUT/pmc${i}_fusion" -o p -- $FUS > /dev/null 2>&1 || echo "[profile_bench] fusion pmc pass $i failed"
done
python3 tests/perf/pmc_summary.py fusion_lm_kernel "$OUT"/pmc*_fusion > "$OUT/pmc_fusion.json"
cat "$OUT/pmc_fusion.json"
# keep the summaries only: the raw traces and counter CSVs are tens of MB (gpurun copies back at most 64 MiB)
rm -rf "$OUT"/trace_bench "$OUT"/trace_fusion "$OUT"/pmc*_bench "$OUT"/pmc*_fusion
