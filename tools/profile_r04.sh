#!/bin/bash
# Round-4 rocprofv3 evidence on the GPU box:  tools/profile_r04.sh OUTDIR [cfg2|cfg3|cfg5|cfg4|cfg1|twist|t500|all ...]   (run from the repo root; summaries land
# in OUTDIR: copy them to profiles/r04_final/).  Per workload: --kernel-trace --stats once, then separate --pmc passes (never combined with a trace
# domain).  Every window workload runs in the REFERENCE's configuration (numeric Jacobians), the snapshot / fusion kernels in both modes.
#   cfg2  headline snapshot kernel (bench.py itself: numeric = `value`, and --jacobian analytic)        cfg3  fusion kernel
#   cfg5  tree_wave_kernel, 16 384 windows       cfg4  arrow3_lm_kernel, 1 024 and 128 hypotheses       cfg1  wave3_lm_kernel, 65 536 windows + the node's launches
set -eo pipefail
OUT=$1; shift
WHAT="${*:-all}"
want() { [[ " $WHAT " == *" all "* || " $WHAT " == *" $1 "* ]]; }
mkdir -p "$OUT"
export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P2="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
P5="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM TCC_HIT_sum TCC_MISS_sum"
profile() {   # profile TAG KERNEL_SUBSTRING -- command ...
  local TAG=$1 KER=$2; shift 3
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$TAG" -o t -- "$@" > "$OUT/${TAG}_under_rocprof.json" 2>/dev/null
  head -6 "$(find "$OUT/trace_$TAG" -name '*kernel_stats.csv' | head -1)" > "$OUT/${TAG}_kernel_stats_head.csv"
  local i=0
  for P in "$P1" "$P2" "FETCH_SIZE" "WRITE_SIZE" "$P5"; do
    i=$((i+1))
    rocprofv3 --pmc $P --output-format csv -d "$OUT/pmc${i}_$TAG" -o p -- "$@" > /dev/null 2>&1 || echo "[profile_r04] $TAG pmc pass $i failed"
  done
  python3 tests/perf/pmc_summary.py "$KER" "$OUT"/pmc*_"$TAG" > "$OUT/${TAG}_pmc.json"
  if [ -n "$ALSO_TAG" ]; then python3 tests/perf/pmc_summary.py "$ALSO_KER" "$OUT"/pmc*_"$TAG" > "$OUT/${ALSO_TAG}_pmc.json"; cp "$OUT/${TAG}_kernel_stats_head.csv" "$OUT/${ALSO_TAG}_kernel_stats_head.csv"; fi
  rm -rf "$OUT/trace_$TAG" "$OUT"/pmc*_"$TAG"
  echo "[profile_r04] $TAG done: $(head -2 "$OUT/${TAG}_kernel_stats_head.csv" | tail -1 | cut -c1-160)"
}
SHORT="--steps 12 --warmup 3 --no-cpu-baseline --legs none --datagen numpy"   # (enough launches for the trace average to be a warm one)
if want cfg2; then
  profile cfg2_numeric snapshot_lm_kernel -- python3 bench.py $SHORT
  profile cfg2_analytic snapshot_lm_kernel -- python3 bench.py $SHORT --jacobian analytic
fi
if want cfg3; then   # the bench's own cfg3 leg (both modes in one run: fusion_lm_kernel<1> = numeric = `value`, <0> = analytic)
  ALSO_TAG=cfg3_analytic ALSO_KER="fusion_lm_kernel<0>" profile cfg3_numeric "fusion_lm_kernel<1>" -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --legs cfg3 --datagen numpy
fi
W="tests/perf/bench_window.py --reps 3 --no-latency --cpu-n 0 --jacobian numeric"
if want cfg5; then
  python3 $W --shape pose64 --batch 16384 --cache /tmp/wb_p64.npz --bw 8 > /dev/null
  profile cfg5 tree_wave_kernel -- python3 $W --shape pose64 --batch 16384 --cache /tmp/wb_p64.npz --bw 8
fi
if want cfg4; then
  python3 $W --shape selfcal --batch 1024 --tile 32 --cache /tmp/wb_sc1024.npz > /dev/null
  profile cfg4_1024 arrow3_lm_kernel -- python3 $W --shape selfcal --batch 1024 --cache /tmp/wb_sc1024.npz
  python3 $W --shape selfcal --batch 128 --tile 32 --cache /tmp/wb_sc128.npz > /dev/null
  profile cfg4_128 arrow3_lm_kernel -- python3 $W --shape selfcal --batch 128 --cache /tmp/wb_sc128.npz
fi
if want twist; then   # cfg/uwb_twist.yaml's 15-pose window (EdgeSE3 between consecutive poses) as a batch: wave6_lm_kernel<JAC, SE3>
  python3 $W --shape uwb_twist --batch 65536 --tile 1024 --cache /tmp/wb_tw.npz > /dev/null
  profile twist_windows wave6_lm_kernel -- python3 $W --shape uwb_twist --batch 65536 --cache /tmp/wb_tw.npz
fi
if want t500; then   # the node's 500-pose key-frame window (cfg/uwb_pose.yaml at its own trajectory_length): window_lm_kernel, eight waves
  profile node_pose_T500 window_lm_kernel -- python3 tools/dev/node_pose_T500.py
fi
if want cfg1; then
  python3 $W --shape uwb_only --batch 65536 --tile 4096 --cache /tmp/wb_t10.npz --bw 1 > /dev/null
  profile cfg1_windows wave3_lm_kernel -- python3 $W --shape uwb_only --batch 65536 --cache /tmp/wb_t10.npz --bw 1
  # the node's own launches (one ten-pose window per range message of the example recording): kernel trace only
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_node" -o t -- python3 tools/dev/node_latency2.py > "$OUT/cfg1_node_under_rocprof.txt" 2>/dev/null
  head -4 "$(find "$OUT/trace_node" -name '*kernel_stats.csv' | head -1)" > "$OUT/cfg1_node_kernel_stats_head.csv"
  rm -rf "$OUT/trace_node"
fi
ls -la "$OUT"
