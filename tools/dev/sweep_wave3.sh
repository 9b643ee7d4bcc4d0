#!/bin/bash
# dev: wave3_lm_kernel against the lane-per-window kernel over batch sizes (cfg1's window as a batch)
for b in ${SWEEP_B:-1024 4096 8192 16384 32768 65536}; do
  for j in numeric analytic; do
    for mn in 100000000 1; do
      LOCAMD_CHAIN_MIN_BATCH=$mn timeout -k 10 200 python tests/perf/bench_window.py --shape uwb_only --batch $b --bw 1 --jacobian $j --cpu-n 8 --tile 2048 --no-latency > /tmp/sw.json 2>/tmp/sw.err || { tail -3 /tmp/sw.err; exit 1; }
      tail -1 /tmp/sw.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['batch'], d['jacobian'], d['roofline']['kernel'], round(d['gpu_kernel_ms_per_batch'],4), 'ms', '%.3g'%d['gpu_windows_per_s_kernel'])"
    done
  done
done
