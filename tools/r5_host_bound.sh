#!/bin/bash
# round 5 (gpurun): does the GPU ever wait for the host? bench.py's host_bound block over scene sizes, and rocprofv3 kernel
# traces (ground truth: union of the kernel intervals per step) at 2 M, 500 k, 100 k and 1 k Gaussians
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r5c8; mkdir -p $O
for n in 1000 20000 100000 500000 1000000 2000000; do
  timeout 300 python bench.py --cpu-sample 0 --n $n --steps 60 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); h = d['roofline_step']['host_bound']
print('N=$n', 'ms_per_step', round(d['ms_per_step'], 4), 'gpu_span', h['gpu_span_ms_per_step'], 'gap', h['gpu_gap_between_steps_ms'], 'busy_lower_bound', d['roofline_step']['gpu_busy_ms_per_step'])"
done > $O/host_bound.txt 2>&1; cat $O/host_bound.txt
for n in 2000000 500000 100000 1000; do
  BENCH_EXTRA="--n $n --steps 100 --warmup 10" timeout 400 tools/collect_profiles.sh r5_hb_$n "kt" > /dev/null 2>&1
  echo "== N=$n"; grep -A1 "steady state" gpurun_out/prof_r5_hb_$n/summary.txt | head -3; grep "dispatch gaps" gpurun_out/prof_r5_hb_$n/summary.txt
done > $O/kernel_trace_idle.txt 2>&1; cat $O/kernel_trace_idle.txt
