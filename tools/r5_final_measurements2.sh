#!/bin/bash
# round 5 (gpurun): the whole -m gpu suite on the final tree, the round's measurement set and the rocprofv3 / PMC passes
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r5c34; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -6 $O/tests.txt | cut -c1-600
timeout 1500 tools/measure_set.sh r5_v4 > $O/measure.log 2>&1; tail -45 $O/measure.log | cut -c1-400
timeout 1500 tools/collect_profiles.sh r5_v4 > $O/prof.log 2>&1; tail -40 $O/prof.log | cut -c1-300
BENCH_EXTRA="--steps 100 --warmup 30" timeout 600 tools/collect_profiles.sh r5_v4_100steps "kt" > $O/prof100.log 2>&1; tail -15 $O/prof100.log | cut -c1-300
timeout 1200 bash tools/train_iter_breakdown.sh r5_v4 > $O/train_iter.log 2>&1; tail -3 $O/train_iter.log | cut -c1-300
