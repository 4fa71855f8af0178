#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + the two HBM PMC passes (separate runs, as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass) of bench.py, and
# summarises them. usage: tools/collect_profiles.sh <tag>   -> gpurun_out/prof_<tag>/{summary.txt,traffic.json}
set -u
TAG=${1:-run}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=$PWD/gpurun_out/prof_$TAG
mkdir -p "$R"
ARGS="--steps 5 --warmup 2 --cpu-sample 0"
timeout 300 rocprofv3 --kernel-trace --stats -d "$R/kt" -o kt -- python bench.py $ARGS > "$R/kt.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$R/fetch" -o fetch -- python bench.py $ARGS > "$R/fetch.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$R/write" -o write -- python bench.py $ARGS > "$R/write.log" 2>&1
python tools/prof_summary.py "$R/kt/kt_results.db" --pmc "$R/fetch/fetch_results.db" "$R/write/write_results.db" \
    --json "$R/traffic.json" > "$R/summary.txt" 2>&1
grep '^{' "$R/kt.log" | tail -1 > "$R/bench_under_rocprof.json"
rm -rf "$R/kt" "$R/fetch" "$R/write"
tail -40 "$R/summary.txt"
