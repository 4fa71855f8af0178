set -u
cd "${GRAFT_REPO_ROOT:-.}"
bash tools/measure_set.sh r3v3 > gpurun_out/meas_r3v3.log 2>&1
bash tools/collect_profiles.sh r3v3 > gpurun_out/prof_r3v3.log 2>&1
export TMPDIR=/tmp
R=$PWD/gpurun_out/prof_r3v3
timeout 300 rocprofv3 --kernel-trace --stats -d "$R/kt100" -o kt -- python bench.py --steps 100 --warmup 30 --cpu-sample 0 > "$R/kt100.log" 2>&1
python tools/prof_summary.py "$R/kt100/kt_results.db" > "$R/kt100_summary.txt" 2>&1
grep '^{' "$R/kt100.log" | tail -1 > "$R/bench_under_rocprof_100steps.json"
rm -rf "$R/kt100"
head -20 "$R/kt100_summary.txt"
grep -h "ms/step" gpurun_out/meas_r3v3.log | head -20
