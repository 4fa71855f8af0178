set -u
cd "${GRAFT_REPO_ROOT:-.}"
bash tools/measure_set.sh r3v2 > gpurun_out/meas_r3v2.log 2>&1
bash tools/collect_profiles.sh r3v2 > gpurun_out/prof_r3v2.log 2>&1
tail -40 gpurun_out/meas_r3v2.log
