set -u
cd "${GRAFT_REPO_ROOT:-.}"
bash tools/measure_set.sh r3v1 > gpurun_out/meas_r3v1.log 2>&1
tail -40 gpurun_out/meas_r3v1.log
bash tools/collect_profiles.sh r3v1 > gpurun_out/prof_r3v1.log 2>&1
tail -45 gpurun_out/prof_r3v1.log
