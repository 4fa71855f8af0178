#!/bin/bash
# One gpurun call of round 3's last session: validation of HEAD + the one-wave-workgroup A/B. Stages in priority order,
# every stage logs to gpurun_out/r3f/ as it finishes (the call may be cut by the GPU budget).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=$PWD/gpurun_out/r3f; mkdir -p "$R"
T0=$(date +%s); stamp() { echo "$(( $(date +%s) - T0 )) s: $*" | tee -a "$R/timeline.txt"; }
WG1=$PWD/skyfall-gs_amd/sfgs/_exp/lib_wg1.so
stamp start
timeout 500 python -m pytest tests -m gpu -x -q > "$R/pytest_head.log" 2>&1; stamp "pytest HEAD rc=$? $(tail -1 $R/pytest_head.log)"
timeout 120 python tools/bitcompare.py > "$R/bits_base.json" 2> "$R/bits_base.err"; stamp "bits base rc=$?"
for v in wg1 wg2; do
  SFGS_LIB=$PWD/skyfall-gs_amd/sfgs/_exp/lib_$v.so timeout 120 python tools/bitcompare.py > "$R/bits_$v.json" 2> "$R/bits_$v.err"; stamp "bits $v rc=$?"
  python tools/bitcompare.py --diff "$R/bits_base.json" "$R/bits_$v.json" > "$R/bits_diff_$v.txt" 2>&1; stamp "bits diff $v: $(cat $R/bits_diff_$v.txt)"
done
ROUNDS=2 timeout 300 tools/ab.sh skyfall-gs_amd/sfgs/libsfgs.so skyfall-gs_amd/sfgs/_exp/lib_wg1.so skyfall-gs_amd/sfgs/_exp/lib_wg2.so > "$R/ab_wg.txt" 2>&1; stamp "ab done"; cat "$R/ab_wg.txt"
ROUNDS=2 timeout 200 tools/ab.sh skyfall-gs_amd/sfgs/libsfgs.so skyfall-gs_amd/sfgs/_exp/lib_wg1.so -- --forward-only > "$R/ab_wg1_fwd.txt" 2>&1; stamp "ab fwd-only done"; cat "$R/ab_wg1_fwd.txt"
timeout 200 tools/collect_profiles.sh r3v4 "kt" > "$R/prof_kt.txt" 2>&1; stamp "kt profile done"
SFGS_LIB=$WG1 timeout 500 python -m pytest tests -m gpu -x -q > "$R/pytest_wg1.log" 2>&1; stamp "pytest wg1 rc=$? $(tail -1 $R/pytest_wg1.log)"
timeout 300 tools/collect_profiles.sh r3v4pmc "fetch write" > "$R/prof_pmc.txt" 2>&1; stamp "pmc profile done"
timeout 300 python bench.py --steps 20 --warmup 5 > "$R/bench_driver_like.json" 2> "$R/bench_driver_like.err"; stamp "bench rc=$?"
stamp end
