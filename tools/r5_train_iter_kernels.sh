cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; R=$PWD/gpurun_out/train_iter_r5_v6; mkdir -p $R
ONLY=fused timeout 400 rocprofv3 --kernel-trace --stats -d "$R/kt" -o kt -- python tools/bench_train_iter.py > "$R/kt.log" 2>&1
WALL=$(grep '^{' "$R/kt.log" | tail -1 | python -c "import sys, json; print(json.loads(sys.stdin.read())['fused_hooks_ms'])" 2>/dev/null)
DB=$(find "$R/kt" -name '*_results.db' | head -1)
python tools/train_iter_breakdown.py "$DB" 130 ${WALL:-} > "$R/breakdown.json"; rm -rf "$R/kt"
