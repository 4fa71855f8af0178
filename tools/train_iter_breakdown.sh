#!/bin/bash
# Runs on the GPU box: whole-iteration timing with the eval_sh fold on / off, then a rocprofv3 kernel trace of the
# fused iteration grouped per component -> gpurun_out/train_iter_<tag>/   usage: tools/train_iter_breakdown.sh <tag>
set -u
TAG=${1:-run}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=$PWD/gpurun_out/train_iter_$TAG
mkdir -p "$R"
SFGS_SH_FOLD=1 timeout 300 python tools/bench_train_iter.py > "$R/iteration_sh_fold_on.json" 2> "$R/on.err"
SFGS_SH_FOLD=0 ONLY=fused timeout 300 python tools/bench_train_iter.py > "$R/iteration_sh_fold_off.json" 2> "$R/off.err"
for fold in 1 0; do
  SFGS_SH_FOLD=$fold ONLY=fused timeout 400 rocprofv3 --kernel-trace --stats -d "$R/kt$fold" -o kt -- python tools/bench_train_iter.py > "$R/kt$fold.log" 2>&1
  WALL=$(grep '^{' "$R/kt$fold.log" | tail -1 | python -c "import sys, json; print(json.loads(sys.stdin.read())['fused_hooks_ms'])" 2>/dev/null)
  DB=$(find "$R/kt$fold" -name '*_results.db' | head -1)
  python tools/train_iter_breakdown.py "$DB" 130 ${WALL:-} > "$R/breakdown_sh_fold_$fold.json" 2> "$R/breakdown_$fold.err"
  rm -rf "$R/kt$fold" "$R/kt$fold.log"
done
cat "$R"/iteration_*.json; cat "$R/breakdown_sh_fold_1.json"
