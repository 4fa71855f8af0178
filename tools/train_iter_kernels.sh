#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of the fused training iteration (tools/bench_train_iter.py ONLY=fused), per-kernel
# averages of the library's own kernels -> gpurun_out/train_iter_kernels_<tag>.txt    usage: tools/train_iter_kernels.sh <tag>
set -u
TAG=${1:-run}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=$PWD/gpurun_out/tik_$TAG
mkdir -p "$R"
ONLY=fused timeout 400 rocprofv3 --kernel-trace --stats -d "$R/kt" -o kt -- python tools/bench_train_iter.py > "$R/kt.log" 2>&1
DB=$(find "$R/kt" -name '*_results.db' | head -1)
python - "$DB" > "$PWD/gpurun_out/train_iter_kernels_$TAG.txt" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, count(*), sum(duration), min(duration) from kernels group by name order by sum(duration) desc").fetchall()
print(f"{'kernel':100s} {'calls':>7s} {'avg_us':>9s} {'min_us':>9s} {'ms/iter(130)':>12s}")
for name, calls, total, mn in rows[:40]:
    print(f"{name[:100]:100s} {calls:7d} {total / calls / 1e3:9.2f} {mn / 1e3:9.2f} {total / 1e6 / 130:12.4f}")
PY
rm -rf "$R"
cat "$PWD/gpurun_out/train_iter_kernels_$TAG.txt"
