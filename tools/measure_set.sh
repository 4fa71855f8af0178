#!/bin/bash
# Runs on the GPU box (via gpurun): the round's measurement set -> gpurun_out/meas_<tag>/ (copied to profiles/ by hand).
# usage: tools/measure_set.sh <tag>
set -u
TAG=${1:-run}
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD/gpurun_out/meas_$TAG
mkdir -p "$R"
B="timeout 300 python bench.py --cpu-sample 0"
timeout 400 python bench.py > "$R/bench_default.json" 2> "$R/bench_default.err"
$B --n 500000 > "$R/bench_500000.json" 2>/dev/null
$B --n 1000000 > "$R/bench_1000000.json" 2>/dev/null
$B --config cfg4 > "$R/bench_cfg4.json" 2>/dev/null          # 5 M, 2560x1440, z ~ U(500, 700), dL/ddepth != 0
$B --config cfg3 > "$R/bench_cfg3.json" 2>/dev/null          # 108 forward-only 1024^2 renders + the timed fwd+bwd steps
$B --n 5000000 --width 2560 --height 1440 > "$R/bench_5M_1440p_cfg2geometry.json" 2>/dev/null   # (the r1/r2 "cfg 4" line)
HSA_ENABLE_IPC_MODE_LEGACY=0 $B --force-dist > "$R/bench_force_dist_rccl.json" 2> "$R/bench_force_dist_rccl.err"      # RCCL all-reduce every step, world = 1
$B --sh-degree 1 > "$R/bench_sh1.json" 2>/dev/null
$B --sh-degree 3 > "$R/bench_sh3.json" 2>/dev/null
$B --forward-only > "$R/fps_2M.json" 2>/dev/null
$B --forward-only --config cfg4 > "$R/fps_cfg4.json" 2>/dev/null
$B --forward-only --config cfg3 > "$R/fps_cfg3_1024sq.json" 2>/dev/null
$B --forward-only --n 16000000 > "$R/fps_16M.json" 2>/dev/null
timeout 600 python tools/bench_regimes.py > "$R/regimes.jsonl" 2>/dev/null
timeout 300 python tools/bench_train_iter.py > "$R/train_iteration.json" 2>/dev/null
REAL=1 APPEARANCE=0 timeout 400 python tools/bench_train_iter.py 2>/dev/null | tail -1 > "$R/train_iteration_real_classes.json"     # the reference's own GaussianModel / render()
REAL=1 timeout 400 python tools/bench_train_iter.py 2>/dev/null | tail -1 > "$R/train_iteration_real_classes_appearance.json"   # + --appearance_enabled (its MLP GEMMs: not this path's)
timeout 400 python tools/prof_host.py 1000 100000 500000 2000000 2>/dev/null | grep "^N=" > "$R/host_floor.txt"
timeout 300 python tools/bench_next_rows.py > "$R/next_rows.json" 2> "$R/next_rows.err"
for f in "$R"/*.json "$R"/*.jsonl; do echo "== $f"; python - "$f" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    if "ms_per_step" in d and "roofline_step" in d:
        print(d["config"]["workload"][:60], "| ms/step", round(d["ms_per_step"], 4), "| value", f'{d["value"]:.4g}', d["unit"], "|", d["roofline_step"]["kernel_ms_per_step"])
    else:
        print(json.dumps(d)[:400])
PY
done
