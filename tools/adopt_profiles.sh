#!/bin/bash
# Runs HERE (authoring container) after a gpurun call of tools/measure_set.sh / tools/collect_profiles.sh / tools/train_iter_breakdown.sh:
# copies the merged summaries from gpurun_out/ into profiles/ under the tag's name and stamps the traffic file with the commit
# the passes were taken at (bench.py reports it as roofline.traffic_source).   usage: tools/adopt_profiles.sh <tag> [commit]
set -u
TAG=$1
cd "$(dirname "$0")/.."
COMMIT=${2:-$(git rev-parse --short HEAD)}
M=gpurun_out/meas_$TAG; P=gpurun_out/prof_$TAG; I=gpurun_out/train_iter_$TAG
[ -d $M ] && for f in $M/*.json $M/*.jsonl; do [ -s "$f" ] && cp "$f" profiles/${TAG}_$(basename "$f"); done
[ -f $P/summary.txt ] && cp $P/summary.txt profiles/${TAG}_rocprofv3.txt
[ -f $P/sq.json ] && cp $P/sq.json profiles/${TAG}_sq_counters.json
[ -f $P/bench_under_rocprof.json ] && cp $P/bench_under_rocprof.json profiles/${TAG}_bench_under_rocprof.json
[ -f $I/breakdown_sh_fold_1.json ] && cp $I/breakdown_sh_fold_1.json profiles/${TAG}_train_iteration_breakdown.json
[ -f $I/iteration_sh_fold_on.json ] && cp $I/iteration_sh_fold_on.json profiles/${TAG}_train_iteration.json
if [ -f $P/traffic.json ]; then
  python3 - "$P/traffic.json" "profiles/${TAG}_traffic.json" "$COMMIT" "$TAG" <<'PY'
import json, sys
src, dst, commit, tag = sys.argv[1:5]
d = json.load(open(src))
d["_meta"] = {"commit": commit,
              "fetch_correction": "FETCH_SIZE x2 = bytes of the 128-byte lines fetched, for streaming reads AND for 48- / 64-byte gathers "
                                  "(profiles/r4_fetch_calibration.json); WRITE_SIZE as reported",
              "collected_by": f"tools/collect_profiles.sh {tag} (separate rocprofv3 --pmc passes of bench.py --steps 5 --warmup 2)"}
json.dump(d, open(dst, "w"), indent=1)
PY
fi
ls profiles | grep "^${TAG}_" | tr '\n' ' '; echo
