#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration for the compositing kernels' access patterns (runs on the GPU box):
# tools/microbench/fetch_bench -- kernels whose compulsory byte counts are exact -- under the same two PMC passes as
# tools/collect_profiles.sh. Writes gpurun_out/fetch_cal/calibration.json: per kernel the raw counter bytes and the
# factors that would make them read the useful bytes / the 64-byte-sector bytes / the 128-byte-line bytes.
# (tools/prof_summary.py doubles FETCH_SIZE for every kernel, which MI355X_MICROARCH.md establishes for wide streaming
# reads only; the gather factors found here decide how the traffic column of the bench line should be corrected.)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=$PWD/gpurun_out/fetch_cal; mkdir -p "$R"
BIN=tools/microbench/fetch_bench
[ -x $BIN ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/microbench/fetch_bench.hip -o $BIN
$BIN > "$R/expected.jsonl" 2> "$R/run.err"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$R/fetch" -o fetch -- $BIN > "$R/fetch.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$R/write" -o write -- $BIN > "$R/write.log" 2>&1
python - "$R" <<'PY'
import json, re, sqlite3, sys
R = sys.argv[1]
exp = {}
for line in open(f"{R}/expected.jsonl"):
    if line.startswith("{"):
        d = json.loads(line); exp[d["kernel"]] = d
def counters(db):
    out = {}
    cur = sqlite3.connect(db).cursor()
    for name, cn, v in cur.execute("select name, counter_name, avg(counter_value) from pmc_events group by name, counter_name"):
        m = re.search(r"(wscatter48v|wscatter48|wstream16|stream16|stream4|gather<\d, \d>)", name)
        if not m:
            continue
        out.setdefault(m.group(1), {})[cn] = v * 1024.0   # rocprofv3 reports KiB
    return out
f, w = counters(f"{R}/fetch/fetch_results.db"), counters(f"{R}/write/write_results.db")
res = {}
for k, e in exp.items():
    key = {"gather48": "gather<3, 0>", "gather48_coh": "gather<3, 1>", "gather64": "gather<4, 0>"}.get(k, k)
    raw_f, raw_w = f.get(key, {}).get("FETCH_SIZE"), w.get(key, {}).get("WRITE_SIZE")
    raw = raw_w if k.startswith("w") else raw_f
    res[k] = {"raw_counter_bytes": raw, **{n: e[n] for n in ("useful_bytes", "bytes_64B_sectors", "bytes_128B_lines", "ms")}}
    if raw:
        res[k].update({"factor_to_useful": e["useful_bytes"] / raw, "factor_to_sectors": e["bytes_64B_sectors"] / raw,
                       "factor_to_lines": e["bytes_128B_lines"] / raw})
json.dump(res, open(f"{R}/calibration.json", "w"), indent=1)
for k, v in res.items():
    print(k, {n: (round(x, 3) if isinstance(x, float) and x < 100 else x) for n, x in v.items()})
PY
rm -rf "$R/fetch" "$R/write"
