#!/usr/bin/env python
"""Per-component breakdown of one training iteration: groups the kernels of a rocprofv3 kernel trace
of `ONLY=fused python tools/bench_train_iter.py` by the component of the iteration they belong to.

usage (GPU box): tools/train_iter_breakdown.py <kt_results.db> <iterations> [wall_ms_per_iteration]
The trace holds the bench's 30 warm-up + 100 timed iterations (+ the model set-up): per-iteration figures are totals / 130;
set-up kernels (a handful of launches) end up in `other`."""
import json
import re
import sqlite3
import sys

GROUPS = [   # first match wins
    ("raster_forward", r"sfgs::(preprocess_kernel|bin_|big_walk|plan_scan|fine_bin|select_sort|sort_tiles|composite_fwd|subpix)|fillBufferAligned"),
    ("raster_backward", r"sfgs::(composite_bwd|dupgrad_|preprocess_bwd)"),
    ("adam", r"sfgs::adam"),
    ("ssim_loss", r"sfgs::ssim"),
    ("eval_sh", r"sfgs::sh_eval"),
    ("activations_prepass", r"sfgs::prepass"),
    ("densification_stats", r"sfgs::densify_stats"),
    ("torch_elementwise_and_reductions (L1, depth term, dir_pp normalisation, autograd glue)", r"at::native|at::cuda|elementwise|reduce_kernel|copyBuffer|CatArray"),
]


def main():
    db, iters = sys.argv[1], int(sys.argv[2])
    wall = float(sys.argv[3]) if len(sys.argv) > 3 else None
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(duration) from kernels group by name").fetchall()
    out = {}
    detail = {}
    for name, calls, total in rows:
        grp = next((g for g, pat in GROUPS if re.search(pat, name)), "other")
        e = out.setdefault(grp, {"ms_per_iteration": 0.0, "launches_per_iteration": 0.0})
        e["ms_per_iteration"] += total / 1e6 / iters
        e["launches_per_iteration"] += calls / iters
        detail.setdefault(grp, []).append((total / 1e6 / iters, calls / iters, name[:90]))
    busy = sum(e["ms_per_iteration"] for e in out.values())
    res = {"iterations": iters, "gpu_busy_ms_per_iteration": round(busy, 4), "wall_ms_per_iteration": wall,
           "components": {k: {"ms_per_iteration": round(v["ms_per_iteration"], 4),
                              "launches_per_iteration": round(v["launches_per_iteration"], 2),
                              "share_of_busy": round(v["ms_per_iteration"] / busy, 4)}
                          for k, v in sorted(out.items(), key=lambda kv: -kv[1]["ms_per_iteration"])}}
    non_raster = [(v["ms_per_iteration"], k) for k, v in out.items() if not k.startswith("raster_")]
    res["largest_non_raster_component"] = max(non_raster)[1] if non_raster else None
    res["top_kernels_of_torch_group"] = [
        {"ms_per_iteration": round(a, 4), "launches_per_iteration": round(b, 2), "kernel": n}
        for a, b, n in sorted(detail.get(GROUPS[-1][0], []), reverse=True)[:8]]
    res["library_kernels"] = [
        {"ms_per_iteration": round(a, 4), "launches_per_iteration": round(b, 2), "kernel": re.sub(r"^void ", "", n)[:70]}
        for g in out if g != GROUPS[-1][0] for a, b, n in sorted(detail.get(g, []), reverse=True)]
    res["library_kernels"].sort(key=lambda r: -r["ms_per_iteration"])
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
