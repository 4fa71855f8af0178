#!/usr/bin/env python
"""Summarise rocprofv3 rocpd (SQLite) outputs into the text tables committed under profiles/.

usage: prof_summary.py kernel-trace.db [--pmc fetch.db write.db ...]
Prints per-kernel launch count / average / total duration (the `--stats` view) and, for PMC databases,
per-kernel average counter values. FETCH_SIZE / WRITE_SIZE are reported in KiB as rocprofv3 gives them and
as bytes per launch; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 counts 64 B per 128-B
request for wide coalesced reads, so `fetch_bytes_corrected` = 2 x FETCH_SIZE x 1024 (upper bound for
narrow/gather access patterns, which the guide leaves uncalibrated)."""
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"(?:sfgs::)?([A-Za-z_0-9]+(?:<\d+>)?)\(", name)
    return m.group(1) if m else name[:60]


def kernel_stats(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, count(*), avg(duration), sum(duration), min(duration), max(duration) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[3] for r in rows) or 1
    print(f"# kernel trace: {path}")
    print(f"{'kernel':44s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_us':>11s} {'pct':>6s}")
    for n, c, a, s, mn, mx in rows:
        print(f"{short(n):44s} {c:6d} {a/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {s/1e3:11.2f} {100*s/tot:6.2f}")
    print()


def kernel_stats_by_grid(path, pattern):
    """average duration per (kernel, launch grid): separates the image sizes of one kernel in a trace of several"""
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    grid = [c for c in cols if "grid" in c.lower()]
    if not grid:
        print(f"# by grid: no grid column among {cols}")
        return
    sel = ", ".join(grid)
    rows = cur.execute(f"select name, {sel}, count(*), avg(duration), min(duration) from kernels group by name, {sel} "
                       "order by name, avg(duration)").fetchall()
    print(f"# per launch grid ({sel}) -- kernels matching '{pattern}'")
    for r in rows:
        if pattern in r[0]:
            print(f"{short(r[0]):32s} grid {tuple(r[1:1 + len(grid)])!s:24s} calls {r[-3]:5d}  avg_us {r[-2] / 1e3:9.2f}  min_us {r[-1] / 1e3:9.2f}")
    print()


def gap_stats(path):
    """Idle time between consecutive kernels on the device (dispatch gaps of dependent launches): end of one kernel to
    the start of the next, over the whole trace; gaps above 50 us (host-side pauses between steps) are left out."""
    cur = sqlite3.connect(path).cursor()
    try:
        rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    except sqlite3.Error:
        return
    gaps = []
    by_next = {}
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        g = (s1 - e0) / 1e3
        if 0 <= g < 50:
            gaps.append(g)
            by_next.setdefault(short(n1), []).append(g)
    if not gaps:
        return
    # per STEP (= per launch of the dominant backward kernel): time with at least one kernel running vs idle time, over the
    # steady part of the trace (between the first and the last composite_bwd launch) -- the answer to "does the GPU ever wait
    # for the host" (kernels of one stream may overlap at their edges: the union of the intervals is what counts)
    steps = [(s0, e0) for n0, s0, e0 in rows if "composite_bwd" in n0]
    if len(steps) > 10:
        t0, t1 = steps[5][0], steps[-1][0]      # from the 6th step's backward to the last step's backward
        busy, cur_s, cur_e = 0, None, None
        for n0, s0, e0 in rows:
            if e0 <= t0 or s0 >= t1:
                continue
            s0, e0 = max(s0, t0), min(e0, t1)
            if cur_e is None or s0 > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                cur_s, cur_e = s0, e0
            else:
                cur_e = max(cur_e, e0)
        if cur_e is not None:
            busy += cur_e - cur_s
        nsteps = len(steps) - 1 - 5
        print(f"# steady state, {nsteps} steps: {(t1 - t0) / nsteps / 1e3:.1f} us per step, a kernel running {busy / nsteps / 1e3:.1f} us of it "
              f"-> GPU idle {(t1 - t0 - busy) / nsteps / 1e3:.1f} us per step ({100.0 * (t1 - t0 - busy) / (t1 - t0):.1f} %)")
    gaps.sort()
    print(f"# dispatch gaps between consecutive kernels (us): n={len(gaps)} median={gaps[len(gaps)//2]:.2f} "
          f"mean={sum(gaps)/len(gaps):.2f} p90={gaps[int(0.9*len(gaps))]:.2f}")
    print(f"{'gap in front of':44s} {'n':>6s} {'median_us':>10s} {'mean_us':>10s}")
    for n, v in sorted(by_next.items(), key=lambda kv: -len(kv[1])):
        v.sort()
        print(f"{n:44s} {len(v):6d} {v[len(v)//2]:10.2f} {sum(v)/len(v):10.2f}")
    print()


TRAFFIC = {}
SQ = {}   # kernel -> {counter: average per launch}
ALL_KERNELS = False   # --all-kernels: also kernels outside namespace sfgs (the microbenchmarks of the SQ calibration)
# Calibration of the SQ-derived ratios: the raw formulas below are scaled so that a kernel that
# saturates a pipe reads 1.00 -- tools/calibrate_sq.sh runs tools/microbench/issue_bench (pure v_fma_f32 / pure ds_read_b64
# loops at 4 waves per SIMD) under the same PMC passes and writes these factors to profiles/r3_sq_calibration.json.
CAL = {"valu_util": 1.0, "lds_busy": 1.0}


def pmc_stats(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, counter_name, count(*), avg(counter_value), avg(duration) from pmc_events "
                       "group by name, counter_name order by avg(counter_value) desc").fetchall()
    print(f"# pmc: {path}")
    print(f"{'kernel':44s} {'counter':12s} {'calls':>6s} {'avg_value':>14s} {'bytes/launch':>14s} {'x2 corrected':>14s} {'avg_us':>9s}")
    for n, cn, c, v, d in rows:
        is_sz = cn in ("FETCH_SIZE", "WRITE_SIZE")
        b = v * 1024 if is_sz else float("nan")
        corr = 2 * b if cn == "FETCH_SIZE" else b
        print(f"{short(n):44s} {cn:12s} {c:6d} {v:14.1f} {b:14.0f} {corr:14.0f} {d/1e3:9.2f}")
        if is_sz and "sfgs" in n:
            TRAFFIC.setdefault(short(n), {})[cn] = corr
        if cn.startswith("SQ_") and ("sfgs" in n or ALL_KERNELS):
            e = SQ.setdefault(short(n), {})
            e[cn] = v
            e.setdefault("avg_us", d / 1e3)
    print()


def sq_derived():
    """One number each per kernel from the SQ passes (MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* /
    SQ_ACTIVE_INST_* count quad-cycles summed over waves; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES):
      valu_util      = CAL x SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CU_CYCLES x 4 SIMDs)   share of SIMD issue time doing VALU
      lds_busy       = CAL x SQ_ACTIVE_INST_LDS x 4 / (SQ_BUSY_CU_CYCLES x 4)         share of the CU's LDS pipe time
                       (CAL: --cal profiles/r3_sq_calibration.json, from the saturating microbenchmarks; 1.0 = uncalibrated)
      valu_of_wave   = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES                      share of a wave's lifetime issuing VALU
      wait_any       = SQ_WAIT_ANY / SQ_WAVE_CYCLES                              parked on s_waitcnt / barrier
      wait_inst      = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES                         issue stalls (pipe busy)
      lds_stall      = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES
      lds_conflict   = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE                  extra LDS cycles from bank conflicts
      occupancy      = SQ_WAVE_CYCLES x 4 / (SQ_BUSY_CU_CYCLES x 4 SIMDs)        average resident waves per SIMD
      valu_per_wave  = SQ_INSTS_VALU / SQ_WAVES"""
    out = {}
    print("# derived SQ metrics (per kernel, averages per launch)")
    hdr = ["kernel", "avg_us", "occup/SIMD", "VALU util", "LDS busy", "VALU/wave-life", "wait_any", "wait_inst", "lds_stall",
           "lds_conflict", "VALU/wave", "LDS/wave", "SALU/wave"]
    print(" ".join(f"{h:>14s}" if i else f"{h:32s}" for i, h in enumerate(hdr)))
    for k, c in SQ.items():
        g = lambda n: c.get(n, float("nan"))
        wc, bcu = g("SQ_WAVE_CYCLES"), g("SQ_BUSY_CU_CYCLES")
        waves = g("SQ_WAVES")
        d = dict(avg_us=c.get("avg_us"), occupancy_waves_per_simd=wc * 4 / (bcu * 4) if bcu == bcu else float("nan"),
                 valu_util=CAL["valu_util"] * g("SQ_ACTIVE_INST_VALU") * 4 / (bcu * 4),
                 valu_util_raw=g("SQ_ACTIVE_INST_VALU") * 4 / (bcu * 4),
                 lds_busy=CAL["lds_busy"] * g("SQ_ACTIVE_INST_LDS") * 4 / (bcu * 4),
                 lds_busy_raw=g("SQ_ACTIVE_INST_LDS") * 4 / (bcu * 4),
                 valu_of_wave=g("SQ_ACTIVE_INST_VALU") / wc,
                 wait_any=g("SQ_WAIT_ANY") / wc, wait_inst=g("SQ_WAIT_INST_ANY") / wc,
                 lds_stall=g("SQ_WAIT_INST_LDS") / wc,
                 lds_conflict=g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") else float("nan"),
                 valu_per_wave=g("SQ_INSTS_VALU") / waves, lds_per_wave=g("SQ_INSTS_LDS") / waves,
                 salu_per_wave=g("SQ_INSTS_SALU") / waves, raw=c)
        out[k] = d
        vals = [d["avg_us"], d["occupancy_waves_per_simd"], d["valu_util"], d["lds_busy"], d["valu_of_wave"], d["wait_any"],
                d["wait_inst"], d["lds_stall"], d["lds_conflict"], d["valu_per_wave"], d["lds_per_wave"], d["salu_per_wave"]]
        print(f"{k:32s} " + " ".join(f"{v:14.3f}" if v is not None else f"{'':14s}" for v in vals))
    print()
    return out


if __name__ == "__main__":
    args = sys.argv[1:]
    if "--all-kernels" in args:
        ALL_KERNELS = True
        args.remove("--all-kernels")
    if "--cal" in args:
        i = args.index("--cal")
        import json as _json
        CAL.update(_json.load(open(args[i + 1]))["factors"])
        args = args[:i] + args[i + 2:]
    sq_json = None
    if "--sq-json" in args:
        i = args.index("--sq-json")
        sq_json = args[i + 1]
        args = args[:i] + args[i + 2:]
    json_out = None
    if "--json" in args:
        i = args.index("--json")
        json_out = args[i + 1]
        args = args[:i] + args[i + 2:]
    by_grid = None
    if "--by-grid" in args:
        i = args.index("--by-grid")
        by_grid = args[i + 1]
        args = args[:i] + args[i + 2:]
    if "--pmc" in args:
        i = args.index("--pmc")
        kts, pmcs = args[:i], args[i + 1:]
    else:
        kts, pmcs = args, []
    for p in kts:
        kernel_stats(p)
        if by_grid:
            kernel_stats_by_grid(p, by_grid)
        gap_stats(p)
    for p in pmcs:
        pmc_stats(p)
    derived = sq_derived() if SQ else {}
    if sq_json and derived:
        import json
        with open(sq_json, "w") as f:
            json.dump(derived, f, indent=1)
    if json_out:
        import json
        with open(json_out, "w") as f:
            json.dump({k: dict(v, hbm_bytes_per_launch=sum(v.values())) for k, v in TRAFFIC.items()}, f, indent=1)
