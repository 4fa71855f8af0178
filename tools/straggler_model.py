#!/usr/bin/env python
"""straggler_model.py -- what eight scenes of DIFFERENT size do to "≥ 6x at 8 GPUs" (no 8-GPU node has
been available to any round, so this is a model fed with single-GPU measurements, not a scaling curve).

The path shards by scene (the reference's farm, scripts/run_jax.py:52-87): GPU r trains scene r of N_r
Gaussians, the rasterizer needs no collective. Two figures follow from the single-GPU step time t(N):

  independent scenes (the reference's and our default): the job -- the same number of iterations on every scene -- is done
      when the LARGEST scene is, against sum_r t(N_r) on one GPU:      speed-up  S = sum_r t(N_r) / max_r t(N_r)
  lock step (--shared-mlp, bench.py --gpus 8: one all-reduce per step): every step lasts max_r t(N_r) + t_allreduce:
      bench value  V = sum_r N_r / (max_r t(N_r) + t_ar),   and the same S with t_ar added to the denominator.

t(N) is interpolated (piecewise linear in N; linear extrapolation beyond the ends) through measured bench.py lines
(`--points N:ms,...` or `--bench-json` files). Equal scenes give S = 8 / (1 + t_ar / t): the 24 966-float all-reduce measured
at world size 1 costs 0.015 ms per step (profiles/r4_v7_bench_force_dist_rccl.json), 1.6 % of a 2 M step. The straggler is
what decides: S >= 6 needs  mean_r t(N_r) >= 0.75 max_r t(N_r).

usage: python tools/straggler_model.py [--points 500000:0.355,1000000:0.526,2000000:0.932] [--sizes N0 N1 ... N7]
       [--allreduce-ms 0.015] [--json]
Without --sizes: a sweep over size spreads (scene r has N_max (1 - s r / 7) Gaussians, r = 0..7) that reports the spread
at which S falls below 6.
"""
import argparse
import json
import sys

# bench.py --n N (1920x1080, headline geometry), ms per fwd+bwd step, one MI355X: profiles/r4_v7_bench_{500000,1000000,default}.json
DEFAULT_POINTS = {500_000: 0.3554, 1_000_000: 0.5255, 2_000_000: 0.9318}


def make_t(points):
    xs = sorted(points)
    if len(xs) < 2:
        raise ValueError("need at least two (N, ms) points")

    def t(n):
        lo, hi = xs[0], xs[1]
        for a, b in zip(xs, xs[1:]):
            lo, hi = a, b
            if n <= b:
                break
        return points[lo] + (points[hi] - points[lo]) * (n - lo) / (hi - lo)
    return t


def predict(sizes, t, allreduce_ms=0.0):
    """dict(speedup_independent, speedup_lockstep, bench_value_gaussians_per_s, ...) for one GPU per scene."""
    ts = [t(n) for n in sizes]
    tmax, tsum = max(ts), sum(ts)
    return dict(sizes=list(sizes), ms_per_step=[round(x, 4) for x in ts], straggler=int(max(range(len(ts)), key=ts.__getitem__)),
                speedup_independent=tsum / tmax, speedup_lockstep=tsum / (tmax + allreduce_ms),
                bench_value_gaussians_per_s=sum(sizes) / ((tmax + allreduce_ms) * 1e-3),
                single_gpu_value_of_the_largest=max(sizes) / (tmax * 1e-3),
                mean_over_max_t=tsum / len(ts) / tmax)


def spread_sizes(n_max, spread, world=8):
    return [int(round(n_max * (1.0 - spread * r / (world - 1)))) for r in range(world)]


def critical_spread(t, n_max, target=6.0, world=8, allreduce_ms=0.0, lockstep=False):
    """largest spread s in [0, 1) (scene r has n_max (1 - s r / (world - 1)) Gaussians) whose speed-up still reaches target"""
    key = "speedup_lockstep" if lockstep else "speedup_independent"
    lo, hi = 0.0, 0.999
    if predict(spread_sizes(n_max, hi, world), t, allreduce_ms)[key] >= target:
        return hi
    for _ in range(60):
        mid = (lo + hi) / 2
        if predict(spread_sizes(n_max, mid, world), t, allreduce_ms)[key] >= target:
            lo = mid
        else:
            hi = mid
    return lo


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--points", default="", help="N:ms,N:ms,... measured single-GPU step times (default: the round-4 bench lines)")
    ap.add_argument("--bench-json", nargs="*", default=[], help="bench.py output files to take (N, ms_per_step) from")
    ap.add_argument("--sizes", type=int, nargs="*", default=None)
    ap.add_argument("--allreduce-ms", type=float, default=0.015)
    ap.add_argument("--target", type=float, default=6.0)
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    points = dict(DEFAULT_POINTS)
    if a.points:
        points = {int(p.split(":")[0]): float(p.split(":")[1]) for p in a.points.split(",")}
    for f in a.bench_json:
        d = json.loads([ln for ln in open(f) if ln.startswith("{")][-1])
        points[int(d["config"]["N"])] = float(d["ms_per_step"])
    t = make_t(points)
    out = {"points_ms": {str(k): v for k, v in sorted(points.items())}, "allreduce_ms": a.allreduce_ms, "target": a.target}
    if a.sizes:
        out["prediction"] = predict(a.sizes, t, a.allreduce_ms)
    else:
        n_max = max(points)
        out["equal_scenes"] = predict([n_max] * 8, t, a.allreduce_ms)
        out["sweep"] = [dict(spread=s, smallest=spread_sizes(n_max, s)[-1],
                             **{k: round(v, 3) for k, v in predict(spread_sizes(n_max, s), t, a.allreduce_ms).items()
                                if k.startswith("speedup") or k == "mean_over_max_t"}) for s in (0.0, 0.1, 0.25, 0.5, 0.75)]
        out["critical_spread_independent"] = round(critical_spread(t, n_max, a.target), 3)
        out["critical_spread_lockstep"] = round(critical_spread(t, n_max, a.target, allreduce_ms=a.allreduce_ms, lockstep=True), 3)
        out["the_eight_sizes_0.5M_to_2M"] = predict(spread_sizes(n_max, 0.75), t, a.allreduce_ms)
    if a.json:
        print(json.dumps(out))
        return
    print("single-GPU step time t(N), ms:", out["points_ms"], "| all-reduce per step:", a.allreduce_ms, "ms")
    if a.sizes:
        p = out["prediction"]
        print(f"sizes {p['sizes']}: per-scene ms {p['ms_per_step']}; straggler = scene {p['straggler']}")
        print(f"  8-GPU speed-up over one GPU doing the scenes in turn: independent {p['speedup_independent']:.2f}x, "
              f"lock step {p['speedup_lockstep']:.2f}x; bench value {p['bench_value_gaussians_per_s']:.3g} Gaussians/s")
        return
    e = out["equal_scenes"]
    print(f"8 equal scenes of {max(points)}: independent {e['speedup_independent']:.2f}x, lock step {e['speedup_lockstep']:.2f}x, "
          f"bench value {e['bench_value_gaussians_per_s']:.3g} Gaussians/s")
    for r in out["sweep"]:
        print(f"  spread {r['spread']:.2f} (smallest scene {r['smallest']}): independent {r['speedup_independent']:.2f}x, "
              f"lock step {r['speedup_lockstep']:.2f}x (mean t / max t = {r['mean_over_max_t']:.3f})")
    print(f"'>= {a.target:g}x' holds up to a size spread of {out['critical_spread_independent']:.2f} (independent scenes), "
          f"{out['critical_spread_lockstep']:.2f} (lock step): the smallest scene may be "
          f"{(1 - out['critical_spread_independent']) * 100:.0f} % of the largest when the sizes are spread evenly between them")


if __name__ == "__main__":
    sys.exit(main())
