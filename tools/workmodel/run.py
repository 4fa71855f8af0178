#!/usr/bin/env python
"""Work model of the compositing backward on the headline scene (design tool; CPU only).
usage: python tools/workmodel/run.py [--n 2000000] [--window 24]"""
import argparse, ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "skyfall-gs_amd"))
from sfgs.synth import city_scene, orbit_scene, scene  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libworkmodel.so")


class WmFrame(C.Structure):
    _fields_ = [("W", C.c_int32), ("H", C.c_int32), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
                ("kernel_size", C.c_float), ("scale_modifier", C.c_float), ("bg", C.c_void_p), ("view", C.c_void_p),
                ("proj", C.c_void_p), ("campos", C.c_void_p)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--window", type=int, default=24, help="window of window x window tiles at the image centre")
    ap.add_argument("--pitch", type=float, default=0.0)
    ap.add_argument("--city", type=float, default=0.0, help="tools/bench_regimes.py's opaque city at this elevation (degrees)")
    ap.add_argument("--orbit", type=float, default=0.0, help="... its orbit view at this elevation")
    a = ap.parse_args()
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", SO,
                           os.path.join(HERE, "workmodel.cpp")])
    lib = C.CDLL(SO)
    kw = dict(pitch_deg=a.pitch, zrange=(40.0, 400.0)) if a.pitch else {}   # --pitch 45: tools/bench_regimes.py's low elevation
    if a.city:
        frame, g = city_scene(a.n, a.width, a.height, a.city, seed=0)
    elif a.orbit:
        frame, g = orbit_scene(a.n, a.width, a.height, a.orbit, seed=0)
    else:
        frame, g = scene(a.n, a.width, a.height, seed=0, **kw)
    f32 = lambda t: np.ascontiguousarray(t.numpy(), np.float32)
    keep = [f32(frame[k]) for k in ("bg", "view", "proj", "campos")]
    fr = WmFrame(a.width, a.height, frame["tanfovx"], frame["tanfovy"], frame["kernel_size"], 1.0,
                 *[k.ctypes.data_as(C.c_void_p) for k in keep])
    m, s, r, o = f32(g["means3D"]), f32(g["scales"]), f32(g["rotations"]), f32(g["opacities"]).reshape(-1)
    TX, TY = (a.width + 7) // 8, (a.height + 7) // 8
    tx0, ty0 = (TX - a.window) // 2, (TY - a.window) // 2
    out = np.zeros(160, np.float64)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    nv = lib.wm_run(C.byref(fr), a.n, p(m), p(s), p(r), p(o), tx0, tx0 + a.window, ty0, ty0 + a.window, p(out), 160)
    names = ["n_tiles", "sumL", "sumK", "hits", "dense16", "sp16", "sp32", "sp64", "spInf", "q16", "q64", "strip16",
             "dead_entries", "live_entries", "behind", "task_nonzero", "task_total", "D_all", "all16_batches",
             "live16_batches", "f64_bbox", "f64_xy", "f64_exact", "f64_ideal", "f32_bbox", "f32_xy", "f32_exact", "f32_ideal",
             "fwd_steps", "fwd_steps_no_accept", "row_task_nonzero", "row_task_total", "p2_groups_zero", "p2_groups", "sp16_ahead",
             "walk_waves", "walk_steps_max", "walk_steps_balanced", "walk_tiles", "walk_rows_max", "walk_waves_over32", "walk_steps_hybrid"]
    v = dict(zip(names, out[:nv]))
    T = v["n_tiles"]
    print(f"tiles {T:.0f}  mean list {v['sumL']/T:.1f}  mean kmax {v['sumK']/T:.1f}  D_all {v['D_all']:.0f}")
    print(f"hit density (accepted pairs / 64 kmax): {v['hits']/(64*v['sumK']):.3f}")
    K = v["sumK"]
    for k in ("dense16", "sp16", "sp32", "sp64", "spInf", "q16", "q64", "strip16"):
        print(f"  phase-1 steps {k:8s}: {v[k]/K:.3f} of kmax")
    print(f"dead entries (< kmax, no accepted pixel): {v['dead_entries']/K:.3f}; entries behind kmax: {v['behind']/v['sumL']:.3f} of L")
    print(f"phase-2 (entry,row-pair) tasks non-zero: {v['task_nonzero']/v['task_total']:.3f}")
    for k in ("f64_bbox", "f64_xy", "f64_exact", "f64_ideal", "f32_bbox", "f32_xy", "f32_exact", "f32_ideal"):
        print(f"  forward strip-list steps {k:10s}: {v[k]/v['sumK']:.3f} of kmax")
    print(f"16-batches after dropping dead entries: {v['live16_batches']/v['all16_batches']:.3f}")
    # round 5: what the "skip work nobody needs" ideas could skip
    print(f"forward (TRAIN, 32-entry halves): strip-list steps in which NO lane of the wave blends its entry: "
          f"{v['fwd_steps_no_accept']/v['fwd_steps']:.4f} of {v['fwd_steps']/K:.3f} kmax steps  (wave-uniform early-out of the blend block)")
    print(f"phase 2: 4-pixel groups that are zero in every lane of a 16-entry batch: {v['p2_groups_zero']/v['p2_groups']:.4f}  "
          f"(skip by a scalar branch on the batch's hit bits)")
    print(f"phase 2 compaction: live (entry, 2 pixel rows) tasks {v['task_nonzero']/v['task_total']:.3f}, live (entry, 1 pixel row) tasks "
          f"{v['row_task_nonzero']/v['row_task_total']:.3f}  (a pass of 64 lanes is saved only below 0.5)")
    print(f"phase 1, batches of 16, an idle lane may take ONE hit of the next batch early: {v['sp16_ahead']/K:.3f} of kmax (now {v['sp16']/K:.3f})")
    print(f"preprocess tile walk: {v['walk_tiles'] / (64 * v['walk_waves']):.2f} tiles walked per Gaussian; wave steps now (max over the 64 lanes) "
          f"{v['walk_steps_max'] / v['walk_waves']:.2f}, balanced over the wave (lane = (Gaussian, tile) task) {v['walk_steps_balanced'] / v['walk_waves']:.2f}; "
          f"rows: max over lanes {v['walk_rows_max'] / v['walk_waves']:.2f}; waves holding a walk of more than 32 tiles "
          f"{v['walk_waves_over32'] / v['walk_waves']:.4f} (steps with the shipped per-wave choice {v['walk_steps_hybrid'] / v['walk_waves']:.2f})")
    h = out[nv:nv + 65]
    cum = np.cumsum(h) / h.sum()
    print("hits/entry quantiles:", {q: int(np.searchsorted(cum, q)) for q in (0.1, 0.25, 0.5, 0.75, 0.9, 0.99)})


if __name__ == "__main__":
    main()
