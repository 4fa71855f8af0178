"""Launch hints (include/sfgs.h SFGS_HINT_*; diff_gauss keeps per-stream state from the previous frames): a frame's optional
kernels -- huge-splat walk, long-list sorts, dead-entry prefill, chunk pre-reduction -- are only launched when the
previous frames say they have work. A WRONG prediction must never change a result: the huge-splat hint is verified
against the plan's own count and the frame redone, the others are correct by construction. Every case renders a
sequence of frames whose character changes abruptly, once with the mechanism on and once with SFGS_HINTS=0, and compares
bit for bit (images, radii, every gradient).

These are ROUTE-EQUALITY tests: both sides are the HIP library. What they establish is that the hinted route equals the
unhinted one; the unhinted route (every optional kernel launched, two-kernel sort) is the one pinned to the C oracle --
tests/test_gpu_raster.py::test_forward_backward_parity (first frame of a process: no hint learnt yet) and, for the hinted
routes at size, tests/test_gpu_fullsize_parity.py (frame 3, route asserted from the hint word)."""
import os

import numpy as np
import pytest
import torch

from sfgs.synth import city_scene, scene, upstream_grads

pytestmark = pytest.mark.gpu
W, H = 512, 288


def _frames():
    """(name, frame, gaussians): calm -> screen-filling splats -> calm -> long lists -> opaque city (dead entries)"""
    out = []
    f, g = scene(40000, W, H, seed=1, zrange=(250., 350.), scale_range=(0.2, 2.0))
    out.append(("calm", f, g))
    f2, g2 = scene(3000, W, H, seed=2, zrange=(4., 8.), scale_range=(0.5, 6.0))       # splats covering the whole screen
    out.append(("huge_splats", f2, g2))
    out.append(("calm_again", f, g))
    f3, g3 = scene(400000, W, H, seed=3, zrange=(250., 350.), scale_range=(1.0, 6.0))  # lists of > 512 .. > 2048 entries
    out.append(("long_lists", f3, g3))
    f4, g4 = city_scene(120000, W, H, 25.0, seed=4)
    out.append(("opaque_city", f4, g4))
    out.append(("calm_last", f, g))
    return out


def diff_gauss_last_backward_hints():
    import diff_gauss
    return diff_gauss.last_backward_hints()


def _run(frames, reps):
    from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer, last_counters
    dev = torch.device("cuda:0")
    res = []
    for name, frame, g in frames:
        for rep in range(reps):     # the second pass over the same content runs WITH the hints learnt from the first
            settings = GaussianRasterizationSettings(
                image_height=H, image_width=W, tanfovx=frame["tanfovx"], tanfovy=frame["tanfovy"],
                kernel_size=frame["kernel_size"], subpixel_offset=None, bg=frame["bg"].to(dev), scale_modifier=1.0,
                viewmatrix=frame["view"].to(dev), projmatrix=frame["proj"].to(dev), sh_degree=0,
                campos=frame["campos"].to(dev), prefiltered=False, debug=False)
            t = {k: v.to(dev).requires_grad_(True) for k, v in g.items() if v is not None}
            m2 = torch.zeros(t["means3D"].shape[0], 3, device=dev, requires_grad=True)
            color, depth, _, alpha, radii, _ = GaussianRasterizer(settings)(
                means3D=t["means3D"], means2D=m2, opacities=t["opacities"], colors_precomp=t["colors_precomp"],
                scales=t["scales"], rotations=t["rotations"])
            gc, gd = upstream_grads(W, H, 7)
            gd = gd.to(dev).clone()
            gd[torch.isnan(depth)] = 0
            torch.autograd.backward([color, torch.nan_to_num(depth)], [gc.to(dev), gd])
            rec = dict(name=f"{name}#{rep}", color=color.detach().cpu().numpy(), depth=depth.detach().cpu().numpy(),
                       radii=radii.cpu().numpy(), m2=m2.grad.cpu().numpy(),
                       **{"g_" + k: v.grad.cpu().numpy() for k, v in t.items()})
            rec["counters"] = last_counters()
            rec["bwd_hints"] = diff_gauss_last_backward_hints()
            res.append(rec)
    return res


def test_hints_never_change_a_result(monkeypatch):
    import diff_gauss
    monkeypatch.setattr(diff_gauss, "HUGE_QUIET_FRAMES", 1)   # no hysteresis: the sequence below must hit the violated-hint redo
    monkeypatch.setattr(diff_gauss, "ORDER_MIN_DUP", 0)       # the tile-order hint at these frame sizes too
    frames = _frames()
    old = os.environ.get("SFGS_HINTS")
    try:
        os.environ["SFGS_HINTS"] = "0"
        diff_gauss._hint_state.clear()
        ref = _run(frames, 2)
        os.environ["SFGS_HINTS"] = "1"
        diff_gauss._hint_state.clear()
        got = _run(frames, 2)
        state = dict(next(iter(diff_gauss._hint_state.values())))
    finally:
        if old is None:
            os.environ.pop("SFGS_HINTS", None)
        else:
            os.environ["SFGS_HINTS"] = old
    assert state["fb"] is not None and int(state["fb"][0]) == 1 and state["huge"] == 0
    for a, b in zip(ref, got):
        for k in a:
            if k in ("name", "counters", "bwd_hints"):
                continue
            np.testing.assert_array_equal(np.nan_to_num(a[k], nan=-1.0), np.nan_to_num(b[k], nan=-1.0),
                                          err_msg=f"{a['name']}: {k}")
    # the sequence did exercise what it claims to
    by = {r["name"]: r["counters"] for r in got}
    assert by["long_lists#1"]["num_duplicates"] > 10 * by["calm#1"]["num_duplicates"]
    hinted = {r["name"]: r["counters"]["fwd_hints"] for r in got}
    assert any(h & diff_gauss.HINT_TILE_ORDER for h in hinted.values()), hinted          # longest-first tile order
    assert any(h & diff_gauss.HINT_MEDIUM_LISTS for h in hinted.values()), hinted
    assert not any(r["counters"]["fwd_hints"] for r in ref)


def test_huge_splat_hint_waits_for_a_quiet_period(monkeypatch):
    """NO_HUGE_SPLATS is the one hint whose violation costs a second plan + render. After a frame with huge splats it is
    asserted again only after HUGE_QUIET_FRAMES frames without one, so a camera schedule that alternates between views
    with and without such splats (the IDU stage's mixed elevations, train.py:364-420) does not redo every other frame."""
    import diff_gauss
    from diff_gauss import HINT_NO_HUGE_SPLATS
    monkeypatch.setattr(diff_gauss, "HUGE_QUIET_FRAMES", 3)
    fr = {n: (f, g) for n, f, g in _frames()}
    calm, huge = ("calm",) + fr["calm"], ("huge_splats",) + fr["huge_splats"]
    diff_gauss._hint_state.clear()
    seq = [calm, calm, huge, calm, huge, calm, calm, calm, calm, calm]
    res = _run(seq, 1)
    hinted = [bool(r["counters"]["fwd_hints"] & HINT_NO_HUGE_SPLATS) for r in res]
    nhuge = [r["counters"]["num_huge_splats"] for r in res]
    assert nhuge[2] > 0 and nhuge[4] > 0 and nhuge[3] == 0
    # frame 0: nothing known; 1: calm after calm -> hinted; 2: the huge frame was planned WITH the hint, found out, redone
    # without it; 3 .. 7: quiet period (re-armed by frame 4); 8: three calm frames (5, 6, 7) later the hint is back
    assert hinted == [False, True, False, False, False, False, False, False, True, True], hinted


def test_dead_entry_kernel_stays_in_while_some_views_need_it(monkeypatch):
    """The backward's dupgrad_prefill_kernel decides per frame, on the device, whether the frame runs on live flags; the
    wrapper skips its launch (NO_PREFILL) only after PREFILL_QUIET "no"s in a row -- a view that needs the flags and is not
    given them costs 30 % (profiles/r6_live_flags_ab.txt), the kernel 3 us -- and then probes every PREFILL_PROBE_EVERY-th
    backward. A view full of dead entries (screen-filling splats: a pixel saturates after a dozen of its 3 000 entries) between
    calm ones keeps the kernel in for the calm views that follow."""
    import diff_gauss
    from diff_gauss import HINT_NO_PREFILL
    monkeypatch.setattr(diff_gauss, "PREFILL_QUIET", 3)
    monkeypatch.setattr(diff_gauss, "PREFILL_PROBE_EVERY", 1000)
    fr = {n: (f, g) for n, f, g in _frames()}
    calm, city = ("calm",) + fr["calm"], ("huge_splats",) + fr["huge_splats"]
    diff_gauss._hint_state.clear()
    res = _run([calm, calm, calm, city, calm, city, calm, calm, calm, calm, calm], 1)
    skipped = [bool(r["bwd_hints"] & HINT_NO_PREFILL) for r in res]
    # backward 0: nothing known, launched (state 1 -> 0 is seen by the NEXT forward's plan); 1, 2: hinted away. The probe
    # interval is out of reach here, so the heavy view of backward 3 runs without the kernel -- and nothing learns from it.
    assert skipped[:4] == [False, True, True, True], skipped
    # ... which is why the default probes every 32nd backward. Now with a probe at every backward:
    monkeypatch.setattr(diff_gauss, "PREFILL_PROBE_EVERY", 1)
    diff_gauss._hint_state.clear()
    res = _run([calm, calm, city, calm, city, calm, calm, calm, calm, calm], 1)
    state = [r["bwd_hints"] & HINT_NO_PREFILL for r in res]
    assert not any(state), state    # probing every backward = always launched
    hs = next(iter(diff_gauss._hint_state.values()))
    # after the last heavy view (backward 4) five calm ones said no; the last one's answer arrives with the next plan:
    # 3 -> 2 -> 1 -> 0 -> 0
    assert hs["prefilled"] == 0, hs["prefilled"]
    # the quiet period itself: a "yes" re-arms it
    monkeypatch.setattr(diff_gauss, "PREFILL_PROBE_EVERY", 1000)
    diff_gauss._hint_state.clear()
    res = _run([city, calm, calm, calm, calm, calm, calm], 1)
    skipped = [bool(r["bwd_hints"] & HINT_NO_PREFILL) for r in res]
    # 0: launched, says yes (state 3, known from plan 1 on); 1, 2, 3: launched, "no" x 3 -> 2, 1, 0 (known from plans 2, 3, 4);
    # backward 4 onwards: hinted away
    assert skipped == [False, False, False, False, True, True, True], skipped


def test_capacities_shrink_slowly_so_alternating_views_plan_once():
    """The blobs of a frame are sized from the previous frames' counts; an overflowing plan is redone (a second plan +
    render). After a heavy view the capacities shrink by 3 % per frame only, so a schedule that alternates between light
    and heavy views -- 10x the duplicates here -- plans every frame ONCE after it has seen the heavy one."""
    import diff_gauss
    fr = {n: (f, g) for n, f, g in _frames()}
    light, heavy = ("calm",) + fr["calm"], ("long_lists",) + fr["long_lists"]
    diff_gauss._hint_state.clear()
    diff_gauss._cap_hint.clear()
    res = _run([light, heavy, light, heavy, light, light, heavy], 1)
    att = [r["counters"]["plan_attempts"] for r in res]
    dup = [r["counters"]["num_duplicates"] for r in res]
    assert dup[1] > 8 * dup[0]
    assert att[1] >= 2                      # the first heavy view does not fit the light view's blobs
    assert att[2:] == [1, 1, 1, 1, 1], att  # ... nothing after it is planned twice
    caps = [r["counters"]["dup_capacity"] for r in res]
    # (frames 4, 5 are two light views in a row: the second runs with 97 % of the first one's capacity; every heavy view
    # brings it back to 1.25 x its own need)
    assert int(caps[4] * 0.97) - 1 <= caps[5] < caps[4] and caps[6] >= dup[6]


def test_medium_lists_hint_engages_and_changes_nothing():
    """A frame whose longest list has 513 .. 1 024 entries (the reference's low-elevation IDU cameras): from the second
    frame on the wrapper asks for the fused kernel's 1 024-entry form (SHORT_LISTS | MEDIUM_LISTS) instead of falling back
    to fine_bin + two sort kernels; images, radii and gradients equal the hint-less run bit for bit."""
    import diff_gauss
    from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer, collect_full_counters, last_counters
    dev = torch.device("cuda:0")
    w, h = 24, 24
    frame, g = scene(900, w, h, seed=5, zrange=(3., 6.), scale_range=(0.8, 1.5), opacity_range=(0.006, 0.03))
    gc, gd = (t.to(dev) for t in upstream_grads(w, h, 2))

    def run(n_frames):
        out = []
        for _ in range(n_frames):
            settings = GaussianRasterizationSettings(
                image_height=h, image_width=w, tanfovx=frame["tanfovx"], tanfovy=frame["tanfovy"],
                kernel_size=frame["kernel_size"], subpixel_offset=None, bg=frame["bg"].to(dev), scale_modifier=1.0,
                viewmatrix=frame["view"].to(dev), projmatrix=frame["proj"].to(dev), sh_degree=0,
                campos=frame["campos"].to(dev), prefiltered=False, debug=False)
            t = {k: v.to(dev).requires_grad_(True) for k, v in g.items() if v is not None}
            m2 = torch.zeros(900, 3, device=dev, requires_grad=True)
            color, depth, _, alpha, radii, _ = GaussianRasterizer(settings)(
                means3D=t["means3D"], means2D=m2, opacities=t["opacities"], colors_precomp=t["colors_precomp"],
                scales=t["scales"], rotations=t["rotations"])
            torch.autograd.backward([color, torch.nan_to_num(depth)], [gc, gd])
            out.append(dict(color=color.detach().cpu().numpy(), depth=depth.detach().cpu().numpy(),
                            radii=radii.cpu().numpy(), m2=m2.grad.cpu().numpy(), counters=last_counters(),
                            **{"g_" + k: v.grad.cpu().numpy() for k, v in t.items()}))
        return out

    old = os.environ.get("SFGS_HINTS")
    try:
        os.environ["SFGS_HINTS"] = "0"
        diff_gauss._hint_state.clear()
        collect_full_counters(True)
        ref = run(1)
        collect_full_counters(False)
        assert 512 < ref[0]["counters"]["max_tile_list"] <= 1024
        os.environ["SFGS_HINTS"] = "1"
        diff_gauss._hint_state.clear()
        got = run(3)
    finally:
        collect_full_counters(False)
        if old is None:
            os.environ.pop("SFGS_HINTS", None)
        else:
            os.environ["SFGS_HINTS"] = old
    both = diff_gauss.HINT_SHORT_LISTS | diff_gauss.HINT_MEDIUM_LISTS
    assert got[0]["counters"]["fwd_hints"] & both == 0                   # nothing learnt yet: the split route
    assert got[2]["counters"]["fwd_hints"] & both == both                 # lists of 513 .. 1 024: the 1 024-entry fused kernel
    for r in got:
        for k in ref[0]:
            if k != "counters":
                np.testing.assert_array_equal(np.nan_to_num(r[k], nan=-1.0), np.nan_to_num(ref[0][k], nan=-1.0), err_msg=k)
