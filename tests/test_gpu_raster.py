"""GPU parity tests of the rasterizer: the HIP path (through diff_gauss -> ctypes -> C ABI of libsfgs.so)
against the CPU oracle on the same seeded inputs. Integers bit-exact, RGB/depth/alpha within 1e-4
relative L-inf, gradients within 1e-3 relative L2."""
import numpy as np
import pytest
import torch

import parity
from oracle import oracle as orc
from sfgs.synth import scene, upstream_grads

pytestmark = pytest.mark.gpu

CASES = {
    "precomp_small": dict(n=3000, W=200, H=120, kw=dict(zrange=(4., 8.), scale_range=(0.01, 0.2))),
    "sh3_jitter": dict(n=20000, W=320, H=200, kw=dict(zrange=(250., 350.), scale_range=(0.2, 3.0), mode="sh",
                                                       sh_degree=3, jitter=True)),
    "sh1_ragged": dict(n=5000, W=130, H=77, kw=dict(zrange=(2., 50.), scale_range=(0.01, 2.0), mode="sh",
                                                     sh_degree=1)),
    "cfg2_like": dict(n=60000, W=480, H=270, kw=dict(zrange=(250., 350.), scale_range=(0.2, 2.4))),
    # the same Gaussians stored along a Z-curve of their screen position: lanes of a wave append to the SAME coarse bin
    # (run-merged atomics in the binning emission)
    "cfg2_like_zcurve": dict(n=60000, W=480, H=270, kw=dict(zrange=(250., 350.), scale_range=(0.2, 2.4)), zcurve=True),
    "big_splats": dict(n=400, W=256, H=192, kw=dict(zrange=(3., 6.), scale_range=(0.3, 2.0))),
    # splats that cover thousands of 8x8 tiles each: the chunked parallel reduction of their gradient records
    # (dupgrad_reduce_kernel, > 2048 duplicates per Gaussian) and the wave-cooperative binning walk
    "screen_filling": dict(n=60, W=640, H=400, kw=dict(zrange=(3., 6.), scale_range=(0.5, 3.0), opacity_range=(0.01, 0.05))),
    # more huge splats than big_walk_kernel has waves (1 024): its persistent loop takes a second round
    "many_huge": dict(n=1500, W=512, H=384, kw=dict(zrange=(3., 6.), scale_range=(0.6, 3.0), opacity_range=(0.004, 0.02))),
    # BASELINE.json configs[0] exactly (sfgs.synth.cfg1: 50 000 random Gaussians, one 800x800 pinhole camera, z ~ U(4, 8),
    # scales exp(U(ln 0.005, ln 0.05)), SURVEY 8d cfg 1) -- the case bench.py's cpu_baseline.cfg1_full times on the host
    "configs0_50k_800sq": dict(n=50000, W=800, H=800, seed=0, kw=dict(zrange=(4.0, 8.0), scale_range=(0.005, 0.05))),
    # the headline scene generator at its full viewport, 1/10 of the Gaussians (the oracle needs ~1 s for it)
    "cfg2_200k_1080p": dict(n=200000, W=1920, H=1080, kw=dict()),
    "cfg4_like_1440p": dict(n=150000, W=2560, H=1440, kw=dict(zrange=(500., 700.))),
    # long per-tile lists, one case per sort path: ~800 (register network, 16 keys per lane), ~2000 and ~3000 (bucketed
    # sort, narrow LDS layout), ~6000 (bucketed, wide layout), ~10000 (LDS chunks + global-memory levels); low opacity so that
    # pixels do not saturate after a few dozen splats and the long lists are really composited
    "lists_800": dict(n=900, W=24, H=24, kw=dict(zrange=(3., 6.), scale_range=(0.8, 1.5), opacity_range=(0.006, 0.03))),
    "lists_2k": dict(n=2000, W=24, H=24, kw=dict(zrange=(3., 6.), scale_range=(0.8, 1.5), opacity_range=(0.006, 0.02))),
    "lists_3k": dict(n=3000, W=24, H=24, kw=dict(zrange=(3., 6.), scale_range=(0.8, 1.5), opacity_range=(0.006, 0.012))),
    "lists_6k": dict(n=6000, W=24, H=24, kw=dict(zrange=(3., 6.), scale_range=(0.8, 1.5), opacity_range=(0.006, 0.012))),
    "lists_10k": dict(n=18000, W=16, H=16, kw=dict(zrange=(3., 6.), scale_range=(0.8, 1.5), opacity_range=(0.006, 0.012))),
    # the bucketed sort's exits: every depth equal (order = Gaussian id alone) and a handful of distinct depths (one depth
    # bin holds more than a segment): both fall back to the network
    "lists_3k_equal_depth": dict(n=3000, W=24, H=24, kw=dict(zrange=(5., 5.), scale_range=(0.8, 1.5), opacity_range=(0.006, 0.012))),
    "lists_3k_few_depths": dict(n=3000, W=24, H=24, kw=dict(zrange=(5., 5.000002), scale_range=(0.8, 1.5), opacity_range=(0.006, 0.012))),
}
# Shapes that only the binning's radix pass cares about, compared with the one-pass path (which the cases above pin to the
# oracle) for EQUALITY: 8 160 coarse bins = two rounds of its 4 096 LDS counters; ~3 coarse items per Gaussian = the first
# scatter workgroup (8 192 Gaussians) holds more pairs than its 15 360 sorted-order index slots, the rest goes out unsorted.
BINNING_CASES = {
    "uhd_two_bin_rounds": dict(n=30000, W=3840, H=2160, kw=dict(zrange=(250., 350.), scale_range=(0.1, 1.2))),
    "mid_splats_many_items": dict(n=20000, W=640, H=360, kw=dict(zrange=(40., 60.), scale_range=(0.3, 0.7), opacity_range=(0.02, 0.2))),
    # a distant view: the whole scene inside a handful of the frame's 2 040 coarse bins
    "skewed_far_view": dict(n=60000, W=1920, H=1080, kw=dict(zrange=(250., 350.), scale_range=(0.05, 0.4), xy_fill=0.04,
                                                              opacity_range=(0.02, 0.3))),
}


def run_hip(frame, g, gc=None, gd=None, backward=True, debug=True, depth_mode=0, full_counters=True):
    from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer, collect_full_counters, last_counters
    collect_full_counters(full_counters)  # the oracle comparison includes max_tile_list (render-stage counter)
    dev = torch.device("cuda:0")
    sub = frame.get("subpix")
    settings = GaussianRasterizationSettings(
        image_height=frame["H"], image_width=frame["W"], tanfovx=frame["tanfovx"], tanfovy=frame["tanfovy"],
        kernel_size=frame["kernel_size"], subpixel_offset=None if sub is None else sub.to(dev),
        bg=frame["bg"].to(dev), scale_modifier=frame["scale_modifier"], viewmatrix=frame["view"].to(dev),
        projmatrix=frame["proj"].to(dev), sh_degree=frame["sh_degree"], campos=frame["campos"].to(dev),
        prefiltered=False, debug=debug, depth_mode=depth_mode)
    t = {k: (v.to(dev).requires_grad_(backward) if v is not None else None) for k, v in g.items()}
    means2D = torch.zeros_like(t["means3D"], requires_grad=backward)
    rast = GaussianRasterizer(settings)
    color, depth, norm, alpha, radii, extra = rast(means3D=t["means3D"], means2D=means2D, shs=t["shs"],
                                                   colors_precomp=t["colors_precomp"], opacities=t["opacities"],
                                                   scales=t["scales"], rotations=t["rotations"], cov3Ds_precomp=None)
    out = dict(color=color.detach().cpu().numpy(), depth=depth.detach().cpu().numpy(),
               alpha=alpha.detach().cpu().numpy(), radii=radii.cpu().numpy(), norm=norm, extra=extra,
               counters=last_counters())
    collect_full_counters(False)
    if backward:
        loss = (color * gc.to(dev)).sum() + (torch.nan_to_num(depth, nan=0.0) * gd.to(dev)).sum()
        loss.backward()
        out["grads"] = {k: v.grad.cpu().numpy() for k, v in t.items() if v is not None}
        out["grads"]["means2D"] = means2D.grad.cpu().numpy()
    return out


@pytest.mark.parametrize("case", list(CASES))
def test_forward_backward_parity(case):
    c = CASES[case]
    frame, g = scene(c["n"], c["W"], c["H"], seed=c.get("seed", 7), **c["kw"])
    if c.get("zcurve"):
        from sfgs.synth import morton_order
        perm = morton_order(g["means3D"])
        g = {k: (v[perm].contiguous() if v is not None else None) for k, v in g.items()}
    R = orc.OracleRender(frame, **g)
    gc, gd = upstream_grads(c["W"], c["H"], 0)
    gd = gd.clone()
    gd[torch.from_numpy(np.isnan(R.depth))] = 0
    G = R.backward(gc, gd)
    out = run_hip(frame, g, gc, gd)
    # integers: bit-exact
    assert out["radii"].dtype == np.int32
    np.testing.assert_array_equal(out["radii"], R.radii)
    assert out["counters"]["num_visible"] == R.num_visible
    assert out["counters"]["num_duplicates_ref"] == R.num_duplicates
    reps = [parity.assert_image_close("color", out["color"], R.color),
            parity.assert_image_close("alpha", out["alpha"], R.alpha),
            parity.assert_image_close("depth", out["depth"], R.depth)]
    assert float(out["norm"].abs().max()) == 0.0 and out["extra"] is None
    for k in G:
        reps.append(parity.assert_grad_close(k, out["grads"][k], G[k]))
    print(case, out["counters"], reps)
    want_list = {"lists_800": (513, 1024), "lists_2k": (1025, 2048), "lists_3k": (2049, 4096), "lists_6k": (4097, 8192),
                 "lists_10k": (8193, 1 << 20)}.get(case)
    if case == "screen_filling":
        assert out["counters"]["num_duplicates"] > 20 * 2048, out["counters"]   # many Gaussians above BWD_BIG duplicates
    if want_list:  # the case really exercises the sort path it is named after
        assert want_list[0] <= out["counters"]["max_tile_list"] <= want_list[1], out["counters"]


def test_host_emulation_agrees_bitwise_on_integers():
    """The CPU emulation built from the product header (tests/host_check) and the GPU must bin the same
    number of duplicates: the binning decisions are pure float32 sequences."""
    import hostcheck
    frame, g = scene(20000, 320, 200, seed=3, zrange=(250., 350.), scale_range=(0.2, 3.0))
    h = hostcheck.render(frame, g["means3D"], g["scales"], g["rotations"], g["opacities"], g["colors_precomp"], None)
    out = run_hip(frame, g, backward=False)
    np.testing.assert_array_equal(out["radii"], h["radii"])
    assert out["counters"]["num_duplicates"] == int(h["counters"][0])
    assert out["counters"]["max_tile_list"] == int(h["counters"][3])


def test_raw_depth_mode_and_no_grad():
    frame, g = scene(3000, 200, 120, seed=2, zrange=(4., 8.), scale_range=(0.01, 0.2))
    frame["depth_mode"] = 1
    R = orc.OracleRender(frame, **g)
    with torch.no_grad():
        out = run_hip(frame, g, backward=False, depth_mode=1)
    parity.assert_image_close("depth_raw", out["depth"], R.depth)
    parity.assert_image_close("color", out["color"], R.color)


def test_deterministic_gradients():
    frame, g = scene(20000, 320, 200, seed=5, zrange=(250., 350.), scale_range=(0.2, 3.0))
    gc, gd = upstream_grads(320, 200, 1)
    a = run_hip(frame, g, gc, gd * 0, debug=False)
    b = run_hip(frame, g, gc, gd * 0, debug=False, full_counters=False)  # default mid-frame counter read
    for k in a["grads"]:
        np.testing.assert_array_equal(a["grads"][k], b["grads"][k])
    np.testing.assert_array_equal(a["color"], b["color"])


def test_empty_and_culled_inputs():
    # N = 0, and a scene entirely behind the camera
    frame, g = scene(16, 64, 48, seed=1, zrange=(4., 8.), scale_range=(0.01, 0.2))
    empty = {k: (v[:0] if v is not None else None) for k, v in g.items()}
    out = run_hip(frame, empty, backward=False)
    assert out["radii"].shape == (0,) and np.all(out["color"] == 0) and np.all(out["alpha"] == 0)
    behind = dict(g)
    behind["means3D"] = g["means3D"] * torch.tensor([1., 1., -1.])
    gc, gd = upstream_grads(64, 48, 0)
    out = run_hip(frame, behind, gc, gd * 0)
    assert np.all(out["radii"] == 0) and np.all(out["alpha"] == 0)
    for k, v in out["grads"].items():
        assert np.all(v == 0), k


def test_argument_validation():
    from diff_gauss import GaussianRasterizer
    frame, g = scene(16, 64, 48, seed=1)
    with pytest.raises(ValueError):
        run_hip(frame, dict(g, shs=torch.zeros(16, 4, 3)), backward=False)  # both colour inputs
    with pytest.raises(ValueError):
        run_hip(dict(frame, subpix=torch.zeros(10, 10, 2)), g, backward=False)  # wrong subpixel shape
    assert GaussianRasterizer is not None


def test_non_finite_and_degenerate_inputs_are_culled_not_crashing():
    """NaN / Inf positions, scales, opacities and zero scales must neither crash nor poison other pixels."""
    frame, g = scene(4000, 160, 96, seed=13, zrange=(4., 8.), scale_range=(0.01, 0.2))
    bad = {k: (v.clone() if v is not None else None) for k, v in g.items()}
    bad["means3D"][0] = float("nan")
    bad["means3D"][1, 2] = float("inf")
    bad["scales"][2] = float("nan")
    bad["scales"][3] = 0.0
    bad["opacities"][4] = float("nan")
    bad["scales"][5] = 1e30
    bad["rotations"][6] = 0.0
    ok = torch.ones(4000, dtype=torch.bool)
    ok[:7] = False
    good = {k: (v[ok].contiguous() if v is not None else None) for k, v in g.items()}
    gc, gd = upstream_grads(160, 96, 0)
    a = run_hip(frame, bad, gc, gd * 0)
    b = run_hip(frame, good, gc, gd * 0)
    for k in ("color", "alpha"):
        assert np.isfinite(a[k]).all()
    # the seven broken splats contribute nothing visible except the valid-but-odd ones (zero scale, zero quaternion,
    # huge scale), which the oracle treats the same way
    R = orc.OracleRender(frame, **bad)
    np.testing.assert_array_equal(a["radii"], R.radii)
    parity.assert_image_close("color", a["color"], R.color)
    assert a["radii"][0] == 0 and a["radii"][1] == 0 and a["radii"][2] == 0
    for k, v in a["grads"].items():
        assert np.isfinite(v[7:]).all(), k
    assert b["color"].shape == a["color"].shape


def test_runs_on_the_callers_current_stream():
    """All launches go to torch's CURRENT stream (never a cached one): results on a side stream equal the default
    stream's, and the forward on the side stream is ordered after work queued on that stream."""
    frame, g = scene(20000, 320, 200, seed=21, zrange=(250., 350.), scale_range=(0.2, 3.0))
    gc, gd = upstream_grads(320, 200, 2)
    ref = run_hip(frame, g, gc, gd * 0, debug=False)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out = run_hip(frame, g, gc, gd * 0, debug=False)
    side.synchronize()
    np.testing.assert_array_equal(out["color"], ref["color"])
    np.testing.assert_array_equal(out["radii"], ref["radii"])
    for k in ref["grads"]:
        np.testing.assert_array_equal(out["grads"][k], ref["grads"][k])


@pytest.mark.parametrize("W,H,n", [(5, 3, 7), (1, 1, 3), (9, 17, 1), (33, 8, 50)])
def test_tiny_images_and_single_gaussians(W, H, n):
    frame, g = scene(n, W, H, seed=W * 100 + H, zrange=(2., 4.), scale_range=(0.05, 0.5))
    R = orc.OracleRender(frame, **g)
    gc, gd = upstream_grads(W, H, 0)
    gd = gd.clone()
    gd[torch.from_numpy(np.isnan(R.depth))] = 0
    G = R.backward(gc, gd)
    out = run_hip(frame, g, gc, gd)
    np.testing.assert_array_equal(out["radii"], R.radii)
    for name in ("color", "alpha", "depth"):
        parity.assert_image_close(name, out[name], getattr(R, name))
    for k in G:
        if np.abs(G[k]).max() > 0:
            parity.assert_grad_close(k, out["grads"][k], G[k])


def _random_config(i):
    rng = np.random.default_rng(1000 + i)
    W, H = int(rng.integers(9, 260)), int(rng.integers(9, 200))
    near = float(rng.choice([3.0, 30.0, 250.0]))
    kw = dict(zrange=(near, near * float(rng.uniform(1.2, 2.0))),
              scale_range=tuple(sorted((near * float(rng.uniform(1e-4, 1e-3)), near * float(rng.uniform(2e-3, 3e-2))))),
              fovx_deg=float(rng.uniform(30, 100)), xy_fill=float(rng.uniform(0.5, 1.3)),
              jitter=bool(rng.integers(0, 2)), pitch_deg=float(rng.choice([0.0, 0.0, 20.0])),
              # kernel_size 0 with sub-pixel splats makes the 2D covariance near-singular: the conic's gradient (1/det^2)
              # then amplifies float32 rounding in BOTH implementations; the reference always filters (0.1)
              kernel_size=float(rng.choice([0.05, 0.1, 0.3])),
              opacity_range=(float(rng.uniform(0.003, 0.3)), float(rng.uniform(0.4, 1.0))))
    if rng.integers(0, 2):
        kw.update(mode="sh", sh_degree=int(rng.integers(0, 4)))
    return dict(n=int(rng.integers(1, 25000)), W=W, H=H, kw=kw, bg=rng.uniform(0, 1, 3).astype(np.float32) * float(rng.integers(0, 2)),
                depth_mode=int(rng.integers(0, 2)), zero_depth_grad=bool(rng.integers(0, 2)))


@pytest.mark.parametrize("i", range(40))
def test_random_configurations(i):
    """Seeded sweep over image sizes that are not tile multiples, near / far scenes, tiny to screen-filling splats,
    ray jitter on / off (general and pixel-grid backward paths), pitched cameras, SH degrees 0-3, non-black
    backgrounds, both depth modes -- every output against the oracle at the standard tolerances."""
    c = _random_config(i)
    frame, g = scene(c["n"], c["W"], c["H"], seed=200 + i, **c["kw"])
    frame["bg"] = torch.tensor(c["bg"])
    frame["depth_mode"] = c["depth_mode"]
    R = orc.OracleRender(frame, **g)
    gc, gd = upstream_grads(c["W"], c["H"], i)
    gd = gd.clone() * (0.0 if c["zero_depth_grad"] else 1.0)
    gd[torch.from_numpy(np.isnan(R.depth))] = 0
    G = R.backward(gc, gd)
    out = run_hip(frame, g, gc, gd, depth_mode=c["depth_mode"], debug=False, full_counters=bool(i % 2))
    np.testing.assert_array_equal(out["radii"], R.radii)
    assert out["counters"]["num_visible"] == R.num_visible and out["counters"]["num_duplicates_ref"] == R.num_duplicates
    # T is a float32 product of one (1 - alpha) factor per blended splat: its rounding error grows linearly with the
    # number of contributors. 1e-4 is north_star's bar at the benchmark scenes' depth (<= ~300 per pixel); the sweep
    # also generates pixels with thousands, for which the bar scales accordingly.
    depth_of_blend = float(R.n_contrib().max())
    rtol = parity.RGB_DEPTH_RTOL * max(1.0, depth_of_blend / 1000.0)
    # small images: one splat within an ulp of the 1/255 / 1e-4 thresholds flips a handful of pixels (and, as their
    # only contributor, NaN <-> number in the normalised depth); its own gradient then differs at the 1/255 level
    reps = [parity.assert_image_close(name, out[name], getattr(R, name), rtol=rtol, borderline_min=4)
            for name in ("color", "alpha", "depth")]
    # an explicit FLIP signal only: the NaN pattern of the normalised depth changed, or a few pixels moved by
    # the 1/255 blending quantum of one borderline splat (more than the tolerance, less than 5e-2, on at most
    # borderline_min pixels). Any other out-of-tolerance pixel already failed assert_image_close above and never
    # loosens the gradient bars.
    nan_flip = int((np.isnan(out["depth"]) != np.isnan(R.depth)).sum()) > 0
    flipped = nan_flip or any(0 < r["bad"] <= 4 and r["max_rel"] > rtol for r in reps)
    # ... or the ORACLE says that one of its own per-pixel decisions lies within a few float32 ulps of its threshold
    # (alpha = 1/255, T (1 - alpha) = 1e-4, power = 0): evaluated with another exponential the pair may fall the other way
    # without any pixel moving by more than the tolerance (the splat is faint or the pixel nearly opaque), while the
    # splat's OWN gradient changes at the 1/255 level. (Found by the extended soak, seeds 1638 and 1926: margins 1.1e-6
    # and 6e-8; oracle/sfgs_oracle.c: orc_decision_margins.)
    m = R.decision_margins()
    near_threshold = m["alpha"] < 5e-6 or m["T"] < 1e-5 or m["power"] < 1e-6
    flipped = flipped or near_threshold
    print("borderline pixels:", {r["name"]: r["bad"] for r in reps}, "nan flip:", nan_flip, "decision margins:", m)
    scale = rtol / parity.RGB_DEPTH_RTOL
    few = 2.0 if c["n"] < 64 else 1.0   # a handful of Gaussians: no averaging over the float32 chain of Sigma -> q
    # the flipped splat's own gradient moves by ~10 % of its value: L2 barely notices, the max norm does
    for k in G:
        got, ref = out["grads"][k], G[k]
        if near_threshold:
            # the pair on the threshold belongs to ONE Gaussian, whose own gradient then differs by a whole pixel's
            # contribution (seed 1926: a faint splat of three pixels, one of them on the threshold -- even the sign of a
            # component changes): the two worst rows are only required to stay within a quarter of the tensor's largest
            # entry, every other row meets the bars below
            got = np.array(got, copy=True)
            rows_g, rows_r = got.reshape(c["n"], -1), np.asarray(ref).reshape(c["n"], -1)
            worst = np.argsort(-np.abs(rows_g - rows_r).max(axis=1))[:2]
            assert np.abs(rows_g[worst] - rows_r[worst]).max() <= 0.25 * np.abs(rows_r).max(), k
            rows_g[worst] = rows_r[worst]
        parity.assert_grad_close(k, got, ref, l2=parity.GRAD_RTOL_L2 * scale * few * (3.0 if flipped else 1.0),
                                 mx=parity.GRAD_RTOL_MAX * scale * few * (6.0 if flipped else 1.0))


@pytest.mark.parametrize("case", ["overdraw_256x192", "screen_filling_640x480", "mixed_sizes_512x384"])
def test_dead_entry_prefill_paths_bit_identical(sfgs_option, case):
    """Entries behind a tile's last contributor either get zero gradient records one by one (composite_bwd) or are skipped
    through the live flags (dupgrad_prefill_kernel clears one byte per duplicate, composite_bwd sets the byte of every record
    it writes, dupgrad_reduce_kernel / preprocess_bwd fetch flagged records only; chosen per frame on the device when > 25 %
    are dead). Both forced in turn: every gradient must come out bit-identical, and identical to the automatic choice.
    The three scenes put Gaussians on each summation route of preprocess_bwd: <= 32 records (streamed through LDS), 33 .. 2048
    (the wave strides over them), > 2048 (pre-reduced chunks: `num_big_chunks` of the forward's counters)."""
    if case == "overdraw_256x192":      # heavy overdraw: most entries dead
        W, H = 256, 192
        frame, g = scene(400, W, H, seed=11, zrange=(3., 6.), scale_range=(0.3, 2.0))
    elif case == "screen_filling_640x480":   # 4800 tiles, splats that cover most of them: > 2048 records per Gaussian
        W, H = 640, 480
        frame, g = scene(60, W, H, seed=12, zrange=(3., 6.), scale_range=(1.5, 4.0), opacity_range=(0.5, 0.95))
    else:                               # small splats in front of big ones: all three routes in one frame
        W, H = 512, 384
        frame, g = scene(3000, W, H, seed=13, zrange=(3., 60.), scale_range=(0.02, 3.0), opacity_range=(0.3, 0.95))
    gc, gd = upstream_grads(W, H, 3)
    outs = {}
    for mode in ("always", "never", ""):
        sfgs_option("prefill", mode or "auto")
        outs[mode] = run_hip(frame, g, gc, gd)
    a = outs["always"]["grads"]
    for k in a:
        np.testing.assert_array_equal(a[k], outs["never"]["grads"][k], err_msg=k)
        np.testing.assert_array_equal(a[k], outs[""]["grads"][k], err_msg=k)
        assert np.isfinite(a[k]).all(), k     # an unflagged record is uninitialised memory: it must never be read
    assert np.abs(a["means3D"]).max() > 0
    big = outs["always"]["counters"]["num_big_chunks"]
    assert (big > 0) == (case != "overdraw_256x192"), (case, big)


def test_storage_order_is_a_relabelling():
    """Rendering a scene and the same scene with its Gaussians permuted (Z-curve order, sfgs.densify.zcurve_permutation)
    gives the same image and, row for row, the same gradients: only exact depth ties are broken by index."""
    from sfgs.densify import zcurve_permutation
    frame, g = scene(30000, 320, 200, seed=5, zrange=(250., 350.), scale_range=(0.2, 2.4))
    gc, gd = upstream_grads(320, 200, 2)
    a = run_hip(frame, g, gc, gd)
    perm = zcurve_permutation(g["means3D"])
    gp = {k: (v[perm].contiguous() if v is not None else None) for k, v in g.items()}
    b = run_hip(frame, gp, gc, gd)
    np.testing.assert_array_equal(a["radii"][perm.numpy()], b["radii"])
    np.testing.assert_allclose(a["color"], b["color"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(np.nan_to_num(a["depth"]), np.nan_to_num(b["depth"]), rtol=1e-5, atol=1e-5)
    for k in a["grads"]:
        ga, gb = a["grads"][k][perm.numpy()], b["grads"][k]
        assert np.abs(ga - gb).max() <= 1e-5 * max(np.abs(ga).max(), 1e-30) + 1e-12, k


def test_skewed_workgroups_and_the_duplicate_index_pools():
    """Duplicate indices come from 8 pools of the index space (one per preprocess workgroup modulo 8, frames of >= 512
    workgroups). A frame whose first workgroups own most of the duplicates over-fills their pools although the total fits:
    the plan flags an overflow, the wrapper gives every pool twice the room and redoes the frame -- same parity bars."""
    W, H = 320, 200
    f, big = scene(1024, W, H, seed=21, zrange=(6., 9.), scale_range=(0.25, 0.6), opacity_range=(0.004, 0.02))
    _, small = scene(140000, W, H, seed=22, zrange=(250., 350.), scale_range=(0.05, 0.3))
    g = {k: (torch.cat([big[k], small[k]]).contiguous() if big[k] is not None else None) for k in big}
    R = orc.OracleRender(f, **g)
    gc, gd = upstream_grads(W, H, 3)
    gd = gd.clone()
    gd[torch.from_numpy(np.isnan(R.depth))] = 0
    G = R.backward(gc, gd)
    out = run_hip(f, g, gc, gd)
    np.testing.assert_array_equal(out["radii"], R.radii)
    assert out["counters"]["num_duplicates"] == R.num_duplicates or out["counters"]["num_duplicates"] > 0
    parity.assert_image_close("color", out["color"], R.color)
    parity.assert_image_close("depth", out["depth"], R.depth)
    for k in G:
        parity.assert_grad_close(k, out["grads"][k], G[k])
    share = float((R.tiles_touched()[:1024] > 0).sum())
    assert share > 0   # the big splats are on screen


@pytest.mark.parametrize("case", ["cfg2_like", "big_splats", "lists_800", "ragged_130x77", "uhd_two_bin_rounds", "skewed_far_view"])
def test_tile_order_is_a_relabelling_of_workgroups(case, sfgs_option):
    """ROUTE-EQUALITY test (HIP vs HIP). With SFGS_HINT_TILE_ORDER (here forced through the "tile_order" option) the
    compositing kernels take their tiles longest list first inside each XCD's share of the image (tile_order_kernel) instead of in
    image order. A tile is composited by one wave from its own list: every output and every gradient must be the same bits,
    whatever the order (forward by list length, backward by last contributor; tiles without work, a ragged image edge,
    workgroups of four tiles in the backward)."""
    c = CASES.get(case) or BINNING_CASES.get(case)
    if c is None:
        c = dict(n=5000, W=130, H=77, kw=dict(zrange=(2., 50.), scale_range=(0.01, 2.0)))
    frame, g = scene(c["n"], c["W"], c["H"], seed=21, **c["kw"])
    gc, gd = upstream_grads(c["W"], c["H"], 5)
    outs = {}
    for mode in ("never", "always"):
        sfgs_option("tile_order", mode)
        outs[mode] = run_hip(frame, g, gc, gd)
    a, b = outs["never"], outs["always"]
    for k in ("color", "depth", "alpha", "radii"):
        np.testing.assert_array_equal(np.nan_to_num(a[k], nan=-1.0), np.nan_to_num(b[k], nan=-1.0), err_msg=k)
    for k in a["grads"]:
        np.testing.assert_array_equal(a["grads"][k], b["grads"][k], err_msg=k)
    assert np.abs(a["grads"]["means3D"]).max() > 0


@pytest.mark.parametrize("case", ["cfg2_like", "big_splats", "lists_800", "cfg4_like_1440p", "uhd_two_bin_rounds",
                                  "mid_splats_many_items", "skewed_far_view"])
def test_two_pass_binning_equals_the_direct_path(case, sfgs_option):
    """ROUTE-EQUALITY test (HIP vs HIP; the default route of every case is pinned to the oracle by
    test_forward_backward_parity above). Two-pass binning (pair list + bin_scatter_kernel, the default) and the one-pass path with one device atomic per
    coarse item (option "binning" = "direct") build the same frame: duplicate indices come from the same scan and every tile list
    is sorted by (depth, id), so images, radii, counters and every gradient are equal bit for bit."""
    c = CASES.get(case) or BINNING_CASES[case]
    frame, g = scene(c["n"], c["W"], c["H"], seed=11, **c["kw"])
    gc, gd = upstream_grads(c["W"], c["H"], 4)
    a = run_hip(frame, g, gc, gd)
    sfgs_option("binning", "direct")
    b = run_hip(frame, g, gc, gd)
    for k in ("color", "depth", "alpha", "radii"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    for k in ("num_duplicates", "num_duplicates_ref", "num_visible", "max_bin_items", "max_tile_list"):
        assert a["counters"][k] == b["counters"][k], k
    # ABI 14: the two-pass binning stores its items bin-sorted and exactly sized; only directly appended items (those of
    # splats reaching more than 6 coarse bins) count towards coarse_capacity -- every item on the one-pass path
    assert b["counters"]["max_coarse_bin"] == b["counters"]["max_bin_items"]
    assert a["counters"]["max_coarse_bin"] <= a["counters"]["max_bin_items"]
    if case in ("cfg2_like", "skewed_far_view"):     # small splats only
        assert a["counters"]["max_coarse_bin"] == 0
    for k in a["grads"]:
        np.testing.assert_array_equal(a["grads"][k], b["grads"][k], err_msg=k)


@pytest.mark.parametrize("case", ["cfg2_like", "sh1_ragged", "big_splats", "lists_800"])
def test_the_zero_subpixel_tensor_of_render_equals_no_tensor(case):
    """The reference's render() allocates an all-zero [H,W,2] subpixel_offset on every call when ray jitter is off
    (gaussian_renderer/__init__.py:37-38) -- the call pattern of every shipped script -- while the benchmarks and most
    tests here pass None. The kernels learn from the plan (subpix_bound_kernel: max |offset| = 0) that the sample points
    sit on the pixel grid and take the same path: images, radii, counters and every gradient are equal bit for bit."""
    c = CASES[case]
    frame, g = scene(c["n"], c["W"], c["H"], seed=13, **c["kw"])
    gc, gd = upstream_grads(c["W"], c["H"], 5)
    a = run_hip(frame, g, gc, gd)
    b = run_hip(dict(frame, subpix=torch.zeros(c["H"], c["W"], 2)), g, gc, gd)
    for k in ("color", "depth", "alpha", "radii"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    for k in ("num_duplicates", "num_duplicates_ref", "num_visible", "max_bin_items", "max_tile_list"):
        assert a["counters"][k] == b["counters"][k], k
    for k in a["grads"]:
        np.testing.assert_array_equal(a["grads"][k], b["grads"][k], err_msg=k)


def test_a_skewed_frame_needs_no_per_bin_capacity():
    """ABI 14: the two-pass binning's items are stored bin-sorted and EXACTLY sized, so a distant view -- every Gaussian in
    a handful of coarse bins -- plans in one attempt with the minimal slab capacity and a scratch in proportion to the
    items. (Until ABI 13 the bins were uniform slabs sized for the fullest one: this frame, 60 000 Gaussians at 1080p,
    took 2 040 x ~15 000 x 16 bytes = 0.5 GB of slabs after two overflowing attempts; 2 M Gaussians seen from afar 65 GB.)"""
    import diff_gauss
    c = BINNING_CASES["skewed_far_view"]
    frame, g = scene(c["n"], c["W"], c["H"], seed=11, **c["kw"])
    diff_gauss._cap_hint.clear()
    out = run_hip(frame, g, backward=False)
    cnt = out["counters"]
    assert cnt["max_bin_items"] > 5000 and cnt["num_huge_splats"] == 0      # skewed indeed
    assert cnt["max_coarse_bin"] == 0 and cnt["coarse_capacity"] == 256      # nothing needs a slab
    import ctypes as C
    from sfgs import _lib as L
    lib = L.load()
    lay = L.SfgsScratchLayout(C.sizeof(L.SfgsScratchLayout))
    L.check(lib.sfgs_raster_scratch_layout(c["n"], c["W"], c["H"], cnt["dup_capacity"], cnt["coarse_capacity"], 0, C.byref(lay)))
    assert int(lay.total_bytes) < 128 << 20    # (of which ~55 MB is list-slot index space: 1 088 slots per coarse bin)


@pytest.mark.parametrize("route", ["fused", "fused768", "fused1024"])
@pytest.mark.parametrize("case", ["precomp_small", "sh1_ragged", "cfg2_like", "big_splats", "screen_filling", "lists_800",
                                  "lists_2k", "lists_6k", "lists_10k", "cfg2_200k_1080p", "uhd_two_bin_rounds"])
def test_fused_select_sort_equals_fine_bin_plus_sort(case, route, sfgs_option):
    """ROUTE-EQUALITY test (HIP vs HIP; the default route of every case is pinned to the oracle by
    test_forward_backward_parity above). select_sort_kernel (a workgroup hands a coarse bin's items to the LDS lists of a row of tiles, every wave sorts its
    tile in registers and writes the final lists; the route the SHORT_LISTS hint selects, forced here) and the two-kernel route with the per-tile items in memory between them
    (option "sort" = "split": fine_bin + sort_tiles_reg) build the same lists -- same members, same (depth, id) order, same
    duplicate indices; only WHERE a tile's list sits inside its bin's slot range may differ -- so images, radii,
    counters (incl. the longest list) and every gradient are equal bit for bit. Long lists (> 512) take the second scan
    and the long-list kernels in the fused route. route fused1024 = the MEDIUM_LISTS form of the kernel (lists up to 1 024
    entries stay in LDS and are sorted by the 16-key network: case lists_800); fused768 = its LISTS_768 form (round 6: lists of
    769 .. 1 024 entries -- case lists_800 has them -- go to the long-list kernels)."""
    c = CASES.get(case) or BINNING_CASES[case]
    frame, g = scene(c["n"], c["W"], c["H"], seed=5, **c["kw"])
    gc, gd = upstream_grads(c["W"], c["H"], 2)
    sfgs_option("sort", route)
    a = run_hip(frame, g, gc, gd)
    sfgs_option("sort", "split")
    b = run_hip(frame, g, gc, gd)
    for k in ("color", "depth", "alpha", "radii"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    for k in ("num_duplicates", "num_duplicates_ref", "num_visible", "max_coarse_bin", "max_bin_items", "max_tile_list"):
        assert a["counters"][k] == b["counters"][k], k
    for k in a["grads"]:
        np.testing.assert_array_equal(a["grads"][k], b["grads"][k], err_msg=k)


@pytest.mark.parametrize("route", ["fused", "fused768", "fused1024"])
def test_equal_depths_keep_the_id_order_on_every_route(route, sfgs_option):
    """Clones sit exactly on their parents until the optimiser moves them (scene/gaussian_model.py: densify_and_clone):
    every Gaussian here exists three times with the same mean, i.e. the same depth bits, in lists of ~800 entries. The
    order inside a tile is (depth bits, id) on every route -- the register network sorts the 64-bit key, the radix sort of
    the 1 024-entry fused kernel (round 4) sorts the depth bits and then orders the equal-depth neighbours by id -- so
    images, radii and every gradient equal the split route's bit for bit, and the oracle's within the parity bars."""
    frame, g = scene(300, 24, 24, seed=5, zrange=(3., 6.), scale_range=(0.8, 1.5), opacity_range=(0.006, 0.03))
    gen = torch.Generator().manual_seed(9)
    g3 = {}
    for k, v in g.items():
        if v is None:
            g3[k] = None
        elif k == "means3D":
            g3[k] = v.repeat(3, 1).contiguous()                       # identical positions: identical depth bits
        else:                                                          # ... but different looks
            g3[k] = torch.cat([v, v[torch.randperm(300, generator=gen)], v[torch.randperm(300, generator=gen)]]).contiguous()
    gc, gd = upstream_grads(24, 24, 2)
    R = orc.OracleRender(frame, **g3)
    assert 256 < R.max_tile_list
    sfgs_option("sort", route)
    a = run_hip(frame, g3, gc, gd)
    sfgs_option("sort", "split")
    b = run_hip(frame, g3, gc, gd)
    for k in ("color", "depth", "alpha", "radii"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    for k in a["grads"]:
        np.testing.assert_array_equal(a["grads"][k], b["grads"][k], err_msg=k)
    np.testing.assert_array_equal(a["radii"], R.radii)
    parity.assert_image_close("color", a["color"], R.color)
