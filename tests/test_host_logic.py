"""CPU tests of the product's host-side logic and of its math header (compiled by g++ into tests/host_check):
projection / radii / opacity-aware binning / sort order / per-pixel compositing / chain rule against the
oracle, plus the C-ABI contract (symbols, struct layouts, GPU-free entry points). No GPU needed."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest
import torch

import hostcheck
import parity
from oracle import oracle as orc
from sfgs import _lib as L
from sfgs.synth import scene, upstream_grads

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {
    "precomp": dict(n=3000, W=200, H=120, kw=dict(zrange=(4., 8.), scale_range=(0.01, 0.2))),
    "sh3_jitter": dict(n=8000, W=256, H=160, kw=dict(zrange=(250., 350.), scale_range=(0.2, 3.0), mode="sh",
                                                      sh_degree=3, jitter=True)),
    "ragged_big": dict(n=1500, W=130, H=77, kw=dict(zrange=(2., 50.), scale_range=(0.01, 2.0), mode="sh", sh_degree=1)),
    "low_elevation": dict(n=6000, W=240, H=136, kw=dict(zrange=(20., 400.), scale_range=(0.05, 2.0), pitch_deg=45.0)),
}


@pytest.mark.parametrize("case", list(CASES))
def test_product_math_matches_oracle(case):
    c = CASES[case]
    frame, g = scene(c["n"], c["W"], c["H"], seed=1, **c["kw"])
    R = orc.OracleRender(frame, **g)
    gc, gd = upstream_grads(c["W"], c["H"], 0)
    gd = gd.clone()
    gd[torch.from_numpy(np.isnan(R.depth))] = 0
    G = R.backward(gc, gd)
    o = hostcheck.render(frame, g["means3D"], g["scales"], g["rotations"], g["opacities"], g["colors_precomp"],
                         g["shs"], gc, gd, None, backward=True)
    np.testing.assert_array_equal(o["radii"], R.radii)                      # integers: bit-exact
    assert int(o["counters"][1]) == R.num_duplicates                        # the reference's tiles_touched total
    assert int(o["counters"][2]) == R.num_visible
    # the opacity-aware 8x8 binning must never drop a contributor: images agree to float round-off
    for name in ("color", "alpha", "depth"):
        parity.assert_image_close(name, o[name], getattr(R, name))
    for k in G:
        parity.assert_grad_close(k, o["grads"][k], G[k])


def test_raw_depth_mode_and_background():
    frame, g = scene(2000, 160, 96, seed=4, zrange=(4., 8.), scale_range=(0.01, 0.2))
    frame["depth_mode"] = 1
    frame["bg"] = torch.tensor([0.2, 0.5, 0.9])
    R = orc.OracleRender(frame, **g)
    o = hostcheck.render(frame, g["means3D"], g["scales"], g["rotations"], g["opacities"], g["colors_precomp"], None)
    for name in ("color", "alpha", "depth"):
        parity.assert_image_close(name, o[name], getattr(R, name))
    assert np.isfinite(o["depth"]).all()


def test_binning_is_tighter_than_reference_rule_where_it_can_be():
    """low-opacity splats reach alpha >= 1/255 on a smaller area than the 3-sigma square: D_eff counts
    8x8 tiles, D_ref 16x16 tiles; per unit of covered area the binning must not exceed the reference's."""
    frame, g = scene(20000, 320, 200, seed=2, zrange=(250., 350.), scale_range=(0.2, 3.0), opacity_range=(0.01, 0.05))
    o = hostcheck.render(frame, g["means3D"], g["scales"], g["rotations"], g["opacities"], g["colors_precomp"], None)
    d_eff, d_ref = int(o["counters"][0]), int(o["counters"][1])
    assert d_eff * 64 < d_ref * 256  # pixel area binned < pixel area the reference would composite


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "sfgs.h")).read()
    declared = set(re.findall(r"\b(sfgs_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(L.SYMBOLS), (declared ^ set(L.SYMBOLS))
    lib = L.load()  # binds every symbol, checks the ABI version
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.sfgs_abi_version() == L.ABI_VERSION


def test_struct_layouts_match_the_c_header(tmp_path):
    src = tmp_path / "layout.c"
    names = ["SfgsFrame", "SfgsGaussians", "SfgsGaussianGrads", "SfgsRasterSizes", "SfgsRasterCounters"]
    body = "\n".join(f'  printf("{n} %zu\\n", sizeof({n}));' for n in names)
    offs = "\n".join(f'  printf("SfgsFrame.{f} %zu\\n", offsetof(SfgsFrame, {f}));' for f, _ in L.SfgsFrame._fields_)
    src.write_text(f'#include <stdio.h>\n#include <stddef.h>\n#include "sfgs.h"\nint main(void) {{\n{body}\n{offs}\n  return 0;\n}}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    for n in names:
        assert int(out[n]) == C.sizeof(getattr(L, n)), n
    for f, _ in L.SfgsFrame._fields_:
        assert int(out[f"SfgsFrame.{f}"]) == getattr(L.SfgsFrame, f).offset, f


def test_gpu_free_entry_points_and_error_convention():
    lib = L.load()
    sizes = L.SfgsRasterSizes(C.sizeof(L.SfgsRasterSizes))
    assert lib.sfgs_raster_sizes(2_000_000, 1920, 1080, 7_000_000, 8192, C.byref(sizes)) == 0
    assert sizes.geom_bytes >= 2_000_000 * 56 and sizes.bins_bytes >= 7_000_000 * 24 + sizes.coarse_bins * 8192 * 16
    assert sizes.coarse_bins == 60 * 34
    # 48-byte records, one flag byte per duplicate index (256-byte aligned regions), one line of zeros
    assert sizes.dupgrad_bytes == 7_000_000 * 48 + (7_000_000 + 255) // 256 * 256 + 256 and sizes.image_bytes >= 1920 * 1080 * 12
    # errors: negative status + thread-local message, never an exception or exit
    bad = L.SfgsRasterSizes(4)
    assert lib.sfgs_raster_sizes(10, 64, 64, 0, 0, C.byref(bad)) == -1
    assert b"struct_size" in lib.sfgs_last_error()
    assert lib.sfgs_raster_sizes(-1, 64, 64, 0, 0, C.byref(sizes)) == -1
    assert lib.sfgs_raster_sizes(10, 64, 64, 1 << 33, 0, C.byref(sizes)) == -4  # > 2^32 duplicates unsupported
    with pytest.raises(RuntimeError, match="libsfgs error"):
        L.check(lib.sfgs_raster_sizes(-1, 64, 64, 0, 0, C.byref(sizes)))
    # a NULL frame is rejected before any HIP call
    assert lib.sfgs_raster_forward_plan(None, None, None, None, 0, None, 0, None, 0, 0, 0, None, None) == -1
    assert lib.sfgs_ssim_scratch_bytes(1, 3, 1080, 1920, 1) > 3 * 3 * 1080 * 1920 * 4
    # one partial sum per 32 x 22 output tile (60 x 50 x 3 at 1080p) in front of the three derivative maps
    assert lib.sfgs_ssim_scratch_bytes(1, 3, 1080, 1920, 0) == 60 * 50 * 3 * 4 + (-(60 * 50 * 3 * 4)) % 256
    # a plane of 2^30 pixels or more is refused before any HIP call (the kernels address a plane through a 32-bit buffer
    # descriptor); the pointers are never dereferenced
    dummy = C.c_float(0.0)
    fp = C.cast(C.byref(dummy), C.c_void_p)
    assert lib.sfgs_ssim_forward(fp, fp, 1, 1, 32768, 32768, None, fp, fp, C.c_size_t(1 << 62), 0, None) == -4
    assert b"2^30" in lib.sfgs_last_error()
    assert lib.sfgs_ssim_backward(fp, fp, 1, 1, 32768, 32768, fp, fp, fp, None) == -4
    assert lib.sfgs_knn_scratch_bytes(1000) == 0
    assert lib.sfgs_profile_kernel_count() >= 10 and lib.sfgs_profile_kernel_name(1) == b"preprocess"


def test_operator_argument_validation_without_gpu():
    from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer
    from fused_ssim import fused_ssim
    from simple_knn._C import distCUDA2
    frame, g = scene(8, 32, 32, seed=0)
    s = GaussianRasterizationSettings(32, 32, frame["tanfovx"], frame["tanfovy"], 0.1, None, frame["bg"], 1.0,
                                      frame["view"], frame["proj"], 0, frame["campos"], False, False)
    assert s._fields[:14] == ("image_height", "image_width", "tanfovx", "tanfovy", "kernel_size", "subpixel_offset",
                              "bg", "scale_modifier", "viewmatrix", "projmatrix", "sh_degree", "campos", "prefiltered",
                              "debug")  # field order of gaussian_renderer/__init__.py:40-55
    r = GaussianRasterizer(raster_settings=s)
    with pytest.raises(ValueError, match="exactly one"):
        r(means3D=g["means3D"], means2D=None, opacities=g["opacities"], scales=g["scales"], rotations=g["rotations"])
    with pytest.raises(ValueError, match="cov3Ds_precomp"):
        r(means3D=g["means3D"], means2D=None, opacities=g["opacities"], colors_precomp=g["colors_precomp"],
          scales=g["scales"], rotations=g["rotations"], cov3Ds_precomp=torch.zeros(8, 6))
    with pytest.raises(ValueError, match="GPU"):  # no CPU fallback, by design
        r(means3D=g["means3D"], means2D=None, opacities=g["opacities"], colors_precomp=g["colors_precomp"],
          scales=g["scales"], rotations=g["rotations"])
    with pytest.raises(ValueError):
        fused_ssim(torch.zeros(1, 3, 8, 8), torch.zeros(1, 3, 8, 8))
    with pytest.raises(ValueError):
        distCUDA2(torch.zeros(10, 3))


def test_launch_hint_bits_match_the_c_header():
    """The wrapper's HINT_* constants are the header's SFGS_HINT_* bits (distinct single bits), and the SHORT_LISTS route
    is only asked for inside what the fused kernel's register sort takes (lists of at most 512 entries)."""
    import diff_gauss
    hdr = open(os.path.join(ROOT, "include", "sfgs.h")).read()
    bits = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+SFGS_HINT_([A-Z_0-9]+)\s+(\d+)u", hdr)}
    assert bits == {"NO_HUGE_SPLATS": diff_gauss.HINT_NO_HUGE_SPLATS, "FEW_LONG_LISTS": diff_gauss.HINT_FEW_LONG_LISTS,
                    "NO_PREFILL": diff_gauss.HINT_NO_PREFILL, "NO_BIG_CHUNKS": diff_gauss.HINT_NO_BIG_CHUNKS,
                    "SHORT_LISTS": diff_gauss.HINT_SHORT_LISTS, "MEDIUM_LISTS": diff_gauss.HINT_MEDIUM_LISTS,
                    "TILE_ORDER": diff_gauss.HINT_TILE_ORDER, "LISTS_768": diff_gauss.HINT_LISTS_768}
    vals = sorted(bits.values())
    assert all(v & (v - 1) == 0 for v in vals) and len(set(vals)) == len(vals)
    assert diff_gauss.SHORT_LIST_MAX <= 512 and diff_gauss.MEDIUM_LIST_MAX <= 1024   # what select_sort_kernel<512 / 1024> sort
    assert int(re.search(r"#define\s+SFGS_ABI_VERSION\s+(\d+)", hdr).group(1)) == L.ABI_VERSION


def test_capacity_and_huge_splat_hints_follow_their_rules():
    """The wrapper's per-viewport capacity hints and its huge-splat state (diff_gauss: _next_capacities, _next_huge) as pure
    functions: headroom over the frame's need, no growth on their own, slow shrinking (3 % per frame, floors at 2x / 3x
    the need), a quiet period after a frame with huge splats. The GPU side: tests/test_gpu_launch_hints.py."""
    import diff_gauss as dg
    over = 1088 * 2040
    cap, ccap = dg._next_capacities(30_000_000, 4096, 6_000_000, 0, over, False)
    assert cap == int(30_000_000 * 0.97) and ccap == int(4096 * 0.97)          # shrinking slowly, not snapping to the need
    assert dg._next_capacities(10_000_000, 4096, 6_000_000, 0, over, False)[0] == 10_000_000   # below the 2x floor: kept
    c, cc = 50_000_000, 100_000
    for _ in range(400):                                                        # ... down to the floors
        c, cc = dg._next_capacities(c, cc, 6_000_000, 1000, over, False)
    assert c == 2 * 6_000_000 + 1024 + over and cc == 3 * 1000 + 256
    cap, ccap = dg._next_capacities(8_000_000, 300, 7_500_000, 290, over, False)  # a tight fit: 25 % / 50 % of headroom
    assert cap == int(7_500_000 * 1.25) + 1024 + over and ccap == int(290 * 1.5) + 256
    assert dg._next_capacities(8_000_000, 300, 100, 0, over, True)[0] == 8_000_000   # a pool overran: keep the doubled room
    assert dg._next_capacities(0, 0, 0, 0, over, False) == (1024 + over, 256)
    h = 1                                                                       # first frame: nothing known, walk launched
    seq = []
    for n in (0, 0, 5, 0, 0, 0):
        h = dg._next_huge(h, n)
        seq.append(h)
    q = dg.HUGE_QUIET_FRAMES
    assert seq == [0, 0, q, q - 1, q - 2, q - 3] and q >= 8
    # the dead-entry kernel of the backward: launched while the state is > 0; one "yes" keeps it in for PREFILL_QUIET "no"s
    p, seq = 1, []
    for yes in (False, False, True, False, False, True, False):
        p = dg._next_prefilled(p, yes)
        seq.append(p)
    q = dg.PREFILL_QUIET
    assert seq == [0, 0, q, q - 1, q - 2, q, q - 1] and q >= 64


def test_sort_route_hints_follow_the_list_statistics():
    """diff_gauss._sort_hints: which fused-sort form the next frame asks for, from the previous frame's list statistics (the
    measured cases behind the rule: profiles/r6_sort_route_policy.txt). Performance only -- every route builds the same lists."""
    import diff_gauss as dg
    S, M, T = dg.HINT_SHORT_LISTS, dg.HINT_SHORT_LISTS | dg.HINT_MEDIUM_LISTS, 32400
    f = dg._sort_hints   # (long_tiles, maxlist, cmax, over512, mean_list, tiles)
    assert f(0, 277, 3000, 0, 214, T) == S                         # headline
    assert f(0, 277, 9000, 0, 214, T) == 0                         # ... with a huge coarse bin: split
    assert f(25984, 652, 6000, 25984, 536, T) == M | dg.HINT_LISTS_768   # low elevation: 80 % of the tiles beyond 512, none beyond 768
    assert f(32400, 1023, 6000, 32400, 858, T) == M                # dense 8 M
    assert f(7179, 3361, 20000, 7179, 269, T) == M                 # opaque city at 25 degrees: a few very long lists, short mean
    assert f(4196, 1998, 20000, 4196, 240, T) == M                 # ... at 60 degrees (13 % of the tiles)
    assert f(2412, 852, 5000, 2412, 164, T) == S                   # orbit at 25 degrees: 7 % of the tiles beyond 512 -> the 512 form + long-list kernels
    assert f(2921, 1575, 20000, 2921, 203, T) == S                 # city at 89 degrees: 9 %
    assert f(658, 1269, 6452, 658, 130, T) == S                    # 1 M Gaussians, city at 45 degrees: 2 %
    assert f(3000, 1600, 9000, 3000, 900, T) == 0                  # ... but not when the mean list is long
    assert f(32400, 3000, 50000, 32400, 1700, T) == 0              # 16 M Gaussians: most lists beyond 1 024 -> split
    assert f(32400, 1676, 50000, 32400, 1311, T) == 0              # screen-filling splats
    assert f(7179, 3361, 20000, 7179, 269, T, medium_on=False) == 0
    # longest-first tile order: only for frames whose longest list is several times the mean, and big enough to matter
    o, O = dg._order_hint, dg.HINT_TILE_ORDER
    assert o(277, 214, 6_945_415) == 0 and o(652, 536, 17_361_927) == 0 and o(1023, 858, 27_801_544) == 0   # headline, low elevation, dense
    assert o(1575, 203, 6_590_021) == O and o(3361, 269, 8_722_904) == O and o(852, 164, 5_321_148) == O    # city 89 / 25, orbit 25
    assert o(3, 0.1, 3530) == 0 and o(1 << 30, 214, 6_945_415) == 0                                          # tiny scene; first frame (nothing known)


def test_route_options_are_set_through_the_abi_not_the_environment():
    """ABI 16: the library no longer calls getenv() on every render. The route options are process-wide
    atomics, read from the environment ONCE at load time and changed only through sfgs_set_option (host code: runs here)."""
    import subprocess
    import sys
    for key, values in (("sort", ["auto", "fused", "fused768", "fused1024", "split"]), ("plan_scan", ["fused", "separate"]),
                        ("binning", ["auto", "direct"]), ("prefill", ["auto", "always", "never"]), ("knn", ["auto", "brute"]),
                        ("tile_order", ["auto", "always", "never"])):
        first = L.get_option(key)
        assert first == values[0]                       # the defaults (this process's environment sets none)
        for v in values[::-1]:
            assert L.set_option(key, v) in values and L.get_option(key) == v
        with pytest.raises(RuntimeError, match="has no value"):
            L.set_option(key, "bogus")
        assert L.get_option(key) == values[0]           # a rejected value changes nothing
    with pytest.raises(RuntimeError, match="unknown option"):
        L.set_option("no_such_option", "1")
    with pytest.raises(KeyError):
        L.get_option("no_such_option")
    # the environment is honoured at LOAD time (a fresh process), never afterwards
    os.environ["SFGS_SORT"] = "split"
    try:
        assert L.get_option("sort") == "auto"
        code = ("import sys; sys.path.insert(0, %r); from sfgs import _lib as L; "
                "print(L.get_option('sort'), L.get_option('prefill'))" % os.path.join(ROOT, "skyfall-gs_amd"))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                           env=dict(os.environ, SFGS_PREFILL="never"))
        assert r.stdout.split() == ["split", "never"], r.stdout + r.stderr
    finally:
        del os.environ["SFGS_SORT"]
    src = "".join(open(os.path.join(ROOT, "skyfall-gs_amd", "csrc", f)).read() for f in ("raster_fwd.hip", "raster_bwd.hip", "composite_bwd.hip", "knn.hip"))
    assert "getenv" not in src


def _bench_module():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("sfgs_bench_for_tests", os.path.join(root, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_bench_never_falls_back_to_gloo_where_rccl_can_run(capsys):
    """`bench.py --gpus N` on a node with a GPU per rank can only ever run its collectives over RCCL
    (backend "nccl"); the gloo hook is a TEST hook for boxes with fewer GPUs than ranks, refused otherwise, never silent."""
    b = _bench_module()
    assert b.pick_backend(8, 8, {}) == "nccl"
    assert b.pick_backend(2, 1, {}) == "nccl"                  # no silent fallback either: nccl is asked for and fails loudly
    for world, gpus in ((2, 2), (8, 8), (2, 8)):
        with pytest.raises(SystemExit, match="refused"):
            b.pick_backend(world, gpus, {"SFGS_BENCH_BACKEND": "gloo"})
    assert b.pick_backend(2, 1, {"SFGS_BENCH_BACKEND": "gloo"}) == "gloo"      # the hook: fewer GPUs than ranks
    assert "TEST HOOK" in capsys.readouterr().err                             # ... and it says so
    assert b.pick_backend(1, 1, {"SFGS_BENCH_BACKEND": "gloo"}) == "gloo"      # world 1 (--force-dist tests)
    with pytest.raises(SystemExit):
        b.pick_backend(2, 2, {"SFGS_BENCH_BACKEND": "mpi"})


def test_bench_step_traffic_sums_the_steady_state_kernels():
    """roofline_step.traffic = the committed PMC bytes of the kernels a steady-state step launches."""
    b = _bench_module()
    t, src = b.load_traffic(2_000_000, 1920, 1080, False)
    assert src and src["file"].startswith("profiles/")
    for k in ("composite_bwd", "composite_fwd", "preprocess", "preprocess_bwd", "bin_scatter", "select_sort"):
        assert t.get(k, 0) > 0, k
    assert b.load_traffic(1000, 64, 64, False) == ({}, None)
