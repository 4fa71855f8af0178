"""diff_gauss -- drop-in for the reference's rasterizer extension, backed by libsfgs.so (gfx950 HIP).

Mirrors the operator API the reference binds at gaussian_renderer/__init__.py:14,40-57,132-140:

    GaussianRasterizationSettings(image_height, image_width, tanfovx, tanfovy, kernel_size,
        subpixel_offset, bg, scale_modifier, viewmatrix, projmatrix, sh_degree, campos, prefiltered, debug)
    GaussianRasterizer(raster_settings)(means3D, means2D, opacities, shs=None, colors_precomp=None,
        scales=None, rotations=None, cov3Ds_precomp=None)
        -> (color[3,H,W], depth[1,H,W], norm[3,H,W], alpha[1,H,W], radii[N] int32, extra)

Gradient contract (scene/gaussian_model.py:744-749): means2D.grad[:, :2] = dL/dmean2D in NDC units,
means2D.grad[:, 2] = magnitude of the summed absolute 2D gradients. depth = sum(T a z) / (1 - T)
(NaN where nothing was hit; consumers scrub it: train.py:229-231); norm = zeros; extra = None
(none are consumed). All arithmetic runs in the HIP library; this file only validates
arguments, owns the scratch tensors (PyTorch caching allocator) and plumbs the current stream.
"""
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from sfgs import _lib as L

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "last_counters",
           "collect_full_counters"]


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    kernel_size: float
    subpixel_offset: Optional[torch.Tensor]
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    depth_mode: int = L.DEPTH_NORMALISED  # extension: L.DEPTH_RAW returns sum(T a z)
    tile_rows: Optional[tuple] = None     # extension: (begin, end) 8-pixel tile rows to render (sfgs.shard)


import threading

# Wrapper state. The forward runs on the caller's thread, the backward on an autograd worker, and nothing stops two host
# threads from rendering at once: every piece of state below is either immutable once published (the counters dict is
# REPLACED, never mutated), keyed by device / (device, viewport) with idempotent updates, or thread-local.
_last_counters = {}
_stats = {"full": False}
_tls = threading.local()   # .pinned: (device index, stream) -> (pinned host buffer the counters are read through, event);
                           # per host thread, dies with the thread
_ncb_cache = {}  # (W, H) -> number of coarse bins
_zeros = {}      # device index -> 1-element zero tensor
_budget = {}     # device index -> scratch budget in bytes (half of the device memory)
_cap_hint = {}   # (device index, W, H) -> (duplicate capacity, per-coarse-bin capacity) to plan with
# Launch hints (include/sfgs.h SFGS_HINT_*): what the previous frames of a (device, stream) taught about the frame's
# optional kernels. `fb` is the 64-byte persistent device buffer the library leaves a frame's late statistics in
# (SfgsFrame.feedback); SFGS_HINTS=0 in the environment switches the mechanism off (tests compare both).
_hint_state = {}  # (device index, stream) -> dict(fb=tensor, huge=int, long=int, prefilled=int, bwd=int)
HINT_NO_HUGE_SPLATS, HINT_FEW_LONG_LISTS, HINT_NO_PREFILL, HINT_NO_BIG_CHUNKS = 1, 2, 4, 8
HINT_SHORT_LISTS = 16
# SHORT_LISTS (fine binning + short-list sort as one kernel) pays while the lists stay short and the coarse bins small:
# measured on the regime set it wins whenever no list exceeds the register sort (512 entries) and loses
# once lists take its long-list path (a second scan of the slab per long tile)
SHORT_LIST_MAX, SHORT_BIN_MAX = 512, 8192
PREFILL_PROBE_EVERY = 32   # backward passes between two launches of the dead-entry prefill kernel while it keeps saying no


def _hints_on():
    import os
    return os.environ.get("SFGS_HINTS", "1") != "0"


def _scratch_budget(dev):
    b = _budget.get(dev.index)
    if b is None:
        b = _budget[dev.index] = torch.cuda.get_device_properties(dev).total_memory // 2
    return b


def _zero(dev):
    z = _zeros.get(dev.index)
    if z is None:
        z = _zeros[dev.index] = torch.zeros(1, 1, 1, dtype=torch.float32, device=dev)
    return z


def _dupgrad_record_bytes(lib):
    """bytes of one per-duplicate gradient record, as the library lays them out (asked once through the C ABI)."""
    b = _stats.get("dupgrad_record_bytes")
    if b is None:
        sizes = L.SfgsRasterSizes(C_sizeof(L.SfgsRasterSizes))
        L.check(lib.sfgs_raster_sizes(1, 8, 8, 1024, 256, L.C.byref(sizes)))
        b = _stats["dupgrad_record_bytes"] = int(sizes.dupgrad_bytes) // 1024
    return b


def _pinned_counters(dev, stream):
    """This thread's pinned counter buffer + event for (device, stream): frames on different streams or from different
    host threads never share one."""
    table = getattr(_tls, "pinned", None)
    if table is None:
        table = _tls.pinned = {}
    key = (dev.index, stream.cuda_stream)
    entry = table.get(key)
    if entry is None:
        entry = table[key] = (torch.empty(16, dtype=torch.int64).pin_memory(),   # 128 bytes: sfgs_raster_forward_plan
                              torch.cuda.Event(enable_timing=False, blocking=False))
    return entry


def collect_full_counters(on=True):
    """Diagnostics switch: read the counters after the render stage (a full stream sync per frame) so that
    `last_counters()["max_tile_list"]` is filled in; off by default (the counters are then read mid-frame and
    max_tile_list reads 0)."""
    _stats["full"] = bool(on)


def last_counters():
    """Counters of the most recent forward on this process (duplicates, visible Gaussians, capacities ...)."""
    return dict(_last_counters)   # a snapshot of the dict published by the most recent forward


def _f32c(t, name, shape_tail=None):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a tensor")
    if t.dtype != torch.float32:
        raise ValueError(f"{name} must be float32, got {t.dtype}")
    _backend.check_device(t, name)
    if shape_tail is not None and tuple(t.shape[1:]) != tuple(shape_tail):
        raise ValueError(f"{name}: expected shape [N,{','.join(map(str, shape_tail))}], got {tuple(t.shape)}")
    return t.contiguous()


def _settings_tensors(settings, dev):
    """Validation of the settings tuple (above the backend seam): returns the contiguous float32 tensors
    (subpixel_offset or None, bg, viewmatrix, projmatrix, campos)."""
    H, W = int(settings.image_height), int(settings.image_width)
    if H <= 0 or W <= 0:
        raise ValueError(f"image size must be positive, got {W}x{H}")
    sub = settings.subpixel_offset
    if sub is not None:
        if tuple(sub.shape) != (H, W, 2):
            raise ValueError(f"subpixel_offset must have shape ({H},{W},2), got {tuple(sub.shape)}")
        sub = _f32c(sub, "subpixel_offset")
    bg = _f32c(settings.bg, "bg")
    view = _f32c(settings.viewmatrix, "viewmatrix")
    proj = _f32c(settings.projmatrix, "projmatrix")
    campos = _f32c(settings.campos, "campos")
    if bg.numel() != 3 or view.numel() != 16 or proj.numel() != 16 or campos.numel() != 3:
        raise ValueError("bg/campos must have 3 elements, viewmatrix/projmatrix 16")
    for t in (sub, bg, view, proj, campos):
        if t is not None and t.device != dev:
            raise ValueError("all rasterizer inputs must be on the same device")
    return sub, bg, view, proj, campos


def _frame(settings, dev, sh_coeffs, keep, hints=0, feedback=None):
    H, W = int(settings.image_height), int(settings.image_width)
    sub, bg, view, proj, campos = _settings_tensors(settings, dev)
    keep.extend([sub, bg, view, proj, campos])
    rows = getattr(settings, "tile_rows", None) or (0, 0)
    return L.SfgsFrame(C_sizeof(L.SfgsFrame), H, W, float(settings.tanfovx), float(settings.tanfovy),
                       float(settings.kernel_size), float(settings.scale_modifier), int(settings.sh_degree),
                       int(sh_coeffs), int(bool(settings.prefiltered)), int(bool(settings.debug)),
                       int(getattr(settings, "depth_mode", 0)), int(rows[0]), int(rows[1]), L.ptr(sub), L.ptr(bg), L.ptr(view), L.ptr(proj),
                       L.ptr(campos), int(hints), L.ptr(feedback))


def C_sizeof(t):
    import ctypes
    return ctypes.sizeof(t)


def _stream(dev):
    return L.C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class _Rasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, shs, colors_precomp, opacities, scales, rotations, settings, filter_3D=None):
        # filter_3D given: RAW-PARAMETER MODE (include/sfgs.h SfgsGaussians) -- opacities / scales / rotations are the
        # model's raw parameters (opacities possibly float64) and the gradients returned for them are the raw ones
        lib = L.load()
        dev = means3D.device
        N = int(means3D.shape[0])
        H, W = int(settings.image_height), int(settings.image_width)
        sh_coeffs = 0 if shs is None else int(shs.shape[1])
        keep = []
        with torch.cuda.device(dev):
            stream = _stream(dev)
            gs = _gaussians(N, means3D, scales, rotations, opacities, colors_precomp, shs, filter_3D)
            sizes = L.SfgsRasterSizes(C_sizeof(L.SfgsRasterSizes))
            # Neither the duplicate count D nor the fullest coarse bin is known before the plan: plan into a bins
            # blob sized from the previous frames (geometric growth) and redo the plan in the rare case it
            # overflowed.
            ncb = _ncb_cache.get((W, H))
            if ncb is None:
                L.check(lib.sfgs_raster_sizes(N, W, H, 0, 0, L.C.byref(sizes)))
                ncb = _ncb_cache[(W, H)] = max(int(sizes.coarse_bins), 1)
            hint = _cap_hint.get((dev.index, W, H), (0, 0))
            slots = lambda d: int(lib.sfgs_raster_slot_capacity(W, H, d))   # every tile list is 64-slot aligned
            cap = max(hint[0], slots(4 * N))
            ccap = max(hint[1], 8 * N // ncb, 256)
            need_bwd = any(ctx.needs_input_grad[:7])
            ctx.filter_3D = filter_3D
            if need_bwd and getattr(settings, "tile_rows", None):
                raise ValueError("tile_rows (band rendering) is a forward-only extension: the backward needs the whole "
                                 "frame's per-pixel state")
            # few, large allocations: the Python time before the first launch is GPU idle time
            # band rendering (tile_rows extension) leaves the pixels outside the band untouched: start from zeros there
            outs = (torch.zeros if getattr(settings, "tile_rows", None) else torch.empty)(
                5, H, W, dtype=torch.float32, device=dev)
            color, depth, alpha = outs[0:3], outs[3:4], outs[4:5]
            radii = torch.empty(N, dtype=torch.int32, device=dev)
            al = lambda n: (n + 255) // 256 * 256
            tstream = torch.cuda.current_stream(dev)
            hkey = (dev.index, tstream.cuda_stream)
            hs = None
            if _hints_on():
                hs = _hint_state.get(hkey)
                if hs is None:   # first frame on this stream: no hints yet, everything is launched
                    hs = _hint_state[hkey] = dict(fb=torch.zeros(8, dtype=torch.int64, device=dev), huge=1, long=1,
                                                  prefilled=1, bwd=0, prefill_ran=False, maxlist=1 << 30, cmax=1 << 30)
            fwd_hints = 0
            if hs is not None:
                fwd_hints = (HINT_NO_HUGE_SPLATS if hs["huge"] == 0 else 0) | (HINT_FEW_LONG_LISTS if hs["long"] == 0 else 0)
                if hs["long"] == 0 and hs["maxlist"] <= SHORT_LIST_MAX and hs["cmax"] <= SHORT_BIN_MAX:
                    fwd_hints |= HINT_SHORT_LISTS
            feedback = hs["fb"] if hs is not None else None
            tries, pool_grown = 0, False
            while True:
                L.check(lib.sfgs_raster_sizes(N, W, H, cap, ccap, L.C.byref(sizes)))
                o_tiles = al(max(sizes.geom_bytes, 1))
                o_bins = o_tiles + al(sizes.tiles_bytes)
                o_image = o_bins + al(max(sizes.bins_bytes, 1))
                total = o_image + (al(sizes.image_bytes) if need_bwd else 0)
                if total > _scratch_budget(dev):
                    # uniform per-coarse-bin slabs (ncb x fullest bin x 16 B): only a pathologically skewed frame
                    # (most Gaussians inside one 32x32-pixel bin) can get here on a 288 GB device
                    raise RuntimeError(f"rasterizer scratch of {total / 2**30:.1f} GiB exceeds half of the device memory "
                                       f"(duplicates {cap}, fullest coarse bin {ccap}, {ncb} bins)")
                scratch = torch.empty(total, dtype=torch.uint8, device=dev)
                geom, tiles, bins = scratch[:o_tiles], scratch[o_tiles:o_bins], scratch[o_bins:o_image]
                image = scratch[o_image:] if need_bwd else None
                frame = _frame(settings, dev, sh_coeffs, keep, fwd_hints, feedback)
                # plan and render are enqueued back to back. The plan's last kernel writes the frame's counters into
                # pinned host memory; an event recorded between the two stages lets the host read them -- the one
                # host wait of the frame -- WHILE the render stage runs, so the wrapper's epilogue, the caller's loss
                # and the backward's launch overlap with the compositing kernel instead of following a drained
                # stream. Both stages are redone in the rare case a capacity was exceeded (an overflowing plan is
                # memory-safe) or a launch hint turned out wrong.
                pin, ev = _pinned_counters(dev, tstream)
                L.check(lib.sfgs_raster_forward_plan(L.C.byref(frame), L.C.byref(gs), L.ptr(radii), L.ptr(geom),
                                                     geom.numel(), L.ptr(tiles), tiles.numel(), L.ptr(bins),
                                                     bins.numel(), cap, ccap, L.C.c_void_p(pin.data_ptr()), stream))
                ev.record(tstream)
                L.check(lib.sfgs_raster_forward_render(L.C.byref(frame), N, L.ptr(geom), L.ptr(tiles), L.ptr(bins),
                                                       bins.numel(), cap, ccap, -1, L.ptr(color), L.ptr(depth),
                                                       L.ptr(alpha), L.ptr(image), 0 if image is None else image.numel(),
                                                       stream))
                cnt = L.SfgsRasterCounters()
                if _stats["full"]:  # diagnostics: wait for the render too, so that max_tile_list is included
                    ev.synchronize()
                    L.check(lib.sfgs_raster_counters_decode(L.C.c_void_p(pin.data_ptr()), L.C.byref(cnt)))
                    full = L.SfgsRasterCounters()
                    L.check(lib.sfgs_raster_read_counters(L.ptr(tiles), L.C.byref(full), stream))
                    cnt.max_tile_list = full.max_tile_list
                else:
                    ev.synchronize()
                    L.check(lib.sfgs_raster_counters_decode(L.C.c_void_p(pin.data_ptr()), L.C.byref(cnt)))
                D, cmax = int(cnt.num_duplicates), int(cnt.max_coarse_bin)
                if (fwd_hints & HINT_NO_HUGE_SPLATS) and cnt.num_huge_splats:
                    fwd_hints &= ~HINT_NO_HUGE_SPLATS      # this frame HAS splats the skipped walk bins: redo with it
                    continue
                if not cnt.overflow and slots(D) <= cap and cmax <= ccap:
                    break
                tries += 1
                if tries > 24:
                    raise RuntimeError(f"rasterizer plan still overflows after {tries} attempts (duplicates {D}, capacity "
                                       f"{cap}, fullest coarse bin {cmax} of {ccap})")
                new_cap, new_ccap = max(cap, slots(int(D * 1.25) + 1024)), max(ccap, int(cmax * 1.25) + 256)
                if (new_cap, new_ccap) == (cap, ccap):
                    # the totals fit, yet a plan overflowed: one of the duplicate-index pools ran over (a few workgroups
                    # own most of the frame's duplicates): give every pool twice the room
                    new_cap = cap * 2
                    pool_grown = True
                cap, ccap = new_cap, new_ccap
            if hs is not None:
                if cnt.prev_valid:   # the previous frame's render / backward stages, as this frame's plan found them
                    hs["long"] = int(cnt.prev_long_tiles)
                    hs["maxlist"] = int(cnt.prev_max_tile_list)
                    if hs["prefill_ran"]:
                        hs["prefilled"] = int(cnt.prev_prefilled)
                hs["huge"] = int(cnt.num_huge_splats)
                hs["cmax"] = cmax
                hs["prefill_ran"] = False
            _cap_hint[(dev.index, W, H)] = (cap if pool_grown else
                                            max(slots(int(D * 1.25) + 1024), min(cap, slots(2 * D + 1024))),
                                            max(int(cmax * 1.5) + 256, min(ccap, 3 * cmax + 256)))
            global _last_counters
            _last_counters = dict(num_duplicates=D, num_duplicates_ref=int(cnt.num_duplicates_ref),
                                  num_visible=int(cnt.num_visible), max_coarse_bin=cmax,
                                  max_tile_list=int(cnt.max_tile_list), N=N, W=W, H=H, dup_capacity=cap,
                                  coarse_capacity=ccap)     # published by reference assignment (atomic)
        # normals are not produced by this rasterizer (no consumer in the reference): a zero-stride view of one
        # zero, i.e. a read-only all-zeros [3,H,W] tensor that costs no memory and no kernel
        norm = _zero(dev).expand(3, H, W)
        ctx.mark_non_differentiable(radii, norm)
        # an output the loss does not use (alpha, often depth) would otherwise reach backward() as a freshly filled zero
        # image: the library takes NULL for those instead
        ctx.set_materialize_grads(False)
        if need_bwd:
            ctx.settings, ctx.D, ctx.ccap, ctx.sh_coeffs, ctx.ndup = settings, cap, ccap, sh_coeffs, D
            ctx.big_chunks, ctx.hkey = int(cnt.num_big_chunks), hkey
            ctx.keep = keep
            ctx.has_colors, ctx.has_shs = colors_precomp is not None, shs is not None
            ctx.save_for_backward(means3D, scales, rotations, opacities, colors_precomp, shs, radii, geom, tiles, bins,
                                  image)
        return color, depth, norm, alpha, radii

    @staticmethod
    def backward(ctx, g_color, g_depth, g_norm, g_alpha, g_radii):
        lib = L.load()
        means3D, scales, rotations, opacities, colors_precomp, shs, radii, geom, tiles, bins, image = ctx.saved_tensors
        dev = means3D.device
        N = int(means3D.shape[0])
        settings, D = ctx.settings, ctx.D
        with torch.cuda.device(dev):  # autograd worker thread: select the device, use ITS current stream
            keep = []
            bwd_hints = 0
            hs = _hint_state.get(ctx.hkey) if _hints_on() else None
            if hs is not None:
                if ctx.big_chunks == 0:
                    bwd_hints |= HINT_NO_BIG_CHUNKS            # exact: this frame's own plan counted none
                # the dead-entry prefill decides on the device; while it keeps deciding "no" it is only launched every
                # PREFILL_PROBE_EVERY-th backward (either way the gradients are the same bits: tests/test_gpu_raster.py)
                hs["bwd"] += 1
                if hs["prefilled"] == 0 and hs["bwd"] % PREFILL_PROBE_EVERY != 0:
                    bwd_hints |= HINT_NO_PREFILL
                else:
                    hs["prefill_ran"] = True
            frame = _frame(settings, dev, ctx.sh_coeffs, keep, bwd_hints, hs["fb"] if hs is not None else None)
            stream = _stream(dev)
            gs = _gaussians(N, means3D, scales, rotations, opacities, colors_precomp, shs, ctx.filter_3D)
            # one allocation for all gradient tensors (views), one for the per-duplicate scratch
            K = ctx.sh_coeffs
            ncol = 3 * K if ctx.has_shs else 3
            flat = torch.empty(N * (14 + ncol) + 32, dtype=torch.float32, device=dev)
            o = 0

            def take(cols, shape):
                nonlocal o
                o = (o + 3) // 4 * 4         # 16-byte aligned regions (the kernels use float4 stores)
                v = flat[o:o + N * cols].view(shape)
                o += N * cols
                return v
            g_rot = take(4, (N, 4))
            g_means3D, g_means2D, g_scales = take(3, (N, 3)), take(3, (N, 3)), take(3, (N, 3))
            # (raw-parameter mode after the reference's reset_opacity: the raw opacity and its gradient are float64)
            g_opac = take(1, (N, 1)) if opacities.dtype == torch.float32 else torch.empty(N, 1, dtype=opacities.dtype,
                                                                                          device=dev)
            g_col = take(3, (N, 3)) if ctx.has_colors else None
            g_shs = take(3 * K, (N, K, 3)) if ctx.has_shs else None
            grads = L.SfgsGaussianGrads(C_sizeof(L.SfgsGaussianGrads), L.ptr(g_means3D), L.ptr(g_means2D),
                                        L.ptr(g_scales), L.ptr(g_rot), L.ptr(g_opac), L.ptr(g_col), L.ptr(g_shs))
            # one record per duplicate INDEX: the indices come from 8 disjoint ranges of [0, capacity) (no single allocator
            # word), so the array spans the capacity the frame was planned with; the gaps are never touched
            dupgrad = torch.empty(max((D * _dupgrad_record_bytes(lib) + 255) // 256 * 256, 1) if ctx.ndup else 1,
                                  dtype=torch.uint8, device=dev)
            gc = None if g_color is None else g_color.contiguous().float()
            gd = None if g_depth is None else g_depth.contiguous().float()
            ga = None if g_alpha is None else g_alpha.contiguous().float()
            L.check(lib.sfgs_raster_backward(L.C.byref(frame), L.C.byref(gs), L.ptr(radii), L.ptr(geom), L.ptr(tiles),
                                             L.ptr(bins), D, ctx.ccap, ctx.ndup, L.ptr(image), L.ptr(gc), L.ptr(gd), L.ptr(ga),
                                             L.ptr(dupgrad), dupgrad.numel(), L.C.byref(grads), stream))
        return g_means3D, g_means2D, g_shs, g_col, g_opac, g_scales, g_rot, None, None


def _gaussians(N, means3D, scales, rotations, opacities, colors_precomp, shs, filter_3D):
    if filter_3D is None:
        return L.SfgsGaussians(C_sizeof(L.SfgsGaussians), N, L.ptr(means3D), L.ptr(scales), L.ptr(rotations),
                               L.ptr(opacities), L.ptr(colors_precomp), L.ptr(shs))
    from sfgs.prepass import f64_mask
    return L.SfgsGaussians(C_sizeof(L.SfgsGaussians), N, L.ptr(means3D), L.ptr(scales), L.ptr(rotations),
                           L.ptr(opacities), L.ptr(colors_precomp), L.ptr(shs), L.ptr(filter_3D),
                           f64_mask(filter_3D, opacities))


class _HipBackend:
    """The product's only backend: libsfgs.so on the GPU. The argument-validation layer (GaussianRasterizer.forward,
    _f32c) sits above this seam; tests/ may swap `_backend` for a checker-backed double to drive the reference's real
    render() glue through that layer on a GPU-less host -- nothing in this package ever does."""
    name = "hip"

    @staticmethod
    def check_device(t, name):
        if not t.is_cuda:
            raise ValueError(f"{name} must live on the GPU (got {t.device}); this rasterizer has no CPU path")

    @staticmethod
    def rasterize(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, raster_settings):
        return _Rasterize.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, raster_settings)

    @staticmethod
    def rasterize_raw(means3D, means2D, shs, colors_precomp, raw_opacity, raw_scaling, raw_rotation, filter_3D,
                      raster_settings):
        """Raw-parameter mode: the activations + 3D filter of sfgs.prepass run inside preprocess / preprocess_bwd."""
        return _Rasterize.apply(means3D, means2D, shs, colors_precomp, raw_opacity, raw_scaling, raw_rotation,
                                raster_settings, filter_3D)


_backend = _HipBackend


def rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, raster_settings):
    return _backend.rasterize(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3Ds_precomp=None):
        if (shs is None) == (colors_precomp is None):
            raise ValueError("Please provide exactly one of either SHs or precomputed colors!")
        if cov3Ds_precomp is not None:
            raise ValueError("cov3Ds_precomp is not supported: that path is dead in the reference "
                             "(gaussian_renderer/__init__.py:138 calls scales.float() unconditionally)")
        if scales is None or rotations is None:
            raise ValueError("Please provide scales and rotations")
        N = int(means3D.shape[0])
        means3D = _f32c(means3D, "means3D")
        if means3D.dim() != 2 or means3D.shape[1] != 3:
            raise ValueError("means3D must be [N,3]")
        if means2D is None:
            means2D = torch.zeros_like(means3D)
        # Deferred results of sfgs.prepass's patched getters (render() only passed them through .float()): the library
        # applies the activations itself, from the raw parameters. Anything else Deferred is materialised here.
        from sfgs import prepass
        raw = prepass.raw_parameters(scales, opacities, rotations) if hasattr(_backend, "rasterize_raw") else None
        if raw is None:
            scales, opacities, rotations = (prepass.materialise(t) for t in (scales, opacities, rotations))
            scales = _f32c(scales, "scales", (3,))
            rotations = _f32c(rotations, "rotations", (4,))
            opacities = _f32c(opacities, "opacities")
            if opacities.numel() != N:
                raise ValueError(f"opacities must have N={N} elements")
            opacities = opacities.reshape(N, 1)
        else:
            scales, opacities, rotations, filter_3D = raw      # validated by prepass (dtypes, shapes, device)
        colors_precomp = _f32c(colors_precomp, "colors_precomp", (3,))
        if shs is not None:
            shs = _f32c(shs, "shs")
            if shs.dim() != 3 or shs.shape[2] != 3:
                raise ValueError("shs must be [N,K,3]")
            deg = int(self.raster_settings.sh_degree)
            if shs.shape[1] < (deg + 1) ** 2 or shs.shape[1] not in (1, 4, 9, 16):
                raise ValueError(f"shs has {shs.shape[1]} coefficients per Gaussian; expected (max_degree + 1)^2 in "
                                 f"(1, 4, 9, 16) and at least {(deg + 1) ** 2} for active degree {deg}")
        for name, t in (("scales", scales), ("rotations", rotations), ("colors_precomp", colors_precomp), ("shs", shs)):
            if t is not None and (t.shape[0] != N or t.device != means3D.device):
                raise ValueError(f"{name}: first dimension / device must match means3D")
        _settings_tensors(self.raster_settings, means3D.device)   # shapes / dtypes / devices of the 14-field tuple
        if raw is not None:
            color, depth, norm, alpha, radii = _backend.rasterize_raw(means3D, means2D, shs, colors_precomp, opacities,
                                                                      scales, rotations, filter_3D, self.raster_settings)
        else:
            color, depth, norm, alpha, radii = rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities,
                                                                   scales, rotations, self.raster_settings)
        return color, depth, norm, alpha, radii, None
