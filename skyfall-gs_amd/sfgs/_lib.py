"""ctypes binding of libsfgs.so (C ABI: include/sfgs.h). Host-side plumbing only: every kernel lives
in csrc/*.hip. There is NO CPU fallback: if the HIP library is missing, importing an operator fails
loudly (build it with `python __graft_entry__.py` or `make -C skyfall-gs_amd/csrc`)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SFGS_LIB") or os.path.join(_HERE, "libsfgs.so")   # SFGS_LIB: experiment builds (tools/)
ABI_VERSION = 18

SFGS_OK = 0
DEPTH_NORMALISED, DEPTH_RAW = 0, 1


class SfgsFrame(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("image_height", C.c_int32), ("image_width", C.c_int32),
                ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("kernel_size", C.c_float),
                ("scale_modifier", C.c_float), ("sh_degree", C.c_int32), ("sh_coeffs", C.c_int32),
                ("prefiltered", C.c_int32), ("debug", C.c_int32), ("depth_mode", C.c_int32),
                ("tile_row_begin", C.c_int32), ("tile_row_end", C.c_int32),
                ("subpixel_offset", C.c_void_p), ("bg", C.c_void_p), ("viewmatrix", C.c_void_p),
                ("projmatrix", C.c_void_p), ("campos", C.c_void_p), ("launch_hints", C.c_uint32),
                ("feedback", C.c_void_p)]


class SfgsGaussians(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("count", C.c_int32), ("means3D", C.c_void_p), ("scales", C.c_void_p),
                ("rotations", C.c_void_p), ("opacities", C.c_void_p), ("colors_precomp", C.c_void_p),
                ("shs", C.c_void_p), ("filter_3D", C.c_void_p), ("raw_f64_mask", C.c_int32),   # raw-parameter mode
                ("sh_dirs", C.c_void_p), ("shs_channel_major", C.c_int32),                      # eval_sh-folded colour path
                ("shs_rest", C.c_void_p),                                                       # split SH storage (ABI 13)
                ("sh_centers", C.c_void_p)]                                                     # directions from centres (ABI 15)


class SfgsGaussianGrads(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("means3D", C.c_void_p), ("means2D", C.c_void_p),
                ("scales", C.c_void_p), ("rotations", C.c_void_p), ("opacities", C.c_void_p),
                ("colors_precomp", C.c_void_p), ("shs", C.c_void_p), ("sh_dirs", C.c_void_p), ("shs_rest", C.c_void_p)]


class SfgsRasterSizes(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("geom_bytes", C.c_size_t), ("tiles_bytes", C.c_size_t),
                ("bins_bytes", C.c_size_t), ("image_bytes", C.c_size_t), ("dupgrad_bytes", C.c_size_t),
                ("coarse_bins", C.c_int64)]


class SfgsScratchLayout(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("reserved", C.c_uint32), ("geom_offset", C.c_size_t),
                ("tiles_offset", C.c_size_t), ("bins_offset", C.c_size_t), ("image_offset", C.c_size_t),
                ("total_bytes", C.c_size_t), ("dupgrad_bytes", C.c_size_t), ("coarse_bins", C.c_int64),
                ("slot_overhead", C.c_int64)]


class SfgsAdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("count", C.c_int64), ("neg_step_size", C.c_double), ("one_minus_beta1", C.c_double),
                ("beta2", C.c_double), ("one_minus_beta2", C.c_double), ("bias_correction2_sqrt", C.c_double),
                ("eps", C.c_double), ("weight_decay", C.c_double), ("flags", C.c_uint32), ("reserved", C.c_uint32)]


ADAM_F64 = 1

class SfgsCompactTensor(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("row_bytes", C.c_int64)]


class SfgsDensifyTensor(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("row_bytes", C.c_int64), ("zero_new_rows", C.c_int32),
                ("reserved", C.c_int32)]


class SfgsRasterCounters(C.Structure):
    _fields_ = [("num_duplicates", C.c_int64), ("num_duplicates_ref", C.c_int64), ("num_visible", C.c_int64),
                ("max_tile_list", C.c_int64), ("overflow", C.c_int64), ("max_coarse_bin", C.c_int64),
                ("num_huge_splats", C.c_int64), ("num_big_chunks", C.c_int64), ("prev_valid", C.c_int64),
                ("prev_long_tiles", C.c_int64), ("prev_max_tile_list", C.c_int64), ("prev_prefilled", C.c_int64),
                ("prev_tiles_over_512", C.c_int64), ("max_bin_items", C.c_int64)]


# every symbol include/sfgs.h declares: name -> (restype, argtypes)
_V, _I32, _I64, _SZ = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
SYMBOLS = {
    "sfgs_abi_version": (C.c_int, []),
    "sfgs_last_error": (C.c_char_p, []),
    "sfgs_profile_enable": (C.c_int, [_I32]),
    "sfgs_profile_select": (C.c_int, [C.c_uint64]),
    "sfgs_set_option": (C.c_int, [C.c_char_p, C.c_char_p]),
    "sfgs_get_option": (C.c_char_p, [C.c_char_p]),
    "sfgs_profile_kernel_count": (C.c_int, []),
    "sfgs_profile_kernel_name": (C.c_char_p, [_I32]),
    "sfgs_profile_collect": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64), _I32]),
    "sfgs_box_probe": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), _V]),
    "sfgs_raster_sizes": (C.c_int, [_I32, _I32, _I32, _I64, _I64, C.POINTER(SfgsRasterSizes)]),
    "sfgs_raster_slot_capacity": (C.c_int64, [_I32, _I32, _I64]),
    "sfgs_raster_forward_plan": (C.c_int, [C.POINTER(SfgsFrame), C.POINTER(SfgsGaussians), _V, _V, _SZ, _V, _SZ, _V, _SZ,
                                            _I64, _I64, _V, _V]),
    "sfgs_raster_counters_decode": (C.c_int, [_V, C.POINTER(SfgsRasterCounters)]),
    "sfgs_raster_read_counters": (C.c_int, [_V, C.POINTER(SfgsRasterCounters), _V]),
    "sfgs_raster_read_counters_pinned": (C.c_int, [_V, _V, C.POINTER(SfgsRasterCounters), _V]),
    "sfgs_raster_forward_render": (C.c_int, [C.POINTER(SfgsFrame), _I32, _V, _V, _V, _SZ, _I64, _I64, _I64, _V, _V, _V,
                                              _V, _SZ, _V]),
    "sfgs_raster_scratch_layout": (C.c_int, [_I32, _I32, _I32, _I64, _I64, _I32, C.POINTER(SfgsScratchLayout)]),
    "sfgs_raster_forward": (C.c_int, [C.POINTER(SfgsFrame), C.POINTER(SfgsGaussians), _V, _V, _SZ, _I64, _I64, _I32, _V, _V,
                                       _V, _V, _V, _V]),
    "sfgs_raster_backward_scratch": (C.c_int, [C.POINTER(SfgsFrame), C.POINTER(SfgsGaussians), _V, _V, _SZ, _I64, _I64, _I64,
                                                _V, _V, _V, _V, _SZ, C.POINTER(SfgsGaussianGrads), _V]),
    "sfgs_raster_plan_export": (C.c_int, [C.POINTER(SfgsFrame), _I32, _V, _V, _V, _I64, _I64, _I64, _V, _V, _V, _V]),
    "sfgs_raster_plan_merge": (C.c_int, [C.POINTER(SfgsFrame), _I32, C.POINTER(C.c_int32), C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _I64, _V, _SZ, _V, _SZ, _V, _SZ,
                                          _I64, _I64, _V]),
    "sfgs_raster_backward": (C.c_int, [C.POINTER(SfgsFrame), C.POINTER(SfgsGaussians), _V, _V, _V, _V, _I64, _I64, _I64, _V,
                                        _V, _V, _V, _V, _SZ, C.POINTER(SfgsGaussianGrads), _V]),
    "sfgs_ssim_scratch_bytes": (_SZ, [_I32, _I32, _I32, _I32, _I32]),
    "sfgs_ssim_forward": (C.c_int, [_V, _V, _I32, _I32, _I32, _I32, _V, _V, _V, _SZ, _I32, _V]),
    "sfgs_ssim_backward": (C.c_int, [_V, _V, _I32, _I32, _I32, _I32, _V, _V, _V, _V]),
    "sfgs_knn_scratch_bytes": (_SZ, [_I32]),
    "sfgs_knn_dist2": (C.c_int, [_V, _I32, _V, _V, _SZ, _V]),
    "sfgs_filter3d_scratch_bytes": (_SZ, [_I32]),
    "sfgs_filter3d": (C.c_int, [_V, _I32, _V, _I32, C.c_double, _V, _V, _SZ, _V]),
    "sfgs_densify_stats": (C.c_int, [_I32, _V, _V, _V, _V, _V, _V, _V]),
    "sfgs_adam_step": (C.c_int, [C.POINTER(SfgsAdamTensor), _I32, _V]),
    "sfgs_sh_eval_forward": (C.c_int, [_I32, _I32, _I32, _V, _V, _V, _V]),
    "sfgs_sh_eval_backward": (C.c_int, [_I32, _I32, _I32, _V, _V, _V, _V, _V, _V]),
    "sfgs_compact_scratch_bytes": (_SZ, [_I64]),
    "sfgs_compact_plan": (C.c_int, [_V, _I64, _V, _SZ, C.POINTER(C.c_int64), _V]),
    "sfgs_compact_rows": (C.c_int, [_V, _I64, _V, C.POINTER(SfgsCompactTensor), _I32, _V]),
    "sfgs_select_scratch_bytes": (_SZ, []),
    "sfgs_select_kth": (C.c_int, [_V, _I64, _V, _V, _V, _SZ, _V]),
    "sfgs_densify_scratch_bytes": (_SZ, [_I64]),
    "sfgs_densify_decide": (C.c_int, [_I64, _V, _V, _V, _V, _I32, _V, C.c_float, C.c_float, C.c_double, C.c_float,
                                      C.c_float, _I32, _V, _SZ, C.POINTER(C.c_int64), _V]),
    "sfgs_densify_masks": (C.c_int, [_I64, _V, _V, _V, _V, _V]),
    "sfgs_densify_gather": (C.c_int, [_I64, _V, C.POINTER(C.c_int64), C.POINTER(SfgsDensifyTensor), _I32, _V]),
    "sfgs_densify_children": (C.c_int, [_I64, _V, C.POINTER(C.c_int64), _V, _V, _V, _V, C.c_int32, _V, _V, _V]),
    "sfgs_prepass_forward": (C.c_int, [_I32, _V, _V, _V, _V, _I32, _V, _V, _V, _V]),
    "sfgs_prepass_backward": (C.c_int, [_I32, _V, _V, _V, _V, _I32, _V, _V, _V, _V, _V, _V, _V]),
}

_lib = None


def load():
    """Load libsfgs.so and bind every symbol of the ABI. Raises if the library is missing or stale."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP extension has not been built (run `python __graft_entry__.py` "
                "or `make -C skyfall-gs_amd/csrc`). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        v = lib.sfgs_abi_version()
        if v != ABI_VERSION:
            raise ImportError(f"libsfgs.so ABI version {v}, binding expects {ABI_VERSION}")
        _lib = lib
    return _lib


def check(rc):
    if rc != SFGS_OK:
        raise RuntimeError(f"libsfgs error {rc}: {load().sfgs_last_error().decode(errors='replace')}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def set_option(key, value):
    """Process-wide route option of the library (include/sfgs.h: sfgs_set_option): "sort", "plan_scan", "binning",
    "prefill", "knn", "tile_order". Tests and A/B runs only -- every route builds bit-identical results. Returns the previous value."""
    lib = load()
    old = lib.sfgs_get_option(str(key).encode())
    check(lib.sfgs_set_option(str(key).encode(), str(value).encode()))   # raises for an unknown key or value
    return old.decode()


def get_option(key):
    """The library's CURRENT value (one ctypes call: no Python-side mirror that a C caller or a second binding could leave
    stale -- ADVICE r5)."""
    raw = load().sfgs_get_option(str(key).encode())
    if raw is None:
        raise KeyError(f"libsfgs.so has no option {key!r}")
    return raw.decode()


def refresh_options():   # kept for callers of the ABI-16 binding: there is no cache any more
    pass


def box_probe(stream=None):
    """-> (valu_tflops, sclk_mhz_effective) of this box (include/sfgs.h: sfgs_box_probe)."""
    a, b = C.c_double(0.0), C.c_double(0.0)
    check(load().sfgs_box_probe(C.byref(a), C.byref(b), stream))
    return a.value, b.value


def profile_enable(on=True):
    check(load().sfgs_profile_enable(int(bool(on))))


def profile_select(names=None):
    """Time only the kernels whose names are given (None = all)."""
    lib = load()
    mask = 0
    for i in range(lib.sfgs_profile_kernel_count()):
        if names is None or lib.sfgs_profile_kernel_name(i).decode() in names:
            mask |= 1 << i
    check(lib.sfgs_profile_select(mask))


def profile_collect():
    """-> {kernel name: (summed ms, launches)} for the kernels launched since the last collect/enable."""
    lib = load()
    n = lib.sfgs_profile_kernel_count()
    ms = (C.c_double * n)()
    cnt = (C.c_int64 * n)()
    check(lib.sfgs_profile_collect(ms, cnt, n))
    return {lib.sfgs_profile_kernel_name(i).decode(): (ms[i], cnt[i]) for i in range(n) if cnt[i] > 0}
