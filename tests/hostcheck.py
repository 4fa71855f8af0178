"""ctypes front-end of tests/host_check (CPU emulation of the HIP pipeline built from the product's
raster_math.h). Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "host_check", "host_check.cpp")
_HDR = os.path.join(_HERE, "..", "skyfall-gs_amd", "csrc", "raster_math.h")
_OUT = os.path.join(_HERE, "host_check", "_build", "libhostcheck.so")
_lib = None


class HcFrame(C.Structure):
    _fields_ = [("W", C.c_int32), ("H", C.c_int32), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
                ("kernel_size", C.c_float), ("scale_modifier", C.c_float), ("sh_degree", C.c_int32),
                ("sh_coeffs", C.c_int32), ("depth_mode", C.c_int32), ("subpix", C.c_void_p), ("bg", C.c_void_p),
                ("view", C.c_void_p), ("proj", C.c_void_p), ("campos", C.c_void_p)]


def lib():
    global _lib
    if _lib is None:
        newest = max(os.path.getmtime(_SRC), os.path.getmtime(_HDR))
        if not os.path.exists(_OUT) or os.path.getmtime(_OUT) < newest:
            os.makedirs(os.path.dirname(_OUT), exist_ok=True)
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall",
                                   "-o", _OUT, _SRC])
        _lib = C.CDLL(_OUT)
        _lib.hc_render.restype = C.c_int
        _lib.hc_render.argtypes = [C.POINTER(HcFrame), C.c_int32] + [C.c_void_p] * 22
        _lib.hc_cov3d.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
        _lib.hc_quat_to_R.argtypes = [C.c_void_p, C.c_void_p]
        _lib.hc_sh.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.hc_project.argtypes = [C.POINTER(HcFrame), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return _lib


def _f32(x):
    if x is None:
        return None
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(x, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def render(frame, means3D, scales, rotations, opacities, colors_precomp=None, shs=None, dL_dcolor=None,
           dL_ddepth=None, dL_dalpha=None, backward=False):
    L = lib()
    H, W = int(frame["H"]), int(frame["W"])
    N = int(means3D.shape[0])
    k = dict(m=_f32(means3D), s=_f32(scales), r=_f32(rotations), o=_f32(opacities).reshape(-1), c=_f32(colors_precomp),
             sh=_f32(shs), subpix=_f32(frame.get("subpix")), bg=_f32(frame["bg"]), view=_f32(frame["view"]),
             proj=_f32(frame["proj"]), campos=_f32(frame["campos"]))
    M = 0 if k["sh"] is None else int(k["sh"].shape[1])
    fr = HcFrame(W, H, float(frame["tanfovx"]), float(frame["tanfovy"]), float(frame["kernel_size"]),
                 float(frame.get("scale_modifier", 1.0)), int(frame.get("sh_degree", 0)), M,
                 int(frame.get("depth_mode", 0)), _p(k["subpix"]), _p(k["bg"]), _p(k["view"]), _p(k["proj"]),
                 _p(k["campos"]))
    out = dict(color=np.zeros((3, H, W), np.float32), depth=np.zeros((1, H, W), np.float32),
               alpha=np.zeros((1, H, W), np.float32), radii=np.zeros(N, np.int32), rec=np.zeros((N, 12), np.float32),
               counters=np.zeros(4, np.int64))
    g = {}
    if backward:
        g = dict(means3D=np.zeros((N, 3), np.float32), means2D=np.zeros((N, 3), np.float32),
                 scales=np.zeros((N, 3), np.float32), rotations=np.zeros((N, 4), np.float32),
                 opacities=np.zeros((N, 1), np.float32))
        if k["c"] is not None:
            g["colors_precomp"] = np.zeros((N, 3), np.float32)
        if k["sh"] is not None:
            g["shs"] = np.zeros((N, M, 3), np.float32)
    dc, dd, da = _f32(dL_dcolor), _f32(dL_ddepth), _f32(dL_dalpha)
    rc = L.hc_render(C.byref(fr), N, _p(k["m"]), _p(k["s"]), _p(k["r"]), _p(k["o"]), _p(k["c"]), _p(k["sh"]),
                     _p(out["color"]), _p(out["depth"]), _p(out["alpha"]), _p(out["radii"]), _p(out["rec"]), _p(dc),
                     _p(dd), _p(da), _p(g.get("means3D")), _p(g.get("means2D")), _p(g.get("scales")),
                     _p(g.get("rotations")), _p(g.get("opacities")), _p(g.get("colors_precomp")), _p(g.get("shs")),
                     _p(out["counters"]))
    assert rc == 0
    out["grads"] = g
    return out


def _frame_struct(frame, keep):
    k = dict(subpix=_f32(frame.get("subpix")), bg=_f32(frame["bg"]), view=_f32(frame["view"]),
             proj=_f32(frame["proj"]), campos=_f32(frame["campos"]))
    keep.append(k)
    return HcFrame(int(frame["W"]), int(frame["H"]), float(frame["tanfovx"]), float(frame["tanfovy"]),
                   float(frame["kernel_size"]), float(frame.get("scale_modifier", 1.0)), int(frame.get("sh_degree", 0)),
                   0, int(frame.get("depth_mode", 0)), _p(k["subpix"]), _p(k["bg"]), _p(k["view"]), _p(k["proj"]),
                   _p(k["campos"]))


def cov3d(scales, modifier, quats):
    s, q = _f32(scales), _f32(quats)
    out = np.zeros((s.shape[0], 6), np.float32)
    for i in range(s.shape[0]):
        lib().hc_cov3d(_p(s[i]), float(modifier), _p(q[i]), _p(out[i]))
    return out


def quat_to_R(quats):
    q = _f32(quats)
    out = np.zeros((q.shape[0], 9), np.float32)
    for i in range(q.shape[0]):
        lib().hc_quat_to_R(_p(q[i]), _p(out[i]))
    return out.reshape(-1, 3, 3)


def sh_rgb(deg, sh_km3, dirs):
    sh, d = _f32(sh_km3), _f32(dirs)
    out = np.zeros((sh.shape[0], 3), np.float32)
    for i in range(sh.shape[0]):
        lib().hc_sh(int(deg), int(sh.shape[1]), _p(sh[i]), _p(d[i]), _p(out[i]))
    return out


def project(frame, means3D, scales, quats):
    keep = []
    fr = _frame_struct(frame, keep)
    p, s, q = _f32(means3D), _f32(scales), _f32(quats)
    out = np.zeros((p.shape[0], 4), np.float32)
    for i in range(p.shape[0]):
        lib().hc_project(C.byref(fr), _p(p[i]), _p(s[i]), _p(q[i]), _p(out[i]))
    return out
