"""Shared comparison helpers for the parity tests (tolerances of SURVEY A.7 / BASELINE north_star)."""
import numpy as np

RGB_DEPTH_RTOL = 1e-4      # relative L-inf (normalised by max |ref| of the image), float32
BORDERLINE_FRAC = 1e-4     # <= 0.01 % of pixels may differ more: a splat within an ulp of the 1/255 or
                           # T < 1e-4 thresholds flips between exp implementations
GRAD_RTOL_L2 = 1e-3        # relative L2 per gradient tensor
GRAD_RTOL_MAX = 2e-3       # relative L-inf (normalised by max |ref|)


def image_report(name, got, ref, rtol=RGB_DEPTH_RTOL):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    nan_g, nan_r = np.isnan(got), np.isnan(ref)
    assert (nan_g == nan_r).all(), f"{name}: NaN pattern differs at {(nan_g != nan_r).sum()} pixels"
    ok = ~nan_r
    scale = max(np.abs(ref[ok]).max(), 1e-30) if ok.any() else 1.0
    err = np.zeros_like(ref)
    err[ok] = np.abs(got[ok] - ref[ok]) / scale
    bad = err > rtol
    return dict(name=name, max_rel=float(err.max()), bad=int(bad.sum()), total=int(err.size))


def assert_image_close(name, got, ref, rtol=RGB_DEPTH_RTOL):
    r = image_report(name, got, ref, rtol)
    assert r["bad"] <= max(1, int(BORDERLINE_FRAC * r["total"])) or r["max_rel"] <= rtol, r
    assert r["max_rel"] < 5e-2, r   # a flipped borderline splat moves a pixel by <= alpha*T*|c| ~ 1/255
    return r


def grad_report(name, got, ref):
    got = np.asarray(got, np.float64).reshape(-1)
    ref = np.asarray(ref, np.float64).reshape(-1)
    l2 = np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30)
    mx = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)
    return dict(name=name, rel_l2=float(l2), rel_max=float(mx))


def assert_grad_close(name, got, ref, l2=GRAD_RTOL_L2, mx=GRAD_RTOL_MAX):
    r = grad_report(name, got, ref)
    assert np.isfinite(np.asarray(got)).all(), f"{name}: non-finite gradient"
    assert r["rel_l2"] <= l2 and r["rel_max"] <= mx, r
    return r
