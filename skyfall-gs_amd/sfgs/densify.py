"""GaussianModel.densify_and_prune as a handful of HIP launches (the densification half).

Reference: scene/gaussian_model.py -- densify_and_prune :707-742, densify_and_clone :686-705, densify_and_split :653-684,
densification_postfix / cat_tensors_to_optimizer :603-651, prune_points :586-601; called every densification_interval
iterations from train.py:318-322. The reference pushes every parameter, both Adam moments of every parameter and the
statistics tensors through boolean indexing, repeat, cat and boolean indexing again (several dozen kernels and a host
sync per mask), and its quantile is a full sort that torch refuses above 16 M elements (it then falls back to Q = 0.99).

`densify_and_prune(model, max_grad, min_opacity, extent, max_screen_size)` below performs the same surgery on the same
objects -- new nn.Parameters, optimizer state re-keyed like the reference, statistics reset -- with: a few N-sized
elementwise torch ops for the normalised gradients (kept in torch so that the thresholds are compared with exactly the
reference's floats), an exact radix-select quantile for any N, ONE decision + scan pass, ONE host read-back (the output
size), ONE multi-tensor gather and ONE kernel for the computed child rows. The decisions are bit-exact with the
reference's (tests/test_densify_masks.py); the final row order is the reference's. `install(GaussianModel)` swaps the
method in."""
import torch
from torch import nn

from . import _lib as L

__all__ = ["quantile_linear", "decide_masks", "densify_and_prune", "zcurve_permutation", "reorder_zcurve", "install",
           "uninstall"]

_SKIP_GROUPS = ("appearance_mlp", "appearance_embeddings")   # shared, not per-Gaussian (scene/gaussian_model.py:607)


def _stream(dev):
    return L.C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


@torch.no_grad()
def quantile_linear(values, q):
    """torch.quantile(values, q) (linear interpolation) for a 1-D non-negative float32 GPU tensor of ANY length, without
    sorting: rank = q * (n - 1) as torch forms it, the two neighbouring order statistics by radix select, torch.lerp.
    q: 0-dim tensor or float. Returns a 0-dim device tensor; no host synchronisation."""
    if values.dtype != torch.float32 or values.dim() != 1 or not values.is_cuda or values.numel() == 0:
        raise ValueError("quantile_linear expects a non-empty 1-D float32 GPU tensor")
    lib = L.load()
    dev = values.device
    v = values.contiguous()
    n = v.numel()
    q = torch.as_tensor(q, device=dev)
    if not q.is_floating_point():
        q = q.float()
    rank = (q * (n - 1)).to(torch.float32)          # aten/native/Sorting.cpp: ranks = q * (size - 1)
    out2 = torch.empty(2, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        scratch = torch.empty(lib.sfgs_select_scratch_bytes(), dtype=torch.uint8, device=dev)
        rk = rank.reshape(1).contiguous()
        L.check(lib.sfgs_select_kth(L.ptr(v), n, L.ptr(rk), L.ptr(out2), L.ptr(scratch), scratch.numel(), _stream(dev)))
    w = (rank - rank.floor()).to(torch.float32)
    return torch.lerp(out2[0], out2[1], w)


def _decide(model, max_grad, min_opacity, extent, max_screen_size):
    """Stage 1: gradients, Q, per-Gaussian decisions + scan. Returns (scratch, totals, scaling, N)."""
    lib = L.load()
    dev = model._xyz.device
    N = int(model._xyz.shape[0])
    # ---- normalised gradients and the abs-gradient threshold Q, as the reference forms them (:708-723) ----------------
    grads = model.xyz_gradient_accum / model.denom
    grads[grads.isnan()] = 0.0
    grads_abs = model.xyz_gradient_accum_abs / model.denom
    grads_abs[grads_abs.isnan()] = 0.0
    gnorm = torch.norm(grads, dim=-1).contiguous()
    gabs = torch.norm(grads_abs, dim=-1).contiguous()
    Q_dev, Q_host = None, 0.99
    if N > 0:
        # the reference's validity test (:714) would need a host answer; it stays on the device: a non-finite grads_abs
        # selects Q = 0.99 there
        ratio = (gnorm >= max_grad).float().mean()
        finite = torch.isfinite(grads_abs).all()
        Qsel = quantile_linear(torch.nan_to_num(gabs, nan=0.0, posinf=0.0, neginf=0.0), 1 - ratio)
        Q_dev = torch.where(finite, Qsel, torch.full_like(Qsel, 0.99)).reshape(1).contiguous()
    scaling = model.get_scaling.detach().contiguous()            # activated, as the reference reads them
    opacity = model.get_opacity.detach().reshape(-1).contiguous()
    if scaling.dtype != torch.float32 or opacity.dtype not in (torch.float32, torch.float64):
        raise ValueError("densify_and_prune expects float32 scaling and float32 / float64 opacity")
    dense_thr = float(torch.tensor(model.percent_dense * extent, dtype=torch.float32))   # compared in float32 (:691,:663)
    big_thr = float(torch.tensor(0.1 * extent, dtype=torch.float32))                     # :733
    totals = (L.C.c_int64 * 5)()
    with torch.cuda.device(dev):
        scratch = torch.empty(max(lib.sfgs_densify_scratch_bytes(N), 1), dtype=torch.uint8, device=dev)
        L.check(lib.sfgs_densify_decide(N, L.ptr(gnorm), L.ptr(gabs), L.ptr(scaling), L.ptr(opacity),
                                        int(opacity.dtype == torch.float64), L.ptr(Q_dev), Q_host, float(max_grad),
                                        float(min_opacity), dense_thr, big_thr, int(bool(max_screen_size)),
                                        L.ptr(scratch), scratch.numel(), totals, _stream(dev)))
    return scratch, totals, scaling, N, Q_dev


@torch.no_grad()
def decide_masks(model, max_grad, min_opacity, extent, max_screen_size):
    """The decisions only (no surgery): (clone[N], split[N], keep[N,3] = rows of {original, clone, children} that survive
    the final prune, Q). Diagnostics / tests."""
    lib = L.load()
    scratch, totals, _, N, Q_dev = _decide(model, max_grad, min_opacity, extent, max_screen_size)
    dev = model._xyz.device
    clone = torch.empty(N, dtype=torch.uint8, device=dev)
    split = torch.empty(N, dtype=torch.uint8, device=dev)
    keep = torch.empty(N, 3, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        L.check(lib.sfgs_densify_masks(N, L.ptr(scratch), L.ptr(clone), L.ptr(split), L.ptr(keep), _stream(dev)))
    return clone.bool(), split.bool(), keep.bool(), Q_dev


@torch.no_grad()
def densify_and_prune(model, max_grad, min_opacity, extent, max_screen_size, samples=None):
    """Drop-in for GaussianModel.densify_and_prune; returns the same (n_cloned, n_split, n_pruned) triple.
    samples (tests): [2 * n_split, 3] = std * z in the layout of the reference's `samples`; default: drawn here."""
    lib = L.load()
    dev = model._xyz.device
    scratch, totals, scaling, N, _ = _decide(model, max_grad, min_opacity, extent, max_screen_size)
    with torch.cuda.device(dev):
        stream = _stream(dev)
        n_orig, n_clone, n_child, raw_clone, raw_split = (int(v) for v in totals)
        new_n = n_orig + n_clone + 2 * n_child
        # ---- one gather for every per-Gaussian tensor and its Adam moments ----------------------------------------------
        opt = model.optimizer
        jobs, recs, keep_alive = [], [], []

        def add(src, zero_new, job):
            src = src.detach().contiguous()
            row = (src.numel() // N if N else 0) * src.element_size()
            dst = torch.empty((new_n,) + tuple(src.shape[1:]), dtype=src.dtype, device=dev)
            recs.append(L.SfgsDensifyTensor(src.data_ptr(), dst.data_ptr(), row, int(zero_new), 0))
            keep_alive.append(src)
            jobs.append((job, dst))
            return dst
        for group in opt.param_groups:
            if group["name"] in _SKIP_GROUPS:
                continue
            p = group["params"][0]
            st = opt.state.get(p, None)
            add(p, False, ("param", group))
            if st is not None:
                add(st["exp_avg"], True, ("state", group, "exp_avg"))
                add(st["exp_avg_sq"], True, ("state", group, "exp_avg_sq"))
        if recs and N > 0 and new_n > 0:
            arr = (L.SfgsDensifyTensor * len(recs))(*recs)
            L.check(lib.sfgs_densify_gather(N, L.ptr(scratch), totals, arr, len(recs), stream))
        outs = {}
        for job, dst in jobs:
            if job[0] == "param":
                outs[job[1]["name"]] = dst
        # ---- the computed child rows: xyz = parent + R(q) * sample, raw scaling = log(scaling / 1.6) (:666-672) -------------
        if n_child > 0:
            unit = samples is None
            if unit:   # z ~ N(0, 1); the kernel applies the parent's scaling (torch.normal(0, std) = std * z): no mask gather
                samples = torch.randn(2 * raw_split, 3, device=dev)
            samples = samples.to(dev, torch.float32).contiguous()
            if tuple(samples.shape) != (2 * raw_split, 3):
                raise ValueError(f"samples must have shape ({2 * raw_split}, 3)")
            L.check(lib.sfgs_densify_children(N, L.ptr(scratch), totals, L.ptr(model._xyz.detach().contiguous()),
                                              L.ptr(model._rotation.detach().contiguous()), L.ptr(scaling), L.ptr(samples),
                                              int(unit), L.ptr(outs["xyz"]), L.ptr(outs["scaling"]), stream))
    # ---- optimizer surgery exactly like cat_tensors_to_optimizer / _prune_optimizer (:564-624) ----------------------------
    optimizable, new_state = {}, {}
    for job, dst in jobs:
        if job[0] == "param":
            group = job[1]
            old = group["params"][0]
            st = opt.state.get(old, None)
            if st is not None:
                del opt.state[old]
            group["params"][0] = nn.Parameter(dst.requires_grad_(True))
            if st is not None:
                opt.state[group["params"][0]] = st
                new_state[id(group)] = st
            optimizable[group["name"]] = group["params"][0]
        else:
            new_state[id(job[1])][job[2]] = dst
    model._xyz = optimizable["xyz"]
    model._features_dc = optimizable["f_dc"]
    model._features_rest = optimizable["f_rest"]
    model._opacity = optimizable["opacity"]
    model._scaling = optimizable["scaling"]
    model._rotation = optimizable["rotation"]
    if getattr(model, "appearance_enabled", False) and "embeddings" in optimizable:
        model._embeddings = optimizable["embeddings"]
    # densification_postfix (:646-651) reset the statistics before the prunes filtered them: all zeros of the new size
    model.xyz_gradient_accum = torch.zeros((new_n, 1), device=dev)
    model.xyz_gradient_accum_abs = torch.zeros((new_n, 1), device=dev)
    model.xyz_gradient_accum_abs_max = torch.zeros((new_n, 1), device=dev)
    model.denom = torch.zeros((new_n, 1), device=dev)
    model.max_radii2D = torch.zeros((new_n,), device=dev)
    n_before_prune = N + raw_clone + raw_split          # clones appended, parents replaced by two children each
    n_pruned = n_before_prune - new_n
    print(f"Pruning {n_pruned} points out of {n_before_prune} points")   # the reference's message (:737)
    return raw_clone, raw_split, n_pruned


# ---- optional: keep the Gaussians in a spatially coherent storage order ---------------------------------------------------
def zcurve_permutation(xyz, bits=10):
    """Permutation that sorts points along a 3-D Z-curve (Morton order, 2^bits cells per axis of the bounding box)."""
    x = xyz.detach().double()
    lo, hi = x.min(0).values, x.max(0).values
    q = ((x - lo) / (hi - lo).clamp_min(1e-30) * (2 ** bits - 1)).long().clamp_(0, 2 ** bits - 1)
    key = torch.zeros(x.shape[0], dtype=torch.int64, device=x.device)
    for b in range(bits):
        for a in range(3):
            key |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return torch.argsort(key, stable=True)


_PER_GAUSSIAN_BUFFERS = ("xyz_gradient_accum", "xyz_gradient_accum_abs", "xyz_gradient_accum_abs_max", "denom",
                         "max_radii2D", "filter_3D")


@torch.no_grad()
def reorder_zcurve(model, bits=10):
    """Permute every per-Gaussian tensor of the model -- parameters, their Adam moments, the densification statistics,
    filter_3D -- along a Z-curve of the positions. A pure relabelling: the scene, the optimizer trajectory and the
    rendered images are unchanged (the rasterizer breaks exact depth ties by index, nothing else depends on the order),
    but the waves of the binning kernel then append to the SAME coarse bin (one merged atomic, full-line slab stores:
    `preprocess` 0.23 -> 0.17 ms at 2 M Gaussians) and tiles gather neighbouring records. The
    reference's own order -- a gridded point cloud, children appended in parent order -- is partly coherent already;
    install(..., zcurve_order=True) restores it at every densification. Returns the permutation.
    filter_3D is permuted only while it has one row per Gaussian; right after a densification it still has the OLD row
    count and is left alone -- call compute_3D_filter afterwards, as train.py:330 does after every densify_and_prune."""
    perm = zcurve_permutation(model._xyz, bits)
    n = perm.numel()
    opt = model.optimizer
    renamed = {}
    for group in opt.param_groups:
        if group["name"] in _SKIP_GROUPS or group["params"][0].shape[0] != n:
            continue
        old = group["params"][0]
        st = opt.state.pop(old, None)
        new = nn.Parameter(old.detach()[perm].contiguous().requires_grad_(True))
        group["params"][0] = new
        if st is not None:
            for k in ("exp_avg", "exp_avg_sq"):
                if k in st:
                    st[k] = st[k][perm].contiguous()
            opt.state[new] = st
        renamed[group["name"]] = new
    for attr, name in (("_xyz", "xyz"), ("_features_dc", "f_dc"), ("_features_rest", "f_rest"), ("_opacity", "opacity"),
                       ("_scaling", "scaling"), ("_rotation", "rotation"), ("_embeddings", "embeddings")):
        if name in renamed and hasattr(model, attr):
            setattr(model, attr, renamed[name])
    for attr in _PER_GAUSSIAN_BUFFERS:
        t = getattr(model, attr, None)
        if isinstance(t, torch.Tensor) and t.dim() >= 1 and t.shape[0] == n:
            setattr(model, attr, t[perm].contiguous())
    return perm


_ORIG = {}
_VARIANT = {}


def install(gaussian_model_cls, zcurve_order=False):
    """zcurve_order=True: every densify_and_prune is followed by reorder_zcurve (row order then differs from the
    reference's [survivors | clones | children]; everything else is identical)."""
    if gaussian_model_cls in _ORIG:
        if _VARIANT.get(gaussian_model_cls) == bool(zcurve_order):
            return
        uninstall(gaussian_model_cls)       # a different variant was installed: re-patch instead of ignoring the request
    _ORIG[gaussian_model_cls] = gaussian_model_cls.densify_and_prune
    _VARIANT[gaussian_model_cls] = bool(zcurve_order)
    if zcurve_order:
        def densify_and_prune_zcurve(model, *args, **kwargs):
            out = densify_and_prune(model, *args, **kwargs)
            reorder_zcurve(model)
            return out
        gaussian_model_cls.densify_and_prune = densify_and_prune_zcurve
    else:
        gaussian_model_cls.densify_and_prune = densify_and_prune


def uninstall(gaussian_model_cls):
    if gaussian_model_cls in _ORIG:
        gaussian_model_cls.densify_and_prune = _ORIG.pop(gaussian_model_cls)
        _VARIANT.pop(gaussian_model_cls, None)
