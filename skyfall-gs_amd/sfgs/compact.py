"""One-pass pruning of every per-Gaussian tensor.

Reference: `GaussianModel.prune_points(mask)` -> `_prune_optimizer(valid_points_mask)` (scene/gaussian_model.py:
563-603) filters 7 parameters, their 14 Adam moments and 5 statistics tensors with 26 separate `tensor[mask]`
operations, each converting the mask to indices again and synchronising with the host. `prune_points` below does the
same surgery on the same objects (new `nn.Parameter`s, optimizer state re-keyed exactly like the reference) with one
mask scan, one host read-back (the surviving count) and one multi-tensor gather launch. `install(GaussianModel)` swaps
the method; `compact_rows` is the underlying utility."""
import torch
from torch import nn

from . import _lib as L

__all__ = ["compact_rows", "prune_points", "install", "uninstall"]

_SKIP_GROUPS = ("appearance_mlp", "appearance_embeddings")  # shared, not per-Gaussian (scene/gaussian_model.py:566)
_STATS = ("xyz_gradient_accum", "xyz_gradient_accum_abs", "xyz_gradient_accum_abs_max", "denom", "max_radii2D")


@torch.no_grad()
def compact_rows(keep, tensors):
    """[t[keep] for t in tensors] for a bool mask over dim 0, bit-exact, one scan + one gather for the whole list."""
    if keep.dtype == torch.bool:
        keep8 = keep.contiguous().view(torch.uint8)
    elif keep.dtype == torch.uint8:
        keep8 = keep.contiguous()
    else:
        raise ValueError("keep must be a bool (or uint8) mask")
    if keep8.dim() != 1 or not keep8.is_cuda:
        raise ValueError("keep must be a 1-D GPU mask (there is no CPU fallback)")
    n, dev = keep8.numel(), keep8.device
    srcs = []
    for t in tensors:
        if t.device != dev or t.dim() < 1 or t.shape[0] != n:
            raise ValueError(f"tensor of shape {tuple(t.shape)} on {t.device} does not match a mask of {n} rows on {dev}")
        srcs.append(t.detach().contiguous())
    lib = L.load()
    with torch.cuda.device(dev):
        stream = L.C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        scratch = torch.empty(max(lib.sfgs_compact_scratch_bytes(n), 1), dtype=torch.uint8, device=dev)
        kept = L.C.c_int64(0)
        L.check(lib.sfgs_compact_plan(L.ptr(keep8), n, L.ptr(scratch), scratch.numel(), L.C.byref(kept), stream))
        outs = [torch.empty((kept.value,) + tuple(s.shape[1:]), dtype=s.dtype, device=dev) for s in srcs]
        recs = [L.SfgsCompactTensor(s.data_ptr(), o.data_ptr(), (s.numel() // n if n else 0) * s.element_size())
                for s, o in zip(srcs, outs)]
        if recs and kept.value > 0:  # nothing survives -> empty outputs, nothing to copy
            arr = (L.SfgsCompactTensor * len(recs))(*recs)
            L.check(lib.sfgs_compact_rows(L.ptr(keep8), n, L.ptr(scratch), arr, len(recs), stream))
    return outs


@torch.no_grad()
def prune_points(model, mask):
    """Drop-in for GaussianModel.prune_points (mask: True = remove), scene/gaussian_model.py:586-603."""
    keep = ~mask
    opt = model.optimizer
    jobs = []  # (kind, group, key) in the order the outputs come back
    srcs = []
    for group in opt.param_groups:
        if group["name"] in _SKIP_GROUPS:
            continue
        p = group["params"][0]
        st = opt.state.get(p, None)
        srcs.append(p); jobs.append(("param", group, None))
        if st is not None:
            srcs.append(st["exp_avg"]); jobs.append(("state", group, "exp_avg"))
            srcs.append(st["exp_avg_sq"]); jobs.append(("state", group, "exp_avg_sq"))
    stats = [s for s in _STATS if isinstance(getattr(model, s, None), torch.Tensor)
             and getattr(model, s).dim() >= 1 and getattr(model, s).shape[0] == keep.shape[0]]
    for s in stats:
        srcs.append(getattr(model, s)); jobs.append(("stat", None, s))
    outs = compact_rows(keep, srcs)

    optimizable = {}
    new_state = {}
    for (kind, group, key), out in zip(jobs, outs):
        if kind == "param":
            old = group["params"][0]
            st = opt.state.get(old, None)
            if st is not None:
                del opt.state[old]
            group["params"][0] = nn.Parameter(out.requires_grad_(True))
            if st is not None:
                opt.state[group["params"][0]] = st
                new_state[id(group)] = st
            optimizable[group["name"]] = group["params"][0]
        elif kind == "state":
            new_state[id(group)][key] = out
        else:
            setattr(model, key, out)
    model._xyz = optimizable["xyz"]
    model._features_dc = optimizable["f_dc"]
    model._features_rest = optimizable["f_rest"]
    model._opacity = optimizable["opacity"]
    model._scaling = optimizable["scaling"]
    model._rotation = optimizable["rotation"]
    if getattr(model, "appearance_enabled", False):
        model._embeddings = optimizable["embeddings"]


_ORIG = {}


def install(gaussian_model_cls):
    if gaussian_model_cls in _ORIG:
        return
    _ORIG[gaussian_model_cls] = gaussian_model_cls.prune_points
    gaussian_model_cls.prune_points = prune_points


def uninstall(gaussian_model_cls):
    if gaussian_model_cls in _ORIG:
        gaussian_model_cls.prune_points = _ORIG.pop(gaussian_model_cls)
