"""Fused multi-tensor Adam for the Gaussian parameter groups.

Reference: `scene/gaussian_model.py:357-382` builds `torch.optim.Adam(l, lr=0.0, eps=1e-15)` over one group per
tensor (xyz, f_dc, f_rest, opacity, scaling, rotation [, appearance_embeddings (weight_decay), embeddings,
appearance_mlp]) and `train.py:339-340,906-907` steps it every iteration. torch's default CUDA path runs ~8
elementwise passes over every tensor; `FusedAdam.step()` does the whole update of every group in ONE HIP launch
(`sfgs_adam_step`, csrc/adam.hip) with the same arithmetic.

`FusedAdam` IS a `torch.optim.Adam` (subclass): `param_groups`, `state[p] = {"step", "exp_avg", "exp_avg_sq"}`,
`state_dict()/load_state_dict()`, `zero_grad()` are inherited, so the reference's optimizer surgery
(`replace_tensor_to_optimizer`, `_prune_optimizer`, `cat_tensors_to_optimizer`, `scene/gaussian_model.py:549-624`)
and checkpoint capture/restore (`:163-201`) work unchanged. `install(GaussianModel)` swaps the optimizer in right
after the reference's own `training_setup` built it."""
import torch

from . import _lib as L

__all__ = ["FusedAdam", "install", "uninstall"]


def _dense(t):
    """True when t's elements occupy exactly numel() consecutive storage slots (any permutation of a contiguous layout)."""
    if t.is_contiguous():
        return True
    expect = 1
    for size, stride in sorted(((sz, sd) for sz, sd in zip(t.shape, t.stride()) if sz != 1), key=lambda x: x[1]):
        if stride != expect:
            return False
        expect *= size
    return True


class FusedAdam(torch.optim.Adam):
    """Drop-in for `torch.optim.Adam(params, lr, betas, eps, weight_decay)` on float32 / float64 GPU parameters."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if isinstance(lr, torch.Tensor):
            raise NotImplementedError("FusedAdam: tensor learning rates are not supported")
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        L.load()  # fail at construction, not at the first step, if the HIP library is missing

    @classmethod
    def from_adam(cls, opt):
        """Re-home an existing torch.optim.Adam (same parameter objects, groups, hyper-parameters and state)."""
        if not isinstance(opt, torch.optim.Adam):
            raise TypeError("from_adam expects a torch.optim.Adam")
        d = opt.defaults
        for k in ("amsgrad", "maximize", "capturable", "differentiable"):
            if d.get(k) or any(g.get(k) for g in opt.param_groups):
                raise NotImplementedError(f"FusedAdam does not implement {k}=True")
        groups = [dict(g) for g in opt.param_groups]
        new = cls(groups, lr=d["lr"], betas=d["betas"], eps=d["eps"], weight_decay=d["weight_decay"])
        for p, st in opt.state.items():
            new.state[p] = st
        return new

    @torch.no_grad()
    def step(self, closure=None):  # noqa: C901
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        per_device = {}
        for group in self.param_groups:
            for k in ("amsgrad", "maximize", "capturable", "differentiable", "decoupled_weight_decay"):
                if group.get(k):
                    raise NotImplementedError(f"FusedAdam does not implement {k}=True")
            lr, (beta1, beta2), eps, wd = group["lr"], group["betas"], group["eps"], group["weight_decay"]
            if isinstance(lr, torch.Tensor) or isinstance(beta1, torch.Tensor) or isinstance(beta2, torch.Tensor):
                raise NotImplementedError("FusedAdam: tensor hyper-parameters are not supported")
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue  # torch.optim.Adam skips parameters without a gradient (no step increment either)
                if g.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
                # float64: the reference's `_opacity` group from its first reset_opacity on (scene/gaussian_model.py:483-501)
                # Layout: any DENSE, non-overlapping layout (the update is elementwise, so parameter, gradient and moments
                # only have to share ONE index -> offset map): the reference's `_xyz` is column-major from create_from_pcd
                # (fetchPly's `np.vstack([x, y, z]).T`, scene/dataset_readers.py:126-132 -> torch.tensor keeps the strides)
                # until the first densification re-allocates it -- found by running the real train.training() (round 6).
                if not p.is_cuda or p.dtype not in (torch.float32, torch.float64) or not _dense(p):
                    raise ValueError("FusedAdam: parameters must be dense (non-overlapping) float32 / float64 GPU tensors "
                                     f"(got {p.dtype}, {p.device}, shape {tuple(p.shape)}, strides {p.stride()})")
                if g.dtype != p.dtype or g.device != p.device or g.shape != p.shape:
                    raise ValueError("FusedAdam: gradient dtype/device/shape must match its parameter")
                st = self.state[p]
                if len(st) == 0:  # same lazy state as torch/optim/adam.py::_init_group
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                m, v = st["exp_avg"], st["exp_avg_sq"]
                for name, t in (("exp_avg", m), ("exp_avg_sq", v)):
                    if t.dtype != p.dtype or t.device != p.device or t.shape != p.shape:
                        raise ValueError(f"FusedAdam: state '{name}' does not match its parameter "
                                         f"({tuple(t.shape)} {t.dtype} {t.device} vs {tuple(p.shape)})")
                    if t.stride() != p.stride():     # e.g. a state dict loaded into a differently laid out parameter
                        st[name] = t = torch.empty_like(p, memory_format=torch.preserve_format).copy_(t)
                        if name == "exp_avg":
                            m = t
                        else:
                            v = t
                if g.stride() != p.stride():
                    g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
                st["step"] += 1
                t_step = float(st["step"])
                # host scalars in double, exactly as torch/optim/adam.py::_multi_tensor_adam forms them
                bc1 = 1 - beta1 ** t_step
                bc2 = 1 - beta2 ** t_step
                rec = L.SfgsAdamTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
                                       (lr / bc1) * -1, 1 - beta1, beta2, 1 - beta2, bc2 ** 0.5, eps, wd,
                                       L.ADAM_F64 if p.dtype == torch.float64 else 0, 0)
                per_device.setdefault(p.device, ([], [], []))
                per_device[p.device][0].append(rec)
                per_device[p.device][1].append(g)  # keep contiguous copies alive until the launch is enqueued
                per_device[p.device][2].extend((p, m, v))
        for dev, (recs, _keep, mutated) in per_device.items():
            arr = (L.SfgsAdamTensor * len(recs))(*recs)
            with torch.cuda.device(dev):
                stream = L.C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                L.check(L.load().sfgs_adam_step(arr, len(recs), stream))
            # the kernel wrote through raw pointers: tell autograd (and anything keyed on tensor versions, e.g.
            # sfgs.prepass's per-iteration cache) that these tensors changed in place, as torch's own step() does
            for t in mutated:
                torch.autograd.graph.increment_version(t)
        return loss


_ORIG = {}


def install(gaussian_model_cls):
    """Wrap `GaussianModel.training_setup` (scene/gaussian_model.py:350) so the optimizer it builds is re-homed into
    a FusedAdam; everything else the reference does with `self.optimizer` keeps working on the subclass."""
    if gaussian_model_cls in _ORIG:
        return
    orig = gaussian_model_cls.training_setup
    _ORIG[gaussian_model_cls] = orig

    def training_setup(self, *args, **kwargs):
        out = orig(self, *args, **kwargs)
        self.optimizer = FusedAdam.from_adam(self.optimizer)
        return out

    training_setup.__wrapped__ = orig
    gaussian_model_cls.training_setup = training_setup


def uninstall(gaussian_model_cls):
    if gaussian_model_cls in _ORIG:
        gaussian_model_cls.training_setup = _ORIG.pop(gaussian_model_cls)
