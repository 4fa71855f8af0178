"""The storage-less handles the hooks hand to render() (sfgs.features.DeferredFeatures, sfgs.viewdirs.LazyDirs -- each of
its four kinds -- and sfgs.prepass.Deferred is covered in tests/test_prepass.py) must behave like the tensors they stand
for under ANY torch operation the reference, or a user's script, applies to a getter's result: a table of operations --
arithmetic, comparisons, indexing, views, reductions, dtype / device methods, functional and module calls, autograd --
is applied to the handle and to the real tensor and the results compared exactly."""
import pytest
import torch

from sfgs import features, viewdirs as vd

OPS = {
    "add_scalar": lambda t: t + 1.5,
    "radd": lambda t: 2.0 + t,
    "mul_tensor": lambda t: t * torch.arange(t.numel(), dtype=t.dtype, device=t.device).reshape(t.shape),
    "neg": lambda t: -t,
    "pow": lambda t: t ** 2,
    "matmul": lambda t: t.reshape(t.shape[0], -1) @ torch.ones(t.reshape(t.shape[0], -1).shape[1], 2, dtype=t.dtype, device=t.device),
    "gt": lambda t: t > 0.1,
    "index_int": lambda t: t[1],
    "index_slice": lambda t: t[1:4],
    "index_mask": lambda t: t[torch.arange(t.shape[0], device=t.device) % 2 == 0],
    "index_ellipsis": lambda t: t[..., 0],
    "reshape": lambda t: t.reshape(-1),
    "flatten": lambda t: t.flatten(1),
    "permute": lambda t: t.permute(*reversed(range(t.dim()))),
    "unsqueeze": lambda t: t.unsqueeze(0),
    "sum": lambda t: t.sum(),
    "mean_dim": lambda t: t.mean(dim=0),
    "max": lambda t: t.max(dim=-1).values,
    "norm_all": lambda t: t.norm(),
    "double": lambda t: t.double(),
    "half_roundtrip": lambda t: t.half().float(),
    "clone_detach": lambda t: t.clone().detach(),
    "cat": lambda t: torch.cat((t, t), dim=0),
    "stack": lambda t: torch.stack((t, t)),
    "where": lambda t: torch.where(t > 0, t, torch.zeros_like(t)),
    "sigmoid": lambda t: torch.sigmoid(t),
    "normalize": lambda t: torch.nn.functional.normalize(t.reshape(t.shape[0], -1)),
    "zeros_like": lambda t: torch.zeros_like(t),
    "isnan_any": lambda t: torch.isnan(t).any(),
    "tolist_len": lambda t: torch.tensor(len(t.tolist())),
    "numpy": lambda t: torch.from_numpy(t.detach().cpu().numpy().copy()),
    "expand_as": lambda t: t[:1].expand_as(t),
    "chunk": lambda t: t.chunk(2, dim=0)[0],
}


def _features(device="cpu"):
    gen = torch.Generator().manual_seed(0)
    dc = torch.randn(6, 1, 3, generator=gen).to(device).requires_grad_(True)
    rest = torch.randn(6, 3, 3, generator=gen).to(device).requires_grad_(True)
    return {"features": (lambda: features.DeferredFeatures(dc, rest), lambda: torch.cat((dc, rest), dim=1), (dc, rest)),
            "features_T": (lambda: features.DeferredFeatures(dc, rest).transpose(1, 2),
                           lambda: torch.cat((dc, rest), dim=1).transpose(1, 2), (dc, rest))}


def _dirs(device="cpu"):
    gen = torch.Generator().manual_seed(1)
    xyz = torch.nn.Parameter(torch.randn(6, 3, generator=gen).to(device))
    c = torch.randn(3, generator=gen).to(device).repeat(6, 1)
    h = lambda: vd.LazyDirs(vd.XYZ, xyz, tuple(xyz.shape), xyz)
    return {"xyz": (h, lambda: xyz, (xyz,)),
            "dir_pp": (lambda: h() - c, lambda: xyz - c, (xyz,)),
            "norm": (lambda: (h() - c).norm(dim=1, keepdim=True), lambda: (xyz - c).norm(dim=1, keepdim=True), (xyz,)),
            "dirs": (lambda: (lambda d: d / d.norm(dim=1, keepdim=True))(h() - c),
                     lambda: (lambda d: d / d.norm(dim=1, keepdim=True))(xyz - c), (xyz,))}


HANDLES = {**_features(), **_dirs()}
KINDS = sorted(HANDLES)


def check_operation(handles, kind, op):
    make_handle, make_real, leaves = handles[kind]
    h, r = make_handle(), make_real()
    assert isinstance(h, (features.DeferredFeatures, vd.LazyDirs)) and tuple(h.shape) == tuple(r.shape)
    assert h.dtype == r.dtype and h.device == r.device and h.requires_grad == r.requires_grad and h.dim() == r.dim()
    got, ref = OPS[op](h), OPS[op](r)
    assert type(got) is type(ref) or isinstance(got, torch.Tensor)
    assert not isinstance(got, (features.DeferredFeatures, vd.LazyDirs)), "an arbitrary operation returns a real tensor"
    assert got.shape == ref.shape and got.dtype == ref.dtype and got.device == ref.device
    assert torch.equal(got, ref)
    if got.requires_grad and got.is_floating_point():
        for t in leaves:
            t.grad = None
        got.sum().backward()
        g1 = [t.grad.clone() for t in leaves]
        for t in leaves:
            t.grad = None
        OPS[op](make_real()).sum().backward()
        for a, t in zip(g1, leaves):
            assert torch.equal(a, t.grad)


@pytest.mark.parametrize("op", sorted(OPS))
@pytest.mark.parametrize("kind", KINDS)
def test_any_operation_on_a_handle_equals_the_operation_on_the_tensor(kind, op):
    check_operation(HANDLES, kind, op)
