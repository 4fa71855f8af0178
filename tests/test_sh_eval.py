"""Fused eval_sh against golden vectors from the REAL utils/sh_utils.py::eval_sh + torch autograd
(tests/golden/make_golden.py::make_sh_grad_golden), degrees 0-3, K=16 stored coefficients."""
import os
import types

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_sh_grad.npz"))
RTOL, ATOL = 1e-5, 2e-6  # f32, sums of up to 16 products of O(1) terms; summation order differs from the reference's


def _basis_np(deg, d):
    """Independent numpy statement of the real SH basis used by the reference (constants of utils/sh_utils.py:26-55)."""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    B = [np.full_like(x, 0.28209479177387814)]
    if deg > 0:
        B += [-0.4886025119029199 * y, 0.4886025119029199 * z, -0.4886025119029199 * x]
    if deg > 1:
        B += [1.0925484305920792 * x * y, -1.0925484305920792 * y * z, 0.31539156525252005 * (2 * z * z - x * x - y * y),
              -1.0925484305920792 * x * z, 0.5462742152960396 * (x * x - y * y)]
    if deg > 2:
        B += [-0.5900435899266435 * y * (3 * x * x - y * y), 2.890611442640554 * x * y * z,
              -0.4570457994644658 * y * (4 * z * z - x * x - y * y),
              0.3731763325901154 * z * (2 * z * z - 3 * x * x - 3 * y * y),
              -0.4570457994644658 * x * (4 * z * z - x * x - y * y), 1.445305721320277 * z * (x * x - y * y),
              -0.5900435899266435 * x * (x * x - 3 * y * y)]
    return np.stack(B, axis=1)


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_golden_matches_closed_form_basis(deg):
    sh, d = G[f"shg{deg}_sh"].astype(np.float64), G[f"shg{deg}_dirs"].astype(np.float64)
    B = _basis_np(deg, d)
    want = np.einsum("nk,nck->nc", B, sh[:, :, :B.shape[1]])
    np.testing.assert_allclose(G[f"shg{deg}_out"], want, rtol=RTOL, atol=ATOL)
    g_sh = np.zeros_like(sh)
    g_sh[:, :, :B.shape[1]] = B[:, None, :] * G[f"shg{deg}_w"][:, :, None]
    np.testing.assert_allclose(G[f"shg{deg}_g_sh"], g_sh, rtol=RTOL, atol=ATOL)


def test_contract_errors_without_gpu():
    from sfgs.sh import eval_sh
    with pytest.raises(AssertionError):
        eval_sh(2, torch.zeros(4, 3, 4), torch.zeros(4, 3))   # 4 coefficients stored, degree 2 needs 9
    with pytest.raises(ValueError, match="no CPU fallback"):
        eval_sh(1, torch.zeros(4, 3, 4), torch.zeros(4, 3))


@pytest.mark.gpu
@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_fused_eval_sh_matches_reference(deg):
    from sfgs.sh import eval_sh
    dev = "cuda:0"
    sh = torch.tensor(G[f"shg{deg}_sh"], device=dev, requires_grad=True)
    dirs = torch.tensor(G[f"shg{deg}_dirs"], device=dev, requires_grad=True)
    out = eval_sh(deg, sh, dirs)
    (out * torch.tensor(G[f"shg{deg}_w"], device=dev)).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), G[f"shg{deg}_out"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(sh.grad.cpu().numpy(), G[f"shg{deg}_g_sh"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(dirs.grad.cpu().numpy(), G[f"shg{deg}_g_dirs"], rtol=RTOL, atol=5 * ATOL)


@pytest.mark.gpu
def test_fused_eval_sh_degree_4_matches_reference():
    """degree 4 exists only in the reference's Python eval_sh (utils/sh_utils.py:101-111); golden from the real function"""
    import os
    from sfgs.sh import eval_sh
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_sh4.npz"))
    dev = "cuda:0"
    sh = torch.tensor(z["sh"], device=dev, requires_grad=True)
    dirs = torch.tensor(z["dirs"], device=dev, requires_grad=True)
    out = eval_sh(4, sh, dirs)
    (out * torch.tensor(z["w"], device=dev)).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), z["out"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(sh.grad.cpu().numpy(), z["g_sh"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(dirs.grad.cpu().numpy(), z["g_dirs"], rtol=RTOL, atol=10 * ATOL)


@pytest.mark.gpu
def test_views_leading_dims_and_install():
    """render()'s `convert_SHs_python` path hands eval_sh a transposed (non-contiguous) view of get_features
    (gaussian_renderer/__init__.py:121-124); the appearance path a contiguous tensor with K > (active_deg+1)^2."""
    from sfgs import sh as fused
    dev = "cuda:0"
    feats = torch.tensor(G["shg1_sh"], device=dev).transpose(1, 2).contiguous()   # [N,K,3] like get_features
    view = feats.transpose(1, 2).view(-1, 3, 16)
    assert not view.is_contiguous()
    dirs = torch.tensor(G["shg1_dirs"], device=dev)
    np.testing.assert_allclose(fused.eval_sh(1, view, dirs).cpu().numpy(), G["shg1_out"], rtol=RTOL, atol=ATOL)
    out2 = fused.eval_sh(1, view.reshape(8, 12, 3, 16), dirs.reshape(8, 12, 3))
    assert out2.shape == (8, 12, 3)
    np.testing.assert_allclose(out2.reshape(-1, 3).cpu().numpy(), G["shg1_out"], rtol=RTOL, atol=ATOL)
    mod = types.ModuleType("gaussian_renderer_standin")
    mod.eval_sh = lambda *a: (_ for _ in ()).throw(AssertionError("not swapped"))
    fused.install(mod)
    try:
        assert mod.eval_sh is fused.eval_sh_deferred          # default: the folded route (DeferredColor handles)
        fused.install(mod, fold=False)
        assert mod.eval_sh is fused.eval_sh
    finally:
        fused.uninstall(mod)
    assert mod.eval_sh is not fused.eval_sh
    assert fused.eval_sh(3, torch.zeros(0, 3, 16, device=dev), torch.zeros(0, 3, device=dev)).shape == (0, 3)


def test_deferred_handle_records_exactly_renders_two_statements():
    """sfgs.sh.DeferredColor on the host (no kernel runs): `+ 0.5` and `clamp_min(0.0)` are RECORDED -- the statements of
    gaussian_renderer/__init__.py:116-117,124-125 --, the metadata render() / the rasterizer ask for is answered without
    materialising, and the coefficient layout is detected without a copy."""
    from sfgs.sh import DeferredColor
    a = torch.randn(10, 3, 4, requires_grad=True)
    d = torch.randn(10, 3)
    h = DeferredColor(1, a, d)
    assert h.shape == (10, 3) and h.dtype == torch.float32 and h.requires_grad and h.dim() == 2 and len(h) == 10
    assert h.folded_inputs() is None                                   # bare eval_sh: not render()'s expression
    for c in (torch.clamp_min(h + 0.5, 0.0), (h + 0.5).clamp_min(0.0), torch.clamp_min(torch.add(h, 0.5), min=0.0)):
        assert isinstance(c, DeferredColor) and c._sfgs_real is None
        deg, coef, dirs, cm = c.folded_inputs()
        assert deg == 1 and coef is a and dirs is d and cm is True
        assert c.float() is c
    assert torch.clamp_min(h + 0.25, 0.0).folded_inputs() is None      # other constants: recorded, never folded
    assert torch.clamp_min(h + 0.5, 0.1).folded_inputs() is None
    with torch.no_grad():
        assert not DeferredColor(1, a, d).requires_grad
    # convert_SHs_python hands eval_sh a transposed VIEW of the model's [N,K,3] features: recognised, no copy
    feats = torch.randn(10, 4, 3)
    view = feats.transpose(1, 2).view(-1, 3, 4)
    deg, coef, dirs, cm = torch.clamp_min(DeferredColor(1, view, d) + 0.5, 0.0).folded_inputs()
    assert cm is False and coef.shape == (10, 4, 3) and coef.data_ptr() == feats.data_ptr() and coef.is_contiguous()
    # degree 4 / 25 coefficients fold too (round 4); an unusual coefficient count stays with the stand-alone kernel
    assert torch.clamp_min(DeferredColor(4, torch.randn(10, 3, 25), d) + 0.5, 0.0).folded_inputs()[0] == 4
    assert torch.clamp_min(DeferredColor(1, torch.randn(10, 3, 5), d) + 0.5, 0.0).folded_inputs() is None
