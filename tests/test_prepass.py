"""Fused pre-pass: oracle pinned to golden vectors from the reference's REAL GaussianModel
getters (CPU), and the HIP op against both (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle.prepass_torch import prepass_reference

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_helpers.npz"))


def _inputs(tag, device="cpu", grad=True):
    t = lambda k: torch.tensor(G[f"prepass_{tag}_{k}"], device=device)
    return (t("scaling").requires_grad_(grad), t("opacity").requires_grad_(grad), t("rotation").requires_grad_(grad),
            t("filter"))


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_oracle_matches_real_gaussian_model(tag):
    a, b, c, f = _inputs(tag)
    assert f.dtype == (torch.float64 if tag == "f64" else torch.float32)
    sc, op, ro = prepass_reference(a, b, c, f)
    np.testing.assert_array_equal(sc.detach().numpy(), G[f"prepass_{tag}_out_scales"])
    np.testing.assert_array_equal(op.detach().numpy(), G[f"prepass_{tag}_out_opacity"])
    np.testing.assert_array_equal(ro.detach().numpy(), G[f"prepass_{tag}_out_rotation"])
    w = lambda k: torch.tensor(G[f"prepass_{tag}_w_{k}"])
    ((sc * w("scales")).sum() + (op * w("opacity")).sum() + (ro * w("rotation")).sum()).backward()
    for k, p in (("scaling", a), ("opacity", b), ("rotation", c)):
        ref = G[f"prepass_{tag}_g_{k}"]
        assert np.abs(p.grad.numpy() - ref).max() <= 1e-6 * np.abs(ref).max(), k


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_fused_op_matches_golden(tag):
    from sfgs.prepass import fused_activations
    a, b, c, f = _inputs(tag, "cuda:0")
    sc, op, ro = fused_activations(a, b, c, f)
    for got, key in ((sc, "out_scales"), (op, "out_opacity"), (ro, "out_rotation")):
        ref = G[f"prepass_{tag}_{key}"]
        np.testing.assert_allclose(got.detach().cpu().numpy(), ref, rtol=3e-7, atol=1e-12)  # <= 2 float32 ulp
    w = lambda k: torch.tensor(G[f"prepass_{tag}_w_{k}"], device="cuda:0")
    ((sc * w("scales")).sum() + (op * w("opacity")).sum() + (ro * w("rotation")).sum()).backward()
    for k, p in (("scaling", a), ("opacity", b), ("rotation", c)):
        ref = G[f"prepass_{tag}_g_{k}"]
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 2e-6 * np.abs(ref).max(), k


@pytest.mark.gpu
def test_install_patches_a_gaussian_model_shaped_class():
    """The three getters of a class shaped like the reference's GaussianModel are replaced by one fused launch;
    render()-style reads see identical values and gradients flow to the raw parameters."""
    from sfgs import prepass

    class GaussianModel:  # same attribute / property names as scene/gaussian_model.py
        def __init__(self, n):
            g = torch.Generator().manual_seed(0)
            dev = "cuda:0"
            self._scaling = (torch.randn(n, 3, generator=g) - 2).to(dev).requires_grad_(True)
            self._opacity = torch.randn(n, 1, generator=g).to(dev).requires_grad_(True)
            self._rotation = torch.randn(n, 4, generator=g).to(dev).requires_grad_(True)
            self.filter_3D = torch.exp(torch.randn(n, 1, generator=g, dtype=torch.float64) - 3).to(dev)

        @property
        def get_scaling_with_3D_filter(self):
            return prepass_reference(self._scaling, self._opacity, self._rotation, self.filter_3D)[0]

        @property
        def get_opacity_with_3D_filter(self):
            return prepass_reference(self._scaling, self._opacity, self._rotation, self.filter_3D)[1]

        @property
        def get_rotation(self):
            return prepass_reference(self._scaling, self._opacity, self._rotation, self.filter_3D)[2]

    m = GaussianModel(100_000)
    ref = [m.get_scaling_with_3D_filter, m.get_opacity_with_3D_filter, m.get_rotation]
    (ref[0].sum() + 2 * ref[1].sum() + (ref[2] * ref[2][:, :1]).sum()).backward()
    gref = [p.grad.clone() for p in (m._scaling, m._opacity, m._rotation)]
    for p in (m._scaling, m._opacity, m._rotation):
        p.grad = None
    prepass.install(GaussianModel)
    try:
        got = [m.get_scaling_with_3D_filter, m.get_opacity_with_3D_filter, m.get_rotation]
        assert m._sfgs_prepass_cache[1][0] is got[0]  # one launch shared by the three getters
        for a, b in zip(got, ref):
            assert torch.allclose(a, b.float(), rtol=3e-7, atol=1e-12)
        (got[0].sum() + 2 * got[1].sum() + (got[2] * got[2][:, :1]).sum()).backward()
        for p, g in zip((m._scaling, m._opacity, m._rotation), gref):
            assert float((p.grad - g).abs().max()) <= 3e-6 * float(g.abs().max())
        with torch.no_grad():  # optimiser step: parameter version changes -> recomputed
            m._scaling.add_(0.1)
        assert not torch.equal(m.get_scaling_with_3D_filter, got[0])
    finally:
        prepass.uninstall(GaussianModel)
    assert isinstance(GaussianModel.__dict__["get_rotation"], property) and GaussianModel not in prepass._ORIG
