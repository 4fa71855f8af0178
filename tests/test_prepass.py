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


def test_deferred_handles_materialise_for_everyone_but_the_rasterizer(monkeypatch):
    """sfgs.prepass.Deferred (what the patched getters return with fold=True) on a GPU-less host: metadata and render()'s
    `.float()` leave it alone, any other use turns it into the ordinary tensors of ONE launch with their autograd graph,
    and a rasterizer backend without a raw-parameter route (here: the oracle double) gets materialised values."""
    from sfgs import prepass
    a, b, c, f = _inputs("f64")
    launches = []

    def fake(sa, sb, sc_, sf, _state=None):   # the real op needs the GPU library; same contract
        launches.append(1)
        return tuple(t.float() for t in prepass_reference(sa, sb, sc_, sf))
    monkeypatch.setattr(prepass, "fused_activations", fake)
    n = a.shape[0]
    shared = prepass._Shared((a, b, c, f))
    sc, op, ro = (prepass.Deferred(shared, i, s) for i, s in enumerate(((n, 3), (n, 1), (n, 4))))
    assert isinstance(sc, torch.Tensor) and tuple(sc.shape) == (n, 3) and sc.dtype == torch.float32 and not sc.is_cuda
    assert sc.float() is sc and op.numel() == n and ro.dim() == 2 and len(ro) == n and sc.size(1) == 3
    assert prepass.raw_parameters(sc.float(), op.float(), ro) == (a, b, c, f) and not launches
    assert prepass.raw_parameters(sc, op, torch.zeros(n, 4)) is None
    assert prepass.raw_parameters(op, sc, ro) is None            # handles in the wrong slots
    ref = [t.float() for t in prepass_reference(a, b, c, f)]
    got = sc * 1.0
    assert type(got) is torch.Tensor and torch.equal(got, ref[0]) and len(launches) == 1
    assert torch.equal(op[3:5], ref[1][3:5]) and torch.equal(torch.cat([ro, ro])[n:], ref[2]) and len(launches) == 1
    (sc.sum() + (op * op).sum()).backward()
    assert a.grad is not None and b.grad is not None and float(a.grad.abs().sum()) > 0
    assert "tensor" in repr(ro)

    # through the validation layer of diff_gauss with a backend that has no raw-parameter route
    import diff_gauss
    from tests import oracle_backend
    seen = {}

    class Probe(oracle_backend.OracleBackend):
        @staticmethod
        def rasterize(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, settings):
            seen.update(scales=scales, opacities=opacities, rotations=rotations)
            z = torch.zeros(1)
            return z, z, z, z, z
    monkeypatch.setattr(diff_gauss, "_backend", Probe)
    shared2 = prepass._Shared((a, b, c, f))
    h = [prepass.Deferred(shared2, i, s) for i, s in enumerate(((n, 3), (n, 1), (n, 4)))]
    settings = diff_gauss.GaussianRasterizationSettings(8, 8, 1.0, 1.0, 0.1, None, torch.zeros(3), 1.0, torch.eye(4),
                                                        torch.eye(4), 0, torch.zeros(3), False, False)
    diff_gauss.GaussianRasterizer(settings)(means3D=torch.zeros(n, 3), means2D=None, opacities=h[1].float(),
                                            colors_precomp=torch.zeros(n, 3), scales=h[0].float(), rotations=h[2])
    for k, r in zip(("scales", "opacities", "rotations"), ref):
        assert type(seen[k]) is torch.Tensor and torch.equal(seen[k], r), k
