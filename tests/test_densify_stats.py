"""add_densification_stats: fused in-place kernel against golden vectors produced by the reference's
REAL GaussianModel.add_densification_stats over two consecutive steps."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_helpers.npz"))
NAMES = ("xyz_gradient_accum", "xyz_gradient_accum_abs", "xyz_gradient_accum_abs_max", "denom")


def test_golden_is_self_consistent():
    """numpy restatement of scene/gaussian_model.py:744-749 reproduces the golden (pins our reading of the code)."""
    n = G["dstats_denom"].shape[0]
    acc = {k: np.zeros((n, 1), np.float32) for k in NAMES}
    for step in range(2):
        g, f = G[f"dstats_grad{step}"], G[f"dstats_filter{step}"]
        acc["xyz_gradient_accum"][f] += np.linalg.norm(g[f, :2], axis=-1, keepdims=True)
        na = np.linalg.norm(g[f, 2:], axis=-1, keepdims=True)
        acc["xyz_gradient_accum_abs"][f] += na
        acc["xyz_gradient_accum_abs_max"][f] = np.maximum(acc["xyz_gradient_accum_abs_max"][f], na)
        acc["denom"][f] += 1
    for k in NAMES:
        np.testing.assert_allclose(acc[k], G["dstats_" + k], rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
def test_fused_kernel_matches_real_method():
    from sfgs import densify_stats
    n = G["dstats_denom"].shape[0]
    m = SimpleNamespace(**{k: torch.zeros(n, 1, device="cuda:0") for k in NAMES})
    for step in range(2):
        vs = SimpleNamespace(grad=torch.tensor(G[f"dstats_grad{step}"], device="cuda:0"))
        densify_stats.add_densification_stats(m, vs, torch.tensor(G[f"dstats_filter{step}"], device="cuda:0"))
    for k in NAMES:
        np.testing.assert_allclose(getattr(m, k).cpu().numpy(), G["dstats_" + k], rtol=1e-6, atol=1e-7)
    # works without the *_abs_max buffer too, and install() swaps the method
    class GaussianModel:
        def add_densification_stats(self, v, f):
            raise AssertionError
    densify_stats.install(GaussianModel)
    try:
        mm = GaussianModel()
        for k in NAMES:
            if k != "xyz_gradient_accum_abs_max":
                setattr(mm, k, torch.zeros(n, 1, device="cuda:0"))
        mm.add_densification_stats(SimpleNamespace(grad=torch.tensor(G["dstats_grad0"], device="cuda:0")),
                                   torch.tensor(G["dstats_filter0"], device="cuda:0"))
        assert float(mm.denom.sum()) == float(G["dstats_filter0"].sum())
    finally:
        densify_stats.uninstall(GaussianModel)
