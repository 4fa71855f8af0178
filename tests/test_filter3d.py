"""compute_3D_filter: numpy oracle pinned to the reference's REAL GaussianModel.compute_3D_filter
(golden), and the fused HIP pass against both."""
import os

import numpy as np
import pytest
import torch

from oracle.filter3d_np import compute_3D_filter as oracle_filter

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_helpers.npz"))


def golden_cameras():
    cams = []
    for i in range(G["cam_R"].shape[0]):
        cams.append(dict(R=G["cam_R"][i], T=G["cam_T"][i], cx=float(G["cam_cx"][i]), cy=float(G["cam_cy"][i]),
                         image_width=int(G["cam_W"][i]), image_height=int(G["cam_H"][i]),
                         focal_x=float(G["cam_focal_x"][i]), focal_y=float(G["cam_focal_y"][i])))
    return cams


def test_oracle_matches_real_compute_3D_filter():
    out = oracle_filter(G["filter3d_xyz"], golden_cameras())
    np.testing.assert_allclose(out, G["filter3d_out"], rtol=1e-12)
    assert len(np.unique(out[:5])) == 1  # the unseen points share the largest seen distance


@pytest.mark.gpu
def test_fused_filter_matches_golden_and_oracle():
    from types import SimpleNamespace
    from sfgs.filter3d import compute_3D_filter
    cams = [SimpleNamespace(**c) for c in golden_cameras()]
    out = compute_3D_filter(torch.tensor(G["filter3d_xyz"], device="cuda:0"), cams)
    assert out.dtype == torch.float64 and out.shape == (3000, 1)
    np.testing.assert_allclose(out.cpu().numpy(), G["filter3d_out"], rtol=1e-12)
    # a larger cloud against the oracle
    g = torch.Generator().manual_seed(3)
    xyz = torch.randn(200_000, 3, generator=g) * 8.0
    ref = oracle_filter(xyz.numpy(), golden_cameras())
    got = compute_3D_filter(xyz.to("cuda:0"), cams).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-12)


@pytest.mark.gpu
def test_install_replaces_the_method():
    from types import SimpleNamespace
    from sfgs import filter3d

    class GaussianModel:
        def __init__(self):
            self._xyz = torch.tensor(G["filter3d_xyz"], device="cuda:0")
            self.filter_3D = None

        @property
        def get_xyz(self):
            return self._xyz

        def compute_3D_filter(self, cameras):
            raise AssertionError("the reference loop must not run once the fused pass is installed")

    filter3d.install(GaussianModel)
    try:
        m = GaussianModel()
        m.compute_3D_filter([SimpleNamespace(**c) for c in golden_cameras()])
        np.testing.assert_allclose(m.filter_3D.cpu().numpy(), G["filter3d_out"], rtol=1e-12)
    finally:
        filter3d.uninstall(GaussianModel)
