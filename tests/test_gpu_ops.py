"""GPU parity of fused_ssim, simple_knn._C.distCUDA2 and band rendering (through the drop-in packages ->
C ABI), against the oracle and the golden vectors from the reference's own utils/loss_utils.ssim."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from sfgs.synth import scene

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_helpers.npz"))
DEV = "cuda:0"


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_fused_ssim_matches_reference_golden(tag):
    from fused_ssim import fused_ssim
    img1 = torch.tensor(G[f"ssim_{tag}_img1"], device=DEV, requires_grad=True)
    img2 = torch.tensor(G[f"ssim_{tag}_img2"], device=DEV)
    val = fused_ssim(img1, img2)
    assert val.shape == () and abs(float(val.detach()) - float(G[f"ssim_{tag}_value"])) < 2e-6  # 1e-6 class
    (0.2 * (1.0 - val)).backward()  # lambda_dssim * (1 - ssim), train.py:222-224
    ref = -0.2 * G[f"ssim_{tag}_grad"]
    got = img1.grad.cpu().numpy()
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-10


def test_fused_ssim_1080p_against_oracle_and_determinism():
    from fused_ssim import fused_ssim
    g = torch.Generator().manual_seed(0)
    a = torch.rand(1, 3, 270, 480, generator=g)
    b = (a + 0.05 * torch.randn(1, 3, 270, 480, generator=g)).clamp(0, 1)
    val, _, grad = orc.ssim(a.numpy(), b.numpy(), want_grad=True)
    x = a.to(DEV).requires_grad_(True)
    v = fused_ssim(x, b.to(DEV))
    v.backward()
    assert abs(float(v.detach()) - val) < 2e-6
    assert np.abs(x.grad.cpu().numpy() - grad).max() <= 5e-5 * np.abs(grad).max()
    v2 = fused_ssim(x.detach(), b.to(DEV), train=False)
    assert float(v2) == float(v)  # fixed-order reduction: bit-reproducible
    # full size smoke (shape the training loop uses): finite, in range
    big = torch.rand(1, 3, 1080, 1920, device=DEV)
    s = fused_ssim(big, big)
    assert abs(float(s) - 1.0) < 1e-5


@pytest.mark.parametrize("shape", [(1, 1, 1, 1), (1, 1, 1, 7), (1, 2, 5, 3), (2, 1, 4, 43), (1, 1, 21, 31), (1, 3, 22, 32),
                                   (1, 1, 23, 33), (1, 2, 44, 64), (2, 3, 45, 65), (1, 1, 100, 37), (1, 1, 67, 130),
                                   (1, 1, 11, 200)])
def test_fused_ssim_at_tile_boundaries_and_tiny_images_against_oracle(shape):
    """Image sizes around the kernel's 32 x 22 output tile and its 5-pixel halo (one short of, equal to, one past a tile;
    images smaller than the window; widths that are not a multiple of 4; batches and channels): value, per-pixel map
    through the gradient, and the train=False route, against the oracle."""
    from fused_ssim import fused_ssim
    g = torch.Generator().manual_seed(shape[2] * 1000 + shape[3])
    a = torch.rand(*shape, generator=g)
    b = (a + 0.1 * torch.randn(*shape, generator=g)).clamp(0, 1)
    val, _, grad = orc.ssim(a.numpy(), b.numpy(), want_grad=True)
    x = a.to(DEV).requires_grad_(True)
    v = fused_ssim(x, b.to(DEV))
    v.backward()
    assert abs(float(v.detach()) - val) < 2e-6
    assert np.abs(x.grad.cpu().numpy() - grad).max() <= 5e-5 * np.abs(grad).max() + 1e-9
    assert float(fused_ssim(a.to(DEV), b.to(DEV), train=False)) == float(v.detach())
    assert abs(float(fused_ssim(b.to(DEV), b.to(DEV), train=False)) - 1.0) < 1e-6


@pytest.mark.parametrize("n", [1, 2, 3, 5, 1000, 20000])
def test_distcuda2_matches_brute_force_oracle(n):
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(n)
    pts = rng.normal(size=(n, 3)).astype(np.float32) * 3
    if n > 10:
        pts[7] = pts[3]  # duplicate point: distance 0 counts (index-excluded, not value-excluded)
    got = distCUDA2(torch.tensor(pts, device=DEV)).cpu().numpy()
    ref = orc.knn_dist2(pts)
    assert got.shape == (n,)
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-9)
    assert distCUDA2(torch.zeros(0, 3, device=DEV)).shape == (0,)


def test_band_renders_tile_the_full_frame_bit_exactly():
    from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer
    from sfgs import shard
    W, H = 320, 203
    frame, g = scene(20000, W, H, seed=9, zrange=(250., 350.), scale_range=(0.2, 3.0))
    dev = torch.device(DEV)
    base = GaussianRasterizationSettings(H, W, frame["tanfovx"], frame["tanfovy"], frame["kernel_size"], None,
                                         frame["bg"].to(dev), 1.0, frame["view"].to(dev), frame["proj"].to(dev), 0,
                                         frame["campos"].to(dev), False, False)
    inp = dict(means3D=g["means3D"].to(dev), means2D=None, opacities=g["opacities"].to(dev),
               colors_precomp=g["colors_precomp"].to(dev), scales=g["scales"].to(dev), rotations=g["rotations"].to(dev))
    with torch.no_grad():
        full = GaussianRasterizer(base)(**inp)
        world = 3
        acc = [torch.full_like(full[i], float("inf")) for i in (0, 1, 3)]
        total_dups = 0
        for r in range(world):
            t0, t1, a, b = shard.band_rows(H, world, r)
            out = GaussianRasterizer(base._replace(tile_rows=(t0, t1)))(**inp)
            from diff_gauss import last_counters
            total_dups += last_counters()["num_duplicates"]
            for k, i in enumerate((0, 1, 3)):
                acc[k][:, a:b] = out[i][:, a:b]
            assert torch.equal(out[4], full[4])  # radii do not depend on the band
    for k, i in enumerate((0, 1, 3)):
        assert torch.equal(torch.nan_to_num(acc[k], nan=-1.0), torch.nan_to_num(full[i], nan=-1.0))
    # per-rank binning work shrinks with the band: the bands' duplicates partition the full frame's
    GaussianRasterizer(base)(**inp)
    from diff_gauss import last_counters
    assert total_dups == last_counters()["num_duplicates"]
