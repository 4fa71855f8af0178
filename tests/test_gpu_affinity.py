"""sfgs.affinity.auto() through the rasterizer (GPU box): every forward reports its Gaussian count; the process is confined
to a few CPUs of one L3 domain while the scene is small and released once it has grown -- for ALL of its threads (autograd's
device thread exists by then). Results do not depend on it (same bits pinned or not)."""
import os

import numpy as np
import pytest
import torch

from sfgs import affinity
from sfgs.synth import scene, upstream_grads

pytestmark = pytest.mark.gpu


def _step(n, W=256, H=144, seed=3):
    from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda:0")
    frame, g = scene(n, W, H, seed=seed, zrange=(250., 350.), scale_range=(0.2, 2.0))
    settings = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=frame["tanfovx"], tanfovy=frame["tanfovy"], kernel_size=frame["kernel_size"],
        subpixel_offset=None, bg=frame["bg"].to(dev), scale_modifier=1.0, viewmatrix=frame["view"].to(dev),
        projmatrix=frame["proj"].to(dev), sh_degree=0, campos=frame["campos"].to(dev), prefiltered=False, debug=False)
    t = {k: v.to(dev).requires_grad_(True) for k, v in g.items() if v is not None}
    m2 = torch.zeros(n, 3, device=dev, requires_grad=True)
    color, depth, *_ = GaussianRasterizer(settings)(means3D=t["means3D"], means2D=m2, colors_precomp=t["colors_precomp"],
                                                    opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
    gc, gd = upstream_grads(W, H, 1)
    torch.autograd.backward([color, torch.nan_to_num(depth)], [gc.to(dev), gd.to(dev)])
    return color.detach().cpu().numpy(), t["means3D"].grad.cpu().numpy()


@pytest.mark.skipif(not hasattr(os, "sched_setaffinity"), reason="no sched_setaffinity on this platform")
def test_small_frames_pin_the_process_and_grown_ones_release_it():
    before = sorted(os.sched_getaffinity(0))
    threads = torch.get_num_threads()
    try:
        ref_small, ref_small_g = _step(2000)          # (autograd's device thread exists from here on)
        affinity.auto(local_rank=0, cores=2, below=5000, above=8000, min_cpus=2)
        assert affinity.state()["pinned_to"] is None
        a, ag = _step(2000)
        got = affinity.state()["pinned_to"]
        if len(before) >= 2:
            assert got is not None and len(got) <= 2 and sorted(os.sched_getaffinity(0)) == got
            # every thread of the process follows (the backward ran on autograd's thread)
            for tid in os.listdir("/proc/self/task"):
                assert sorted(os.sched_getaffinity(int(tid))) == got, tid
        np.testing.assert_array_equal(a, ref_small)
        np.testing.assert_array_equal(ag, ref_small_g)
        _step(6000)                                   # between the thresholds: stays
        assert affinity.state()["pinned_to"] == got
        _step(20000)                                  # grown: released
        assert affinity.state()["pinned_to"] is None and sorted(os.sched_getaffinity(0)) == before
        assert torch.get_num_threads() == threads
    finally:
        affinity.auto(cores=0)
        affinity.unpin()
        os.sched_setaffinity(0, before)
        torch.set_num_threads(threads)
        affinity._ORIGINAL = affinity._ORIGINAL_THREADS = affinity._PINNED = None
