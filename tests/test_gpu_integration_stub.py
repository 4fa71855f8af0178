"""Executes the raw-ctypes binding shown in INTEGRATION.md ("The ctypes stub") verbatim against libsfgs.so and compares
its frame with the packaged wrapper's (the stub was documentation no test ran)."""
import os
import re

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_stub_runs_and_matches_the_wrapper():
    from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer
    from sfgs import _lib as L
    from sfgs.synth import scene
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"## The ctypes stub.*?```python\n(.*?)```", text, re.S).group(1)
    block = block.replace('C.CDLL("libsfgs.so")', f'C.CDLL({L.LIB_PATH!r})')
    ns = dict(SfgsGaussians=L.SfgsGaussians, SfgsGaussianGrads=L.SfgsGaussianGrads, SfgsRasterSizes=L.SfgsRasterSizes,
              SfgsRasterCounters=L.SfgsRasterCounters)
    exec(compile(block, "INTEGRATION.md", "exec"), ns)
    ns["lib"].sfgs_last_error.restype = __import__("ctypes").c_char_p
    dev = torch.device("cuda:0")
    frame, g = scene(30000, 400, 240, seed=9, zrange=(250., 350.), scale_range=(0.2, 2.4))
    settings = GaussianRasterizationSettings(240, 400, frame["tanfovx"], frame["tanfovy"], frame["kernel_size"], None,
                                             frame["bg"].to(dev), 1.0, frame["view"].to(dev), frame["proj"].to(dev), 0,
                                             frame["campos"].to(dev), False, False)
    t = {k: v.to(dev) for k, v in g.items() if v is not None}
    with torch.no_grad():
        color, depth, alpha, radii = ns["rasterize"](settings, t["means3D"], t["scales"], t["rotations"], t["opacities"],
                                                     t["colors_precomp"])
        ref = GaussianRasterizer(settings)(means3D=t["means3D"], means2D=None, opacities=t["opacities"],
                                           colors_precomp=t["colors_precomp"], scales=t["scales"], rotations=t["rotations"])
    torch.cuda.synchronize()
    assert torch.equal(color, ref[0]) and torch.equal(radii, ref[4]) and torch.equal(alpha, ref[3])
    assert torch.equal(torch.nan_to_num(depth, nan=-1.0), torch.nan_to_num(ref[1], nan=-1.0))
    assert float(alpha.mean()) > 0.01 and np.isfinite(color.cpu().numpy()).all()
