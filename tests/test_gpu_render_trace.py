"""Replay of the argument trace of the reference's REAL render() (tests/golden/reference_render_trace.npz, recorded by
tests/golden/make_golden_r3.py while gaussian_renderer.render() ran on the real GaussianModel) into
the HIP path: every call's tensors are rebuilt with the recorded dtypes / shapes / STRIDES, handed to
diff_gauss.GaussianRasterizer exactly as render() does (gaussian_renderer/__init__.py:40-57,132-140), differentiated
with the upstream gradients autograd delivered in the recording, and compared with the oracle outputs / gradients
recorded beside them."""
import json
import os

import numpy as np
import pytest
import torch

import oracle_backend as ob
import parity

pytestmark = pytest.mark.gpu
TRACE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_render_trace.npz")
_Z = np.load(TRACE)
_IDX = json.loads(str(_Z["index"]))


@pytest.mark.parametrize("i", range(len(_IDX["calls"])), ids=[c["name"] for c in _IDX["calls"]])
def test_replay(i):
    import diff_gauss
    assert diff_gauss._backend.name == "hip"
    c, pre, dev = _IDX["calls"][i], f"c{i}_", torch.device("cuda:0")
    get = lambda k: _Z[pre + k] if (pre + k) in _Z.files else None
    st = {k: ob.rebuild(c["settings_meta"][k], get("set_" + k), dev) for k in ob.SETTING_TENSORS}
    settings = diff_gauss.GaussianRasterizationSettings(**{**c["settings_scalars"], **st})
    args = {k: ob.rebuild(c["meta"][k], get("in_" + k), dev) for k in ob.TENSOR_ARGS}
    for k, m in c["meta"].items():
        if m is not None:
            assert list(args[k].stride()) == m["stride"] and str(args[k].dtype) == "torch." + m["dtype"]
    leaves = {k: v for k, v in args.items() if v is not None}
    if c["has_backward"]:
        for v in leaves.values():
            v.requires_grad_(True)
    rast = diff_gauss.GaussianRasterizer(raster_settings=settings)
    with torch.set_grad_enabled(c["has_backward"]):
        color, depth, norm, alpha, radii, extra = rast(means3D=args["means3D"], means2D=args["means2D"], shs=args["shs"],
                                                       colors_precomp=args["colors_precomp"], opacities=args["opacities"],
                                                       scales=args["scales"], rotations=args["rotations"],
                                                       cov3Ds_precomp=None)
    assert extra is None and tuple(norm.shape) == tuple(color.shape) and radii.dtype == torch.int32
    np.testing.assert_array_equal(radii.cpu().numpy(), get("out_radii"))
    small = 4   # a 128 x 80 image: one borderline splat may flip a handful of pixels
    parity.assert_image_close("color", color.detach().cpu().numpy(), get("out_color"), borderline_min=small)
    parity.assert_image_close("depth", depth.detach().cpu().numpy(), get("out_depth"), borderline_min=small)
    parity.assert_image_close("alpha", alpha.detach().cpu().numpy(), get("out_alpha"), borderline_min=small)
    if not c["has_backward"]:
        return
    up_c, up_d = torch.from_numpy(get("up_color")).to(dev), torch.from_numpy(get("up_depth")).to(dev)
    up_d = torch.where(torch.isnan(depth.detach()), torch.zeros_like(up_d), up_d)   # train.py:229-230 zeroes those pixels
    torch.autograd.backward([color, depth], [up_c, up_d])
    for k, v in leaves.items():
        ref = get("grad_" + k)
        assert v.grad is not None and ref is not None, k
        got = v.grad.detach().cpu().numpy().reshape(ref.shape)
        if np.abs(ref).max() == 0:
            assert np.abs(got).max() == 0, k
        else:
            parity.assert_grad_close(k, got, ref)
