"""sfgs.max_radii: train.py:314's masked max without nonzero / host waits ("fuse max_radii2D").
CPU: the subclasses are host logic over torch -- the statement's result, every fall-back to ordinary tensor semantics, and the
absence of `aten::nonzero` in the fused statement. GPU: the rasterizer hands out the RadiiTensor and render()'s own
`radii > 0` arrives as a VisMask."""
import pytest
import torch
from torch.profiler import ProfilerActivity, profile

from sfgs import max_radii as mr


@pytest.fixture()
def on():
    mr.install()
    yield
    mr.uninstall()


def _inputs(n=5000, seed=0):
    g = torch.Generator().manual_seed(seed)
    radii = torch.randint(-4, 30, (n,), generator=g).clamp_min(0).to(torch.int32)
    m = torch.rand(n, generator=g) * 12
    m[::7] = 0.0
    return radii, m


def _ops(fn):
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        fn()
    return {e.key for e in prof.key_averages()}


def test_statement_is_the_references_result_and_launches_no_nonzero(on):
    radii, m0 = _inputs()
    want = m0.clone()
    vf = radii > 0
    want[vf] = torch.max(want[vf], radii[vf])                      # train.py:314 on plain tensors
    got = m0.clone()
    r = mr.wrap_radii(radii.clone())
    assert type(r) is mr.RadiiTensor and r.data_ptr() != 0 and torch.equal(r.as_subclass(torch.Tensor), radii)
    v = r > 0                                                      # gaussian_renderer/__init__.py:160-162
    assert type(v) is mr.VisMask and v.dtype == torch.bool and torch.equal(v.as_subclass(torch.Tensor), vf)

    def stmt():
        got[v] = torch.max(got[v], r[v])
    ops = _ops(stmt)
    assert torch.equal(got, want) and got.dtype == torch.float32
    assert "aten::nonzero" not in ops and "aten::index" not in ops and "aten::index_put_" not in ops, sorted(ops)
    # the plain statement does: that is what the hook removes
    plain = m0.clone()
    ops_plain = _ops(lambda: plain.__setitem__(vf, torch.max(plain[vf], radii[vf])))
    assert "aten::nonzero" in ops_plain


def test_every_other_use_sees_ordinary_tensors(on):
    radii, m0 = _inputs(seed=3)
    vf = radii > 0
    r = mr.wrap_radii(radii.clone())
    v = r > 0
    m = m0.clone()
    # a selection that is used in any other way is the reference's own `base[mask]`
    assert torch.equal(m[v].clone(), m0[vf]) and float(m[v].sum()) == float(m0[vf].sum()) and len(m[v]) == int(vf.sum())
    assert torch.equal(m[v] * 2 + 1, m0[vf] * 2 + 1) and torch.equal(r[v] + 1, radii[vf] + 1)
    assert torch.equal(torch.max(m[v]), torch.max(m0[vf]))                              # one-argument max: not the pattern
    assert torch.equal(torch.max(m[v], m[v] * 0 + 3.0), torch.max(m0[vf], m0[vf] * 0 + 3.0))   # second operand a plain tensor
    # other assignments through the mask
    a, b = m0.clone(), m0.clone()
    a[v] = 5.0; b[vf] = 5.0
    assert torch.equal(a, b)
    a[v] = a[v] * 0.5; b[vf] = b[vf] * 0.5
    assert torch.equal(a, b)
    a[v] = torch.max(a[v], r[v]) + 1.0; b[vf] = torch.max(b[vf], radii[vf]) + 1.0          # a max that is used further
    assert torch.equal(a, b)
    # the mask itself: arithmetic, reductions, tuple indexing (add_densification_stats: grad[update_filter, :2]), views
    assert type(v & v) is torch.Tensor and int(v.sum()) == int(vf.sum()) and torch.equal(~v, ~vf)
    g = torch.arange(3.0 * len(radii)).reshape(-1, 3)
    assert torch.equal(g[v, :2], g[vf, :2]) and torch.equal(v.view(torch.uint8), vf.view(torch.uint8))
    x, y = torch.zeros(len(radii), 1), torch.zeros(len(radii), 1)
    x[v] += torch.norm(g[v, :2], dim=-1, keepdim=True); y[vf] += torch.norm(g[vf, :2], dim=-1, keepdim=True)   # gaussian_model.py:745
    assert torch.equal(x, y)
    # radii: everything but `> 0` is plain
    assert type(r + 1) is torch.Tensor and type(r > 1) is torch.Tensor and type(r.float()) is torch.Tensor
    assert torch.equal(r[vf], radii[vf]) and int(r.max()) == int(radii.max())
    # copies and pickles are ordinary tensors (a checkpoint or a deepcopy of render()'s result must not carry the marks)
    import copy
    import io
    assert type(copy.deepcopy(r)) is torch.Tensor and type(copy.deepcopy(v)) is torch.Tensor and type(r.clone()) is torch.Tensor
    buf = io.BytesIO()
    torch.save({"r": r, "v": v}, buf)
    buf.seek(0)
    back = torch.load(buf, weights_only=True)
    assert type(back["r"]) is torch.Tensor and torch.equal(back["r"], radii) and torch.equal(back["v"], vf)
    assert r.numpy().tolist() == radii.tolist() and torch.equal(torch.where(v)[0], torch.where(vf)[0])
    # a mask of another length is not the pattern
    short = torch.zeros(7)
    assert torch.equal(short[mr.wrap_radii(torch.arange(7, dtype=torch.int32)) > 0], short[1:])


def test_uninstalled_means_plain_tensors():
    mr.uninstall()
    radii, _ = _inputs()
    assert type(mr.wrap_radii(radii)) is torch.Tensor


@pytest.mark.gpu
def test_rasterizer_hands_out_the_handle_and_the_statement_matches_plain_torch(on):
    from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer
    from sfgs.synth import scene
    dev = torch.device("cuda:0")
    W, H, N = 320, 200, 20000
    frame, g = scene(N, W, H, seed=1, zrange=(4.0, 8.0), scale_range=(0.01, 0.2), xy_fill=1.3)    # some Gaussians off screen
    s = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=frame["tanfovx"], tanfovy=frame["tanfovy"],
                                      kernel_size=0.1, subpixel_offset=None, bg=frame["bg"].to(dev), scale_modifier=1.0,
                                      viewmatrix=frame["view"].to(dev), projmatrix=frame["proj"].to(dev), sh_degree=0,
                                      campos=frame["campos"].to(dev), prefiltered=False, debug=False)
    t = {k: v.to(dev) for k, v in g.items() if v is not None}
    *_, radii, _none = GaussianRasterizer(s)(means3D=t["means3D"], means2D=None, colors_precomp=t["colors_precomp"],
                                             opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
    assert type(radii) is mr.RadiiTensor
    vis = radii > 0
    assert type(vis) is mr.VisMask and 0 < int(vis.sum()) < N
    m = torch.rand(N, device=dev) * 3
    want = m.clone()
    pr, pv = radii.as_subclass(torch.Tensor), vis.as_subclass(torch.Tensor)
    want[pv] = torch.max(want[pv], pr[pv])
    m[vis] = torch.max(m[vis], radii[vis])
    assert torch.equal(m, want)
