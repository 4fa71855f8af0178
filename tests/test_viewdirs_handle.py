"""sfgs.viewdirs.LazyDirs on CPU tensors: render()'s two view-direction statements (gaussian_renderer/__init__.py:114-115,
:122-123) are RECORDED on the handle `get_xyz` returns, nothing runs; any other use of any of the handles sees exactly what
the statements compute, with their autograd graph."""
import torch

from sfgs import viewdirs as vd


def _setup(n=9, seed=0):
    gen = torch.Generator().manual_seed(seed)
    xyz = torch.nn.Parameter(torch.randn(n, 3, generator=gen))
    campos = torch.randn(3, generator=gen)
    return xyz, campos


def _handle(xyz):
    return vd.LazyDirs(vd.XYZ, xyz, tuple(xyz.shape), xyz)


def test_the_statements_of_render_are_recorded_not_run():
    xyz, campos = _setup()
    h = _handle(xyz)
    assert tuple(h.shape) == (9, 3) and h.requires_grad and h.dtype == torch.float32 and h.shape[0] == 9
    dir_pp = (h - campos.repeat(9, 1))
    assert isinstance(dir_pp, vd.LazyDirs) and dir_pp._sfgs_kind == vd.DIRPP and tuple(dir_pp.shape) == (9, 3)
    norm = dir_pp.norm(dim=1, keepdim=True)
    assert isinstance(norm, vd.LazyDirs) and norm._sfgs_kind == vd.NORM and tuple(norm.shape) == (9, 1)
    dirs = dir_pp / norm
    assert isinstance(dirs, vd.LazyDirs) and dirs._sfgs_kind == vd.DIRS and dirs.contiguous() is dirs
    assert all(t._sfgs_real is None for t in (h, dir_pp, norm, dirs))
    cen = vd.centers_of(dirs, xyz)
    assert cen is not None and torch.equal(cen, campos.repeat(9, 1))
    assert vd.centers_of(dirs, xyz.detach()) is None             # positions of another tensor: not this handle's means3D
    assert vd.centers_of(dir_pp, xyz) is None and vd.centers_of(xyz, xyz) is None
    with torch.no_grad():                                          # the parameter's handle mirrors the parameter (a leaf)
        assert _handle(xyz).requires_grad and not (_handle(xyz) - campos.repeat(9, 1)).requires_grad


def test_other_statements_run_as_ordinary_torch_operations():
    xyz, campos = _setup(seed=1)
    c = campos.repeat(9, 1)
    ref_pp = xyz - c
    ref = ref_pp / ref_pp.norm(dim=1, keepdim=True)
    # another norm / another divisor / another subtrahend: not render()'s statements
    h = _handle(xyz)
    assert not isinstance((h - c).norm(dim=1), vd.LazyDirs)
    assert not isinstance((h - c).norm(p=1, dim=1, keepdim=True), vd.LazyDirs)
    assert not isinstance(h - campos, vd.LazyDirs)                      # broadcast [3]: torch evaluates it
    assert not isinstance(h - 1.0, vd.LazyDirs) and not isinstance(h - c.requires_grad_(True), vd.LazyDirs)
    c = c.detach()
    d1 = h - c
    assert not isinstance(d1 / (h - c).norm(dim=1, keepdim=True), vd.LazyDirs)   # the norm of ANOTHER dir_pp handle
    # materialised values and gradients are the reference's
    dirs = (lambda d: d / d.norm(dim=1, keepdim=True))(_handle(xyz) - c)
    w = torch.randn(9, 3, generator=torch.Generator().manual_seed(4))
    (dirs * w).sum().backward()
    g = xyz.grad.clone(); xyz.grad = None
    (ref * w).sum().backward()
    torch.testing.assert_close(dirs.materialise(), ref, rtol=0, atol=0)
    torch.testing.assert_close(g, xyz.grad, rtol=0, atol=0)
    assert vd.centers_of(dirs, xyz) is None                              # looked into: stays in torch from now on
    # the handle on the parameter simply stands for the parameter
    assert _handle(xyz).materialise() is xyz and torch.equal(_handle(xyz)[2:4], xyz[2:4])
    assert torch.zeros_like(_handle(xyz), dtype=xyz.dtype, requires_grad=True).shape == xyz.shape
    assert float(_handle(xyz).double().sum().detach()) == float(xyz.double().sum().detach())


def test_install_patches_get_xyz_and_uninstall_restores():
    class Model:
        def __init__(self):
            self._xyz, _ = _setup(seed=2)
        get_xyz = property(lambda self: self._xyz)

    orig = Model.__dict__["get_xyz"]
    vd.install(Model)
    try:
        m = Model()
        assert isinstance(m.get_xyz, vd.LazyDirs) and m.get_xyz._sfgs_src is m._xyz and m.get_xyz.shape[0] == 9
    finally:
        vd.uninstall(Model)
    assert Model.__dict__["get_xyz"] is orig and Model().get_xyz.__class__ is torch.nn.Parameter


def test_attribute_reads_other_than_metadata_come_from_the_tensor_the_handle_stands_for():
    """the patched `get_xyz` must behave like the reference's `return self._xyz` for `.grad`, `.grad_fn`,
    `._version`, `.data` too -- and reading them must not switch the recording off."""
    import torch
    from sfgs import features, viewdirs
    p = torch.nn.Parameter(torch.randn(6, 3))
    p.grad = torch.ones(6, 3)
    h = viewdirs.LazyDirs(viewdirs.XYZ, p, (6, 3), p)
    assert h.grad is p.grad and h.grad_fn is None and h._version == p._version and h.is_leaf and h.requires_grad
    assert h.data.data_ptr() == p.data.data_ptr()
    d = h - torch.zeros(6, 3)
    assert isinstance(d, viewdirs.LazyDirs) and d._sfgs_kind == viewdirs.DIRPP          # still recording
    assert d.grad_fn is not None and d._sfgs_real is not None                             # a result's grad_fn: materialised
    dc, rest = torch.nn.Parameter(torch.randn(6, 1, 3)), torch.nn.Parameter(torch.randn(6, 3, 3))
    f = features.DeferredFeatures(dc, rest) if hasattr(features, "DeferredFeatures") else None
    assert tuple(f.shape) == (6, 4, 3) and f.grad_fn is not None      # cat's node, like the reference's getter
