"""PLY I/O. Golden = the REAL reference functions (save_ply / save_fused_ply / load_ply /
load_standard_ply / detect_sh_degree_from_ply / storePly / fetchPly) executed over sfgs.ply's plyfile stand-in
(tests/golden/make_golden.py::make_ply_golden): pins column names, order, transposes and dtypes. The container format
itself is pinned by hand-assembled PLY files below (the PLY specification; what plyfile emits for 'f4'/'u1' fields)."""
import os
import struct
import sys
import types

import numpy as np
import pytest
import torch

from sfgs import ply

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_ply.npz"))
FIELDS = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "filter_3D")


def _model():
    m = types.SimpleNamespace(max_sh_degree=1, active_sh_degree=0, appearance_enabled=False)
    for k in FIELDS:
        setattr(m, k, torch.tensor(G["ply_in" + k]))
    # the two getters save_fused_ply reads (scene/gaussian_model.py:207-213,237-249), float32 filter
    s2 = torch.square(torch.exp(m._scaling))
    f2 = torch.square(m.filter_3D)
    m.get_scaling_with_3D_filter = torch.sqrt(s2 + f2)
    m.get_opacity_with_3D_filter = torch.sigmoid(m._opacity) * torch.sqrt(s2.prod(dim=1) / (s2 + f2).prod(dim=1))[..., None]
    return m


def test_container_known_answer(tmp_path):
    """A PLY file assembled by hand, as the specification (and plyfile) lays it out."""
    header = (b"ply\nformat binary_little_endian 1.0\ncomment made by hand\nelement vertex 2\nproperty float x\n"
              b"property float32 y\nproperty uchar red\nproperty double w\nproperty int id\nend_header\n")
    rows = struct.pack("<ffBdi", 1.5, -2.0, 200, 3.25, -7) + struct.pack("<ffBdi", 0.0, 4.0, 3, -1e300, 9)
    p = tmp_path / "kat.ply"
    p.write_bytes(header + rows)
    (name, t), = ply.read_ply(str(p))
    assert name == "vertex" and t.dtype.names == ("x", "y", "red", "w", "id")
    assert t["x"].tolist() == [1.5, 0.0] and t["red"].tolist() == [200, 3] and t["w"][1] == -1e300 and t["id"][0] == -7
    # big endian and ascii variants of the same table
    pb = tmp_path / "be.ply"
    pb.write_bytes(header.replace(b"little", b"big") + struct.pack(">ffBdi", 1.5, -2.0, 200, 3.25, -7)
                   + struct.pack(">ffBdi", 0.0, 4.0, 3, -1e300, 9))
    assert ply.read_ply(str(pb))[0][1].tolist() == t.tolist()
    pa = tmp_path / "ascii.ply"
    pa.write_bytes(header.replace(b"binary_little_endian", b"ascii") + b"1.5 -2 200 3.25 -7\n0 4 3 -1e300 9\n")
    assert ply.read_ply(str(pa))[0][1].tolist() == t.tolist()
    # writer: exactly the header plyfile produces for this dtype, then the packed little-endian rows
    out = tmp_path / "w.ply"
    ply.write_ply(str(out), t)
    want_header = (b"ply\nformat binary_little_endian 1.0\nelement vertex 2\nproperty float x\nproperty float y\n"
                   b"property uchar red\nproperty double w\nproperty int id\nend_header\n")
    assert out.read_bytes() == want_header + rows
    with pytest.raises(ValueError, match="truncated"):
        (tmp_path / "cut.ply").write_bytes(header + rows[:-3])
        ply.read_ply(str(tmp_path / "cut.ply"))
    with pytest.raises(NotImplementedError):
        (tmp_path / "list.ply").write_bytes(b"ply\nformat ascii 1.0\nelement face 0\nproperty list uchar int vertex_indices\nend_header\n")
        ply.read_ply(str(tmp_path / "list.ply"))


def test_save_ply_is_byte_identical_to_the_reference(tmp_path):
    m = _model()
    p = str(tmp_path / "new_dir" / "point_cloud.ply")  # the directory is created like mkdir_p does
    ply.save_ply(m, p)
    assert np.array_equal(np.fromfile(p, dtype=np.uint8), G["ply_save_bytes"])
    head = G["ply_save_bytes"].tobytes().split(b"end_header\n")[0].decode()
    assert head.split("\n")[3:9] == [f"property float {n}" for n in ("x", "y", "z", "nx", "ny", "nz")]
    assert "property float f_rest_8" in head and head.rstrip().endswith("property float filter_3D")


def test_save_fused_ply_matches_the_reference(tmp_path):
    m = _model()
    p = str(tmp_path / "fused.ply")
    ply.save_fused_ply(m, p)
    got = ply.read_ply(p)[0][1]
    want_path = tmp_path / "want.ply"
    want_path.write_bytes(G["ply_fused_bytes"].tobytes())
    want = ply.read_ply(str(want_path))[0][1]
    assert got.dtype == want.dtype and "filter_3D" not in got.dtype.names
    for n in want.dtype.names:   # opacity / scale go through log, sigmoid, sqrt: libm-level agreement
        np.testing.assert_allclose(got[n], want[n], rtol=2e-6, atol=1e-7, err_msg=n)
    for n in ("x", "f_dc_1", "f_rest_4", "rot_3"):
        assert np.array_equal(got[n], want[n])


def test_loaders_follow_the_reference(tmp_path):
    save, fused = tmp_path / "a.ply", tmp_path / "f.ply"
    save.write_bytes(G["ply_save_bytes"].tobytes())
    fused.write_bytes(G["ply_fused_bytes"].tobytes())
    m = types.SimpleNamespace(max_sh_degree=1, active_sh_degree=0, _xyz=torch.empty(0))
    ply.load_ply(m, str(save), device="cpu")
    assert m._xyz.numel() == 0                      # the reference's load_ply does not assign the parameters
    assert np.array_equal(m.filter_3D.numpy(), G["ply_load_filter_3D"]) and m.active_sh_degree == int(G["ply_load_active_sh_degree"])
    with pytest.raises((KeyError, ValueError)):
        ply.load_ply(m, str(fused), device="cpu")   # a fused file has no filter_3D column
    assert ply.detect_sh_degree(str(fused)) == int(G["ply_detect_sh_degree"]) == 1
    m3 = types.SimpleNamespace(max_sh_degree=1)
    ply.load_standard_ply(m3, str(fused), device="cpu")
    for k in FIELDS:
        got = getattr(m3, k)
        assert np.array_equal(got.detach().numpy(), G["ply_std" + k]), k
        if k != "filter_3D":
            assert isinstance(got, torch.nn.Parameter) and got.requires_grad and got.is_contiguous()
    with pytest.raises(AssertionError):
        ply.load_standard_ply(types.SimpleNamespace(max_sh_degree=2), str(fused), device="cpu")  # degree mismatch


def test_point_cloud_store_fetch(tmp_path):
    p = str(tmp_path / "pts.ply")
    ply.store_ply(p, G["ply_store_xyz"], G["ply_store_rgb"])
    assert np.array_equal(np.fromfile(p, dtype=np.uint8), G["ply_store_bytes"])
    pts, cols, nrm = ply.fetch_ply(p)
    assert np.array_equal(pts, G["ply_fetch_points"]) and np.array_equal(cols, G["ply_fetch_colors"])
    assert np.array_equal(nrm, G["ply_fetch_normals"])


def test_plyfile_standin_and_install(tmp_path):
    had = sys.modules.pop("plyfile", None)
    try:
        mod = ply.install_as_plyfile()
        from plyfile import PlyData, PlyElement  # the import lines of scene/gaussian_model.py:23
        assert mod.PlyData is PlyData
        el = PlyElement.describe(np.zeros(3, dtype=[("x", "f4"), ("red", "u1")]), "vertex")
        PlyData([el]).write(str(tmp_path / "s.ply"))
        back = PlyData.read(str(tmp_path / "s.ply"))
        assert [p.name for p in back.elements[0].properties] == ["x", "red"] and len(back["vertex"]["x"]) == 3
    finally:
        sys.modules.pop("plyfile", None)
        if had is not None:
            sys.modules["plyfile"] = had

    class GaussianModel:
        def save_ply(self, path):
            raise AssertionError
    ply.install(GaussianModel)
    try:
        m = GaussianModel()
        for k, v in vars(_model()).items():
            setattr(m, k, v)
        m.save_ply(str(tmp_path / "i.ply"))
        assert np.array_equal(np.fromfile(str(tmp_path / "i.ply"), dtype=np.uint8), G["ply_save_bytes"])
        assert GaussianModel.load_ply is ply.load_ply
    finally:
        ply.uninstall(GaussianModel)
    assert not hasattr(GaussianModel, "load_ply")


def test_large_table_round_trip(tmp_path):
    """200 k Gaussians at SH degree 3: one vectorised write / read, bit-exact."""
    n = 200_000
    g = torch.Generator().manual_seed(0)
    m = types.SimpleNamespace(max_sh_degree=3, _xyz=torch.randn(n, 3, generator=g), _features_dc=torch.randn(n, 1, 3, generator=g),
                              _features_rest=torch.randn(n, 15, 3, generator=g), _opacity=torch.randn(n, 1, generator=g),
                              _scaling=torch.randn(n, 3, generator=g), _rotation=torch.randn(n, 4, generator=g),
                              filter_3D=torch.rand(n, 1, generator=g))
    p = str(tmp_path / "big.ply")
    ply.save_ply(m, p)
    t = ply.read_ply(p)[0][1]
    assert len(t) == n and len(t.dtype.names) == 6 + 3 + 45 + 1 + 3 + 4 + 1
    assert np.array_equal(t["f_rest_17"], m._features_rest.transpose(1, 2).flatten(start_dim=1)[:, 17].numpy())
    assert np.array_equal(t["filter_3D"], m.filter_3D[:, 0].numpy())


def test_merge_fused_plys(tmp_path):
    a, b = tmp_path / "a.ply", tmp_path / "b.ply"
    a.write_bytes(G["ply_fused_bytes"].tobytes())
    b.write_bytes(G["ply_fused_bytes"].tobytes())
    merged = ply.merge_fused_plys([str(a), str(b)], offsets=[(0, 0, 0), (512.0, -256.0, 1.5)], out_path=str(tmp_path / "j.ply"))
    one = ply.read_ply(str(a))[0][1]
    n = len(one)
    assert len(merged) == 2 * n and merged.dtype == one.dtype
    assert np.array_equal(merged[:n], one)
    assert np.array_equal(merged["x"][n:], one["x"] + np.float32(512.0)) and np.array_equal(merged["z"][n:], one["z"] + np.float32(1.5))
    assert np.array_equal(merged["opacity"][n:], one["opacity"]) and np.array_equal(merged["rot_2"][n:], one["rot_2"])
    assert np.array_equal(ply.read_ply(str(tmp_path / "j.ply"))[0][1], merged)
    m = types.SimpleNamespace(max_sh_degree=1)
    ply.load_standard_ply(m, str(tmp_path / "j.ply"), device="cpu")      # the joint scene loads like any fused PLY
    assert m._xyz.shape == (2 * n, 3)
    save = tmp_path / "s.ply"
    save.write_bytes(G["ply_save_bytes"].tobytes())
    with pytest.raises(ValueError, match="columns differ"):
        ply.merge_fused_plys([str(a), str(save)])
