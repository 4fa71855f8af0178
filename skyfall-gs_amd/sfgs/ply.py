"""PLY vertex-table I/O for Gaussian point clouds (the data format on either side of the path).

The reference reads and writes its point clouds through the third-party `plyfile` package
(`scene/gaussian_model.py:23,418-547`, `scene/dataset_readers.py:22,126-148`, `render_video_from_ply.py:163-275`),
building every file row by row in Python (`elements[:] = list(map(tuple, attributes))`: minutes for a few million
Gaussians). Here:

  * `read_ply` / `write_ply`: the PLY container itself (header + one scalar-property element table) as numpy
    structured arrays, binary little/big endian and ascii; a write is one `ndarray.tofile`.
  * `save_ply`, `save_fused_ply`, `load_ply`, `load_standard_ply`, `fetch_ply`, `store_ply`: the reference's
    functions with the same column names, order, transposes and quirks, bound to a `GaussianModel`-shaped object;
    `install(GaussianModel)` swaps the three methods.
  * `PlyData` / `PlyElement`: the sliver of plyfile's API the reference touches. `install_as_plyfile()` registers it
    as module `plyfile` ONLY when the real package is not importable, so the reference's own readers run unchanged on
    an image without plyfile (this one).

Host-side code: there is nothing for the GPU to do in a file format; tensors go through one `.cpu()` each."""
import os
import sys
import types

import numpy as np
import torch

__all__ = ["read_ply", "write_ply", "save_ply", "save_fused_ply", "load_ply", "load_standard_ply", "fetch_ply",
           "store_ply", "detect_sh_degree", "merge_fused_plys", "PlyData", "PlyElement", "install", "uninstall",
           "install_as_plyfile"]

# PLY scalar type names <-> numpy codes (both spellings are legal in headers; plyfile writes the short ones)
_PLY2NP = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
           "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
           "double": "f8", "float64": "f8"}
_NP2PLY = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float",
           "f8": "double"}


def _header(elements, fmt):
    lines = ["ply", f"format {fmt} 1.0"]
    for name, table in elements:
        lines.append(f"element {name} {len(table)}")
        for field in table.dtype.names:
            code = table.dtype[field].str[1:]
            if code not in _NP2PLY:
                raise ValueError(f"property '{field}': dtype {table.dtype[field]} has no PLY scalar type")
            lines.append(f"property {_NP2PLY[code]} {field}")
    lines.append("end_header")
    return ("\n".join(lines) + "\n").encode("ascii")


def write_ply(path, table, element="vertex", text=False):
    """Write one element table (numpy structured array of scalar fields) as a PLY file."""
    table = np.asarray(table)
    if table.dtype.names is None or table.ndim != 1:
        raise ValueError("write_ply expects a 1-D structured array")
    with open(path, "wb") as f:
        if text:
            f.write(_header([(element, table)], "ascii"))
            np.savetxt(f, table, fmt=["%.9g" if table.dtype[n].kind == "f" else "%d" for n in table.dtype.names])
            return
        f.write(_header([(element, table)], "binary_little_endian"))
        le = table.dtype.newbyteorder("<")
        (table if table.dtype == le else table.astype(le)).tofile(f)


def read_ply(path):
    """-> list of (element name, structured array). Scalar properties only (point clouds have no lists)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elements = None, []
        while True:
            raw = f.readline()
            if not raw:
                raise ValueError(f"{path}: header not terminated")
            tok = raw.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append((tok[1], int(tok[2]), []))
            elif tok[0] == "property":
                if tok[1] == "list":
                    raise NotImplementedError(f"{path}: list property '{tok[-1]}' (only scalar properties are supported)")
                if tok[1] not in _PLY2NP:
                    raise ValueError(f"{path}: unknown property type '{tok[1]}'")
                elements[-1][2].append((tok[2], _PLY2NP[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt not in ("binary_little_endian", "binary_big_endian", "ascii"):
            raise ValueError(f"{path}: unsupported format '{fmt}'")
        out = []
        for name, count, props in elements:
            if fmt == "ascii":
                native = np.dtype([(n, c) for n, c in props])
                rows = np.loadtxt(f, dtype=np.float64, max_rows=count, ndmin=2) if count else np.zeros((0, len(props)))
                if rows.shape != (count, len(props)):
                    raise ValueError(f"{path}: element '{name}' truncated")
                table = np.empty(count, dtype=native)
                for i, (n, _) in enumerate(props):
                    table[n] = rows[:, i]
            else:
                order = "<" if fmt == "binary_little_endian" else ">"
                dt = np.dtype([(n, order + c) for n, c in props])
                table = np.fromfile(f, dtype=dt, count=count)
                if len(table) != count:
                    raise ValueError(f"{path}: element '{name}' truncated ({len(table)} of {count} rows)")
                table = table.astype(dt.newbyteorder("="))
            out.append((name, table))
        return out


# ---- the sliver of plyfile's API the reference uses --------------------------------------------------------------
class _Property:
    def __init__(self, name, dtype):
        self.name, self.dtype = name, dtype

    def __repr__(self):
        return f"PlyProperty({self.name!r}, {self.dtype!r})"


class PlyElement:
    def __init__(self, name, data):
        self.name, self.data = name, data

    @staticmethod
    def describe(data, name):
        data = np.asarray(data)
        if data.dtype.names is None:
            raise ValueError("only structured arrays can be described as PLY elements")
        return PlyElement(name, data)

    @property
    def properties(self):
        return tuple(_Property(n, self.data.dtype[n].str[1:]) for n in self.data.dtype.names)

    @property
    def count(self):
        return len(self.data)

    def __getitem__(self, key):
        return self.data[key]

    def __len__(self):
        return len(self.data)


class PlyData:
    def __init__(self, elements=(), text=False, byte_order="="):
        self.elements, self.text = list(elements), text

    @staticmethod
    def read(path):
        return PlyData([PlyElement(n, t) for n, t in read_ply(path)])

    def write(self, path):
        if len(self.elements) != 1:
            raise NotImplementedError("this writer handles the single-element files of the reference")
        write_ply(path, self.elements[0].data, self.elements[0].name, text=self.text)

    def __getitem__(self, name):
        for e in self.elements:
            if e.name == name:
                return e
        raise KeyError(name)


def install_as_plyfile():
    """Make `from plyfile import PlyData, PlyElement` work when the real package is absent. Returns the module used."""
    try:
        import plyfile  # noqa: F401  (the real one wins)
        return sys.modules["plyfile"]
    except ImportError:
        mod = types.ModuleType("plyfile")
        mod.PlyData, mod.PlyElement = PlyData, PlyElement
        mod.__doc__ = "sfgs.ply stand-in for the plyfile API used by Skyfall-GS"
        sys.modules["plyfile"] = mod
        return mod


# ---- the reference's point-cloud functions ------------------------------------------------------------------------
def _np(t):
    return t.detach().cpu().numpy()


def _attribute_names(model, exclude_filter=False):
    """construct_list_of_attributes, scene/gaussian_model.py:402-416."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(model._features_dc.shape[1] * model._features_dc.shape[2])]
    names += [f"f_rest_{i}" for i in range(model._features_rest.shape[1] * model._features_rest.shape[2])]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(model._scaling.shape[1])]
    names += [f"rot_{i}" for i in range(model._rotation.shape[1])]
    if not exclude_filter:
        names.append("filter_3D")
    return names


def _write_columns(path, names, blocks):
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)  # mkdir_p of the reference
    cols = np.concatenate(blocks, axis=1)
    if cols.shape[1] != len(names):
        raise ValueError(f"{cols.shape[1]} columns for {len(names)} attribute names")
    table = np.empty(cols.shape[0], dtype=[(n, "f4") for n in names])
    # one strided copy per file instead of one Python tuple per Gaussian; same float32 values
    table.view(np.float32).reshape(cols.shape[0], len(names))[:] = cols
    write_ply(path, table)


def save_ply(model, path):
    """GaussianModel.save_ply, scene/gaussian_model.py:418-436 (features channel-major, raw opacity/scale, filter_3D)."""
    xyz = _np(model._xyz)
    blocks = [xyz, np.zeros_like(xyz),
              _np(model._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous()),
              _np(model._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous()),
              _np(model._opacity), _np(model._scaling), _np(model._rotation), _np(model.filter_3D)]
    _write_columns(path, _attribute_names(model), blocks)


def save_fused_ply(model, path, color_mapped=False):
    """GaussianModel.save_fused_ply, scene/gaussian_model.py:438-481: bakes the 3D filter into opacity/scale (and
    optionally the appearance MLP's tone mapping into the SH coefficients); no filter_3D column."""
    xyz = _np(model._xyz)
    if getattr(model, "appearance_enabled", False) and color_mapped:
        uid = min(model.appearance_embeddings.shape[0] - 1, 6)
        embedding = model.appearance_embeddings[uid]
        expanded = embedding[None].repeat(model._xyz.shape[0], 1)
        toned = model.appearance_mlp(model._embeddings, expanded, model.get_features).clamp_max(1.0)
        shdim = (model.max_sh_degree + 1) ** 2
        toned = toned.view(-1, shdim, 3).contiguous().clamp_max(1.0)
        f_dc = _np(toned[:, :1, :].detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().clamp_max(1.0))
        f_rest = _np(toned[:, 1:, :].detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().clamp_max(1.0))
    else:
        f_dc = _np(model._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous())
        f_rest = _np(model._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous())
    op = model.get_opacity_with_3D_filter
    opacities = _np(torch.log(op / (1 - op)))                       # inverse_sigmoid, utils/general_utils.py:18
    scale = _np(torch.log(model.get_scaling_with_3D_filter))        # scaling_inverse_activation
    _write_columns(path, _attribute_names(model, exclude_filter=True),
                   [xyz, np.zeros_like(xyz), f_dc, f_rest, opacities, scale, _np(model._rotation)])


def _sorted_columns(table, prefix):
    names = [n for n in table.dtype.names if n.startswith(prefix)]
    return sorted(names, key=lambda n: int(n.split("_")[-1]))


def _vertex_table(path):
    return read_ply(path)[0][1]


def _parse_gaussians(table, max_sh_degree):
    """The column -> tensor mapping shared by load_ply (scene/gaussian_model.py:503-547) and load_standard_ply
    (render_video_from_ply.py:229-275): float64 staging arrays like the reference, f_rest reshaped [N, 3, K-1]."""
    n = len(table)
    xyz = np.stack((np.asarray(table["x"]), np.asarray(table["y"]), np.asarray(table["z"])), axis=1)
    opacities = np.asarray(table["opacity"])[..., np.newaxis]
    features_dc = np.zeros((n, 3, 1))
    for c in range(3):
        features_dc[:, c, 0] = np.asarray(table[f"f_dc_{c}"])
    extra = _sorted_columns(table, "f_rest_")
    assert len(extra) == 3 * (max_sh_degree + 1) ** 2 - 3
    features_extra = np.zeros((n, len(extra)))
    for i, name in enumerate(extra):
        features_extra[:, i] = np.asarray(table[name])
    features_extra = features_extra.reshape((n, 3, (max_sh_degree + 1) ** 2 - 1))
    scale_names, rot_names = _sorted_columns(table, "scale_"), _sorted_columns(table, "rot")
    scales = np.stack([np.asarray(table[s]) for s in scale_names], axis=1).astype(np.float64)
    rots = np.stack([np.asarray(table[r]) for r in rot_names], axis=1).astype(np.float64)
    return xyz, opacities, features_dc, features_extra, scales, rots


def load_ply(model, path, device="cuda"):
    """GaussianModel.load_ply AS THE REFERENCE HAS IT (scene/gaussian_model.py:503-547): the file is parsed and
    validated, but the assignments of xyz / features / opacity / scaling / rotation are commented out upstream --
    only `filter_3D` (float32 [N,1]) and `active_sh_degree` are set. Kept faithful: callers rely on it
    (create_fused_ply.py:29 loads a checkpointed model first)."""
    table = _vertex_table(path)
    _parse_gaussians(table, model.max_sh_degree)  # same asserts / KeyErrors as the reference
    filter_3D = np.asarray(table["filter_3D"])[..., np.newaxis]
    model.filter_3D = torch.tensor(filter_3D, dtype=torch.float, device=device)
    model.active_sh_degree = model.max_sh_degree


def load_standard_ply(model, path, device="cuda"):
    """render_video_from_ply.py:229-275: a fused / standard 3DGS PLY without filter_3D -> all parameters, filter 1.0."""
    table = _vertex_table(path)
    xyz, opacities, f_dc, f_extra, scales, rots = _parse_gaussians(table, model.max_sh_degree)

    def param(a, transpose=False):
        t = torch.tensor(a, dtype=torch.float, device=device)
        if transpose:
            t = t.transpose(1, 2).contiguous()
        return torch.nn.Parameter(t.requires_grad_(True))
    model._xyz = param(xyz)
    model._features_dc = param(f_dc, True)
    model._features_rest = param(f_extra, True)
    model._opacity = param(opacities)
    model._scaling = param(scales)
    model._rotation = param(rots)
    model.filter_3D = torch.tensor(np.ones((xyz.shape[0], 1)), dtype=torch.float, device=device)
    model.active_sh_degree = model.max_sh_degree


def detect_sh_degree(path):
    """render_video_from_ply.py:163-189."""
    table = _vertex_table(path)
    n_rest = len([n for n in table.dtype.names if n.startswith("f_rest_")])
    return 0 if n_rest == 0 else int(np.sqrt((n_rest / 3) + 1)) - 1


def merge_fused_plys(paths, offsets=None, out_path=None):
    """Concatenate per-scene fused PLYs (save_fused_ply output: no filter_3D column) into one joint scene, each
    shifted by its world offset [x,y,z] (the reference has no multi-scene merge; the joint scene is what
    `sfgs.shard.render_joint` renders band-sharded). All files must have the same columns (same SH degree).
    Returns the merged vertex table; writes it when out_path is given."""
    tables = [_vertex_table(p) for p in paths]
    if not tables:
        raise ValueError("no input files")
    names = tables[0].dtype.names
    for p, t in zip(paths, tables):
        if t.dtype.names != names:
            raise ValueError(f"{p}: columns differ from {paths[0]} (different SH degree or filter_3D present?)")
    if offsets is None:
        offsets = [(0.0, 0.0, 0.0)] * len(tables)
    if len(offsets) != len(tables):
        raise ValueError("one offset per file expected")
    merged = np.concatenate(tables)
    row = 0
    for t, off in zip(tables, offsets):
        for axis, name in enumerate(("x", "y", "z")):
            merged[name][row:row + len(t)] += np.float32(off[axis])
        row += len(t)
    if out_path is not None:
        write_ply(out_path, merged)
    return merged


def fetch_ply(path):
    """scene/dataset_readers.py:126-132 -> (points [N,3], colors [N,3] in 0..1, normals [N,3])."""
    v = PlyData.read(path)["vertex"]
    positions = np.vstack([v["x"], v["y"], v["z"]]).T
    colors = np.vstack([v["red"], v["green"], v["blue"]]).T / 255.0
    normals = np.vstack([v["nx"], v["ny"], v["nz"]]).T
    return positions, colors, normals


def store_ply(path, xyz, rgb):
    """scene/dataset_readers.py:134-149: float32 positions, zero normals, uint8 colours (C-style truncation)."""
    table = np.empty(xyz.shape[0], dtype=[("x", "f4"), ("y", "f4"), ("z", "f4"), ("nx", "f4"), ("ny", "f4"),
                                          ("nz", "f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    attributes = np.concatenate((xyz, np.zeros_like(xyz), rgb), axis=1)
    for i, n in enumerate(table.dtype.names):
        table[n] = attributes[:, i]
    write_ply(path, table)


_ORIG = {}


def install(gaussian_model_cls):
    if gaussian_model_cls in _ORIG:
        return
    _ORIG[gaussian_model_cls] = {k: getattr(gaussian_model_cls, k, None) for k in ("save_ply", "save_fused_ply", "load_ply")}
    gaussian_model_cls.save_ply = save_ply
    gaussian_model_cls.save_fused_ply = save_fused_ply
    gaussian_model_cls.load_ply = load_ply


def uninstall(gaussian_model_cls):
    for k, v in (_ORIG.pop(gaussian_model_cls, None) or {}).items():
        if v is None:
            delattr(gaussian_model_cls, k)
        else:
            setattr(gaussian_model_cls, k, v)
