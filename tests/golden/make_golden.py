#!/usr/bin/env python
"""Generates tests/golden/reference_helpers.npz by IMPORTING the real reference (read-only, from
/root/reference) and running its own Python on seeded inputs. The reference cannot travel to the GPU box,
so the vectors are committed; nothing under tests/ reads /root/reference at run time.

What the reference can pin (its rasterizer is an un-vendored CUDA submodule): every convention
the rasterizer's inputs/outputs obey --
  * camera tensors                 scene/cameras.py:56-79 (Camera.__init__), utils/graphics_utils.py:38-126
  * quaternion -> R, Sigma packing utils/general_utils.py:64-110 (build_rotation, build_scaling_rotation,
                                   strip_symmetric) as used by scene/gaussian_model.py:75-79
  * SH -> RGB                      utils/sh_utils.py:57-112 (eval_sh) + the +0.5 / clamp of
                                   gaussian_renderer/__init__.py:116-117
  * pixel / focal convention       scene/gaussian_model.py:279-286 (projection used by compute_3D_filter)
  * SSIM forward + autograd grad   utils/loss_utils.py:23-63
The reference hard-codes device="cuda"; this script redirects those allocations to the CPU (monkey-patch
below) -- the arithmetic executed is the reference's own.

usage: python tests/golden/make_golden.py   (only in the authoring container)
"""
import importlib.util
import math
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _cpu_redirect():
    for name in ("zeros", "ones", "empty", "tensor", "eye"):
        orig = getattr(torch, name)

        def wrapped(*a, __orig=orig, **k):
            if str(k.get("device", "")).startswith("cuda"):
                k["device"] = "cpu"
            return __orig(*a, **k)
        setattr(torch, name, wrapped)
    torch.Tensor.cuda = lambda self, *a, **k: self


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


OPT_GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "appearance_embeddings", "embeddings",
              "appearance_mlp")


def make_optimizer_golden(gm):
    """the REAL GaussianModel drives the REAL torch.optim.Adam through training_setup ->
    optimizer.step x3 -> densification_postfix -> prune_points -> step x2 (one with missing grads) ->
    replace_tensor_to_optimizer -> step. Records every input and the final parameters / moments / step counts
    (scene/gaussian_model.py:350-392,549-645; train.py:339)."""
    import types
    from torch import nn
    g = torch.Generator().manual_seed(4321)
    torch.manual_seed(99)  # EmbeddingModel init + appearance_embeddings.normal_ use the global generator
    orig_to = nn.Module.to
    nn.Module.to = lambda self, *a, **k: self if (a and str(a[0]).startswith("cuda")) else orig_to(self, *a, **k)
    try:
        m = gm.GaussianModel(1, appearance_enabled=True, appearance_n_fourier_freqs=4, appearance_embedding_dim=32)
    finally:
        nn.Module.to = orig_to
    n = 301  # odd on purpose: ragged tails, unaligned views after surgery
    rnd = lambda *shape: torch.randn(*shape, generator=g)
    m._xyz = nn.Parameter(rnd(n, 3) * 5)
    m._features_dc = nn.Parameter(rnd(n, 1, 3))
    m._features_rest = nn.Parameter(rnd(n, 3, 3) * 0.1)
    m._opacity = nn.Parameter(rnd(n, 1))
    m._scaling = nn.Parameter(rnd(n, 3) - 2)
    m._rotation = nn.Parameter(rnd(n, 4))
    m._embeddings = nn.Parameter(rnd(n, 24))
    m.max_radii2D = torch.zeros(n)
    m.spatial_lr_scale = 3.5
    args = types.SimpleNamespace(percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016,
                                 position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=0.0025,
                                 opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001, embedding_lr=0.005,
                                 appearance_embedding_lr=0.001, appearance_embedding_regularization=0.01,
                                 appearance_mlp_lr=0.0005, idu_position_lr_max_steps=10000)
    m.training_setup(args, num_train_cameras=5, from_scratch=True)
    assert type(m.optimizer) is torch.optim.Adam
    out = {}

    def params_of(name):
        return [grp for grp in m.optimizer.param_groups if grp["name"] == name][0]["params"]

    for name in OPT_GROUPS:
        grp = [q for q in m.optimizer.param_groups if q["name"] == name][0]
        out[f"opt_lr_{name}"] = np.float64(grp["lr"])
        out[f"opt_wd_{name}"] = np.float64(grp["weight_decay"])
        for i, prm in enumerate(grp["params"]):
            out[f"opt_init_{name}_{i}"] = prm.detach().numpy().copy()
    out["opt_eps"] = np.float64(m.optimizer.defaults["eps"])
    out["opt_betas"] = np.array(m.optimizer.defaults["betas"], np.float64)

    def do_step(k, iteration, skip=()):
        out[f"opt_s{k}_xyzlr"] = np.float64(m.update_learning_rate(iteration))
        for name in OPT_GROUPS:
            for i, prm in enumerate(params_of(name)):
                if name in skip:
                    prm.grad = None
                    continue
                scale = 10.0 ** float(torch.randint(-6, 1, (1,), generator=g))  # gradients span many magnitudes
                prm.grad = rnd(*prm.shape) * scale
                if name == "opacity":
                    prm.grad[::7] = 0.0  # exact zeros: invisible Gaussians
                out[f"opt_s{k}_g_{name}_{i}"] = prm.grad.numpy().copy()
        m.optimizer.step()
        m.optimizer.zero_grad(set_to_none=True)

    for k in range(3):
        do_step(k, 1 + k)
    # densification_postfix: 40 new points
    n_new = 40
    new = dict(xyz=rnd(n_new, 3), f_dc=rnd(n_new, 1, 3), f_rest=rnd(n_new, 3, 3), opacity=rnd(n_new, 1),
               scaling=rnd(n_new, 3), rotation=rnd(n_new, 4), embeddings=rnd(n_new, 24))
    for name, t in new.items():
        out[f"opt_cat_{name}"] = t.numpy().copy()
    m.densification_postfix(new["xyz"], new["f_dc"], new["f_rest"], new["opacity"], new["scaling"], new["rotation"],
                            new["embeddings"])
    mask = torch.rand(n + n_new, generator=g) < 0.3
    out["opt_prune_mask"] = mask.numpy().copy()
    m.prune_points(mask)
    do_step(3, 200, skip=("embeddings", "appearance_mlp"))
    do_step(4, 201)
    new_op = rnd(*m._opacity.shape)
    out["opt_replace_opacity"] = new_op.numpy().copy()
    m._opacity = m.replace_tensor_to_optimizer(new_op, "opacity")["opacity"]
    do_step(5, 202)
    for name in OPT_GROUPS:
        for i, prm in enumerate(params_of(name)):
            st = m.optimizer.state[prm]
            out[f"opt_final_{name}_{i}"] = prm.detach().numpy().copy()
            out[f"opt_final_m_{name}_{i}"] = st["exp_avg"].numpy().copy()
            out[f"opt_final_v_{name}_{i}"] = st["exp_avg_sq"].numpy().copy()
            out[f"opt_final_step_{name}_{i}"] = np.float64(float(st["step"]))
    path = os.path.join(HERE, "reference_optimizer.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


def make_sh_grad_golden(eval_sh):
    """the REAL eval_sh (utils/sh_utils.py:57-112) forward + autograd gradients w.r.t. the coefficients
    and the directions, degrees 0-3, with more stored coefficients than the active degree uses (as in render())."""
    g = torch.Generator().manual_seed(777)
    out = {}
    n, K = 96, 16
    for deg in range(4):
        sh = torch.randn(n, 3, K, generator=g).requires_grad_(True)
        d = torch.randn(n, 3, generator=g)
        dirs = (d / d.norm(dim=1, keepdim=True)).detach().requires_grad_(True)
        w = torch.randn(n, 3, generator=g)
        val = eval_sh(deg, sh, dirs)
        (val * w).sum().backward()
        for k, v in dict(sh=sh.detach(), dirs=dirs.detach(), w=w, out=val.detach(), g_sh=sh.grad,
                         g_dirs=dirs.grad if dirs.grad is not None else torch.zeros_like(dirs)).items():
            out[f"shg{deg}_{k}"] = v.numpy().copy()
    path = os.path.join(HERE, "reference_sh_grad.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


def _ref_functions(path, names, namespace):
    """exec only the named top-level functions of a reference file (its module-level imports are too heavy)."""
    import ast
    tree = ast.parse(open(path).read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), namespace)
    return [namespace[n] for n in names]


def make_ply_golden(gm):
    """the REAL save_ply / save_fused_ply / load_ply (scene/gaussian_model.py:418-547),
    load_standard_ply / detect_sh_degree_from_ply (render_video_from_ply.py:163-275) and storePly / fetchPly
    (scene/dataset_readers.py:126-148) run on top of sfgs.ply's plyfile stand-in (plyfile itself is not installed
    here): pins the column names / order / transposes / dtypes the reference produces and expects."""
    import tempfile
    import types
    sys.path.insert(0, os.path.join(HERE, "..", "..", "skyfall-gs_amd"))
    from sfgs import ply as sply
    assert gm.PlyData is object  # the stub main() installed for the heavy import
    gm.PlyData, gm.PlyElement = sply.PlyData, sply.PlyElement
    sys.modules["plyfile"].PlyData, sys.modules["plyfile"].PlyElement = sply.PlyData, sply.PlyElement
    g = torch.Generator().manual_seed(2468)
    out = {}
    m = gm.GaussianModel(1, False, 4, 32)
    n = 37
    m._xyz = torch.randn(n, 3, generator=g) * 4
    m._features_dc = torch.randn(n, 1, 3, generator=g)
    m._features_rest = torch.randn(n, 3, 3, generator=g) * 0.2
    m._opacity = torch.randn(n, 1, generator=g)
    m._scaling = torch.randn(n, 3, generator=g) - 1.5
    m._rotation = torch.randn(n, 4, generator=g)
    m.filter_3D = torch.exp(torch.randn(n, 1, generator=g) - 2.5)
    for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "filter_3D"):
        out["ply_in" + k] = getattr(m, k).numpy().copy()
    with tempfile.TemporaryDirectory() as td:
        p1, p2, p3 = (os.path.join(td, "sub", f) for f in ("a.ply", "fused.ply", "pts.ply"))
        m.save_ply(p1)
        m.save_fused_ply(p2)
        out["ply_save_bytes"] = np.fromfile(p1, dtype=np.uint8)
        out["ply_fused_bytes"] = np.fromfile(p2, dtype=np.uint8)
        m2 = gm.GaussianModel(1, False, 4, 32)
        m2.load_ply(p1)
        out["ply_load_filter_3D"] = m2.filter_3D.numpy().copy()
        out["ply_load_active_sh_degree"] = np.int64(m2.active_sh_degree)
        assert m2._xyz.numel() == 0  # the reference's load_ply leaves the parameters untouched (assignments commented out)
        ns = {"np": np, "torch": torch, "GaussianModel": gm.GaussianModel}
        load_standard_ply, detect = _ref_functions(os.path.join(REF, "render_video_from_ply.py"),
                                                   ["load_standard_ply", "detect_sh_degree_from_ply"], ns)
        out["ply_detect_sh_degree"] = np.int64(detect(p2))
        m3 = gm.GaussianModel(1, False, 4, 32)
        load_standard_ply(m3, p2)
        for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "filter_3D"):
            out["ply_std" + k] = getattr(m3, k).detach().numpy().copy()
        ns2 = {"np": np, "PlyData": sply.PlyData, "PlyElement": sply.PlyElement,
               "BasicPointCloud": lambda points, colors, normals: (points, colors, normals)}
        fetchPly, storePly = _ref_functions(os.path.join(REF, "scene", "dataset_readers.py"), ["fetchPly", "storePly"], ns2)
        xyz = (torch.randn(29, 3, generator=g) * 10).numpy()
        rgb = (torch.rand(29, 3, generator=g) * 255).numpy()
        storePly(p3, xyz, rgb)
        out["ply_store_xyz"], out["ply_store_rgb"] = xyz, rgb
        out["ply_store_bytes"] = np.fromfile(p3, dtype=np.uint8)
        pts, cols, nrm = fetchPly(p3)
        out["ply_fetch_points"], out["ply_fetch_colors"], out["ply_fetch_normals"] = pts, cols, nrm
    path = os.path.join(HERE, "reference_ply.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


def main():
    _cpu_redirect()
    sys.path.insert(0, REF)
    from utils import general_utils as gu
    from utils import graphics_utils as gfx  # noqa: F401  (imported by scene/cameras.py)
    from utils.loss_utils import ssim
    from utils.sh_utils import eval_sh
    cameras = _load(os.path.join(REF, "scene", "cameras.py"), "ref_cameras")  # avoids scene/__init__'s heavy imports

    g = torch.Generator().manual_seed(1234)
    out = {}

    # ---- cameras --------------------------------------------------------------------------------------
    cams, cam_objs = [], []
    for i in range(6):
        q = torch.randn(4, generator=g, dtype=torch.float64)
        q = q / q.norm()
        w, x, y, z = q.tolist()
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        T = (torch.randn(3, generator=g, dtype=torch.float64) * 3).numpy()
        W, H = [(800, 800), (1920, 1080), (1024, 1024), (130, 77), (960, 540), (2560, 1440)][i]
        fovx = math.radians([60.0, 60.0, 20.0, 45.0, 20.0, 33.0][i])
        fovy = 2 * math.atan(H / (2 * (W / (2 * math.tan(fovx / 2)))))
        cx, cy = [(0.0, 0.0), (0.0, 0.0), (0.01, -0.02), (0.1, 0.05), (0.0, 0.0), (-0.03, 0.04)][i]
        cam = cameras.Camera(colmap_id=i, R=R, T=T, FoVx=fovx, FoVy=fovy, cx=cx, cy=cy,
                             image=torch.zeros(3, H, W), gt_alpha_mask=None, image_name=str(i), uid=i,
                             data_device="cpu")
        cam_objs.append(cam)
        cams.append(dict(R=R, T=T, W=W, H=H, fovx=fovx, fovy=fovy, cx=cx, cy=cy,
                         view=cam.world_view_transform.numpy(), proj=cam.projection_matrix.numpy(),
                         full=cam.full_proj_transform.numpy(), center=cam.camera_center.numpy(),
                         focal_x=cam.focal_x, focal_y=cam.focal_y))
    for k in cams[0]:
        out["cam_" + k] = np.stack([np.asarray(c[k]) for c in cams])

    # ---- pixel convention: the reference's Python projection (gaussian_model.py:279-286) -------------------
    c = cams[3]
    pts = torch.randn(64, 3, generator=g, dtype=torch.float64) * 2 + torch.tensor([0.0, 0.0, 6.0], dtype=torch.float64)
    Rt = torch.tensor(c["R"], dtype=torch.float64)
    Tt = torch.tensor(c["T"], dtype=torch.float64)
    world = (pts - Tt[None]) @ Rt.T          # so that world @ R + T == pts (camera space)
    xyz_cam = world @ Rt + Tt[None]
    x, y, z = xyz_cam[:, 0], xyz_cam[:, 1], xyz_cam[:, 2].clamp(min=0.001)
    cx_ori = c["cx"] / 2 * c["W"] + c["W"] / 2
    cy_ori = c["cy"] / 2 * c["H"] + c["H"] / 2
    out["pix_world"] = world.numpy()
    out["pix_x"] = (x / z * c["focal_x"] + cx_ori).numpy()
    out["pix_y"] = (y / z * c["focal_y"] + cy_ori).numpy()
    out["pix_z"] = xyz_cam[:, 2].numpy()

    # ---- rotation / covariance ------------------------------------------------------------------------------
    quats = torch.randn(256, 4, generator=g)
    scales = torch.exp(torch.randn(256, 3, generator=g))
    L = gu.build_scaling_rotation(1.7 * scales, quats)           # scene/gaussian_model.py:75-79 with modifier 1.7
    cov = gu.strip_symmetric(L @ L.transpose(1, 2))
    out["rot_quats"] = quats.numpy()
    out["rot_R"] = gu.build_rotation(quats).numpy()
    out["cov_scales"] = scales.numpy()
    out["cov_modifier"] = np.float32(1.7)
    out["cov_packed"] = cov.numpy()

    # ---- spherical harmonics -----------------------------------------------------------------------------------
    sh = torch.randn(200, 3, 16, generator=g)
    dirs = torch.randn(200, 3, generator=g)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    out["sh_coeffs"] = sh.numpy()
    out["sh_dirs"] = dirs.numpy()
    for deg in range(4):
        K = (deg + 1) ** 2
        rgb = eval_sh(deg, sh[:, :, :K], dirs)
        out[f"sh_rgb_deg{deg}"] = torch.clamp_min(rgb + 0.5, 0.0).numpy()

    # ---- SSIM ----------------------------------------------------------------------------------------------------
    for tag, (B, C, H, W) in {"a": (1, 3, 37, 53), "b": (2, 1, 16, 11), "c": (1, 3, 64, 96)}.items():
        img1 = torch.rand(B, C, H, W, generator=g).requires_grad_(True)
        img2 = (img1.detach() + 0.1 * torch.randn(B, C, H, W, generator=g)).clamp(0, 1)
        val = ssim(img1, img2)
        val.backward()
        out[f"ssim_{tag}_img1"] = img1.detach().numpy()
        out[f"ssim_{tag}_img2"] = img2.numpy()
        out[f"ssim_{tag}_value"] = np.float64(val.item())
        out[f"ssim_{tag}_grad"] = img1.grad.numpy()

    # ---- fused pre-pass: the REAL GaussianModel getters --------------------------------------
    import types
    for name in ("plyfile", "simple_knn", "simple_knn._C"):  # heavy / CUDA-only imports of scene/gaussian_model.py
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
    sys.modules["simple_knn._C"].distCUDA2 = None
    gm = _load(os.path.join(REF, "scene", "gaussian_model.py"), "ref_gaussian_model")
    for tag, fdtype in (("f64", torch.float64), ("f32", torch.float32)):
        m = gm.GaussianModel.__new__(gm.GaussianModel)
        m.setup_functions()
        n = 512
        m._scaling = (torch.randn(n, 3, generator=g) * 1.5 - 2.0).requires_grad_(True)
        m._opacity = (torch.randn(n, 1, generator=g) * 2.0).requires_grad_(True)
        m._rotation = torch.randn(n, 4, generator=g).requires_grad_(True)
        m.filter_3D = torch.exp(torch.randn(n, 1, generator=g, dtype=torch.float64) - 3.0).to(fdtype)
        sc, op, ro = m.get_scaling_with_3D_filter, m.get_opacity_with_3D_filter, m.get_rotation
        w1, w2, w3 = torch.randn(n, 3, generator=g), torch.randn(n, 1, generator=g), torch.randn(n, 4, generator=g)
        ((sc.float() * w1).sum() + (op.float() * w2).sum() + (ro * w3).sum()).backward()   # render()'s .float() casts
        for k, v in dict(scaling=m._scaling.detach(), opacity=m._opacity.detach(), rotation=m._rotation.detach(),
                         filter=m.filter_3D, out_scales=sc.detach().float(), out_opacity=op.detach().float(),
                         out_rotation=ro.detach(), w_scales=w1, w_opacity=w2, w_rotation=w3,
                         g_scaling=m._scaling.grad, g_opacity=m._opacity.grad, g_rotation=m._rotation.grad).items():
            out[f"prepass_{tag}_{k}"] = v.numpy()

    # ---- compute_3D_filter: the REAL GaussianModel method over the six cameras above ----------
    m = gm.GaussianModel.__new__(gm.GaussianModel)
    m._xyz = torch.randn(3000, 3, generator=g) * 6.0
    m._xyz[:5] = 1e4  # a few points no camera sees
    m.compute_3D_filter(cam_objs)
    out["filter3d_xyz"] = m._xyz.numpy()
    out["filter3d_out"] = m.filter_3D.numpy()
    assert m.filter_3D.dtype == torch.float64 and m.filter_3D.shape == (3000, 1)

    # ---- add_densification_stats: the REAL GaussianModel method, two consecutive steps ---------
    m = gm.GaussianModel.__new__(gm.GaussianModel)
    n = 1000
    for name in ("xyz_gradient_accum", "xyz_gradient_accum_abs", "xyz_gradient_accum_abs_max", "denom"):
        setattr(m, name, torch.zeros(n, 1))
    for step in range(2):
        vs = torch.zeros(n, 3, requires_grad=True)
        vs.grad = torch.randn(n, 3, generator=g) * torch.tensor([1.0, 1.0, 1.0])
        vs.grad[:, 2].abs_()
        filt = torch.rand(n, generator=g) > 0.3
        m.add_densification_stats(vs, filt)
        out[f"dstats_grad{step}"] = vs.grad.numpy().copy()
        out[f"dstats_filter{step}"] = filt.numpy().copy()
    for name in ("xyz_gradient_accum", "xyz_gradient_accum_abs", "xyz_gradient_accum_abs_max", "denom"):
        out["dstats_" + name] = getattr(m, name).numpy().copy()

    path = os.path.join(HERE, "reference_helpers.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")
    make_optimizer_golden(gm)
    make_sh_grad_golden(eval_sh)
    make_ply_golden(gm)


if __name__ == "__main__":
    main()
