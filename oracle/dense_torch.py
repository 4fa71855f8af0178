"""Dense float64 PyTorch oracle with autograd (TEST INFRASTRUCTURE, not product code).

A second, independent restatement of SURVEY Appendix A used to validate the C oracle
(oracle/sfgs_oracle.c): the forward to float32 round-off and the hand-derived backward against
autograd. It evaluates every (Gaussian, pixel) pair densely, so it is only usable for a few hundred
Gaussians on small images.

Reference anchors (same as the C oracle): conventions scene/cameras.py:62-73,
utils/graphics_utils.py:106-126; quaternion->R utils/general_utils.py:78-99; SH utils/sh_utils.py:57-112;
near plane scene/gaussian_model.py:276. PARITY UNPINNED for everything marked [UPSTREAM] in SURVEY App. A.
"""
import math

import torch

TILE = 16
C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def _eval_sh(deg, sh, dirs):
    """sh [N,3,M] , dirs [N,3] -> [N,3]; restates utils/sh_utils.py:74-100."""
    result = C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        result = result - C1 * y * sh[..., 1] + C1 * z * sh[..., 2] - C1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            result = (result + C2[0] * xy * sh[..., 4] + C2[1] * yz * sh[..., 5]
                      + C2[2] * (2.0 * zz - xx - yy) * sh[..., 6] + C2[3] * xz * sh[..., 7]
                      + C2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                result = (result + C3[0] * y * (3 * xx - yy) * sh[..., 9] + C3[1] * xy * z * sh[..., 10]
                          + C3[2] * y * (4 * zz - xx - yy) * sh[..., 11]
                          + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12]
                          + C3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + C3[5] * z * (xx - yy) * sh[..., 14]
                          + C3[6] * x * (xx - 3 * yy) * sh[..., 15])
    return result


def _rotation(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.view(-1, 3, 3)


def preprocess(frame, means3D, scales, rotations, opacities, colors_precomp=None, shs=None, means2D=None,
               dtype=torch.float64):
    """SURVEY A.2 for all Gaussians at once (autograd-tracked). Returns a dict of per-Gaussian tensors."""
    H, W = frame["H"], frame["W"]
    V = frame["view"].to(dtype)
    PM = frame["proj"].to(dtype)
    bg = frame["bg"].to(dtype)
    ks = frame["kernel_size"]
    N = means3D.shape[0]
    p = means3D.to(dtype)
    ones = torch.ones(N, 1, dtype=dtype)
    ph = torch.cat([p, ones], dim=1)
    t = ph @ V  # [N,4]
    tx, ty, tz = t[:, 0], t[:, 1], t[:, 2]
    hom = ph @ PM
    pw = 1.0 / (hom[:, 3] + 1e-7)
    ndc = hom[:, :2] * pw[:, None]
    if means2D is not None:
        ndc = ndc + means2D[:, :2].to(dtype)  # dummy leaf: its grad is dL/dmean2D in NDC units

    R = _rotation(rotations.to(dtype))
    S = frame["scale_modifier"] * scales.to(dtype)
    M = R * S[:, None, :]
    Sig = M @ M.transpose(1, 2)

    limx, limy = 1.3 * frame["tanfovx"], 1.3 * frame["tanfovy"]
    txtz, tytz = tx / tz, ty / tz
    cl_x = (txtz < -limx) | (txtz > limx)
    cl_y = (tytz < -limy) | (tytz > limy)
    # [UPSTREAM] clamped coordinate is treated as a constant in the derivative
    ux = torch.where(cl_x, (txtz.clamp(-limx, limx) * tz).detach(), tx)
    uy = torch.where(cl_y, (tytz.clamp(-limy, limy) * tz).detach(), ty)
    fx = W / (2.0 * frame["tanfovx"])
    fy = H / (2.0 * frame["tanfovy"])
    zeros = torch.zeros_like(tz)
    J = torch.stack([torch.stack([fx / tz, zeros, -(fx * ux) / (tz * tz)], -1),
                     torch.stack([zeros, fy / tz, -(fy * uy) / (tz * tz)], -1)], 1)  # [N,2,3]
    W2C = V[:3, :3].t()  # W2C[a][b] = V[b][a]
    Tm = J @ W2C  # [N,2,3]
    cov = Tm @ Sig @ Tm.transpose(1, 2)
    a0, b0, c0 = cov[:, 0, 0], cov[:, 0, 1], cov[:, 1, 1]
    det0 = (a0 * c0 - b0 * b0).clamp_min(1e-6)
    a, c, b = a0 + ks, c0 + ks, b0
    det1 = (a * c - b * b).clamp_min(1e-6)
    coef = torch.sqrt(det0 / (det1 + 1e-6) + 1e-6)
    coef = torch.where((det0 <= 1e-6) | (det1 <= 1e-6), torch.zeros_like(coef), coef)
    det = a * c - b * b
    cA, cB, cC = c / det, -b / det, a / det
    mid = 0.5 * (a + c)
    disc = torch.sqrt((mid * mid - det).clamp_min(0.1))
    lam = torch.maximum(mid + disc, mid - disc)
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    mx = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    my = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5

    TX, TY = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE

    def trunc_clamp(v, hi):
        return torch.trunc(v).clamp(0, hi).to(torch.int64)
    rminx = trunc_clamp((mx.detach() - radius) / TILE, TX)
    rminy = trunc_clamp((my.detach() - radius) / TILE, TY)
    rmaxx = trunc_clamp((mx.detach() + radius + TILE - 1) / TILE, TX)
    rmaxy = trunc_clamp((my.detach() + radius + TILE - 1) / TILE, TY)
    visible = (tz > 0.2) & (det != 0) & ((rmaxx - rminx) * (rmaxy - rminy) > 0)
    radii = torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32)

    if colors_precomp is not None:
        rgb = colors_precomp.to(dtype)
    else:
        d = p - frame["campos"].to(dtype)[None]
        d = d / d.norm(dim=1, keepdim=True)
        sh = shs.to(dtype).transpose(1, 2)  # [N,3,M]
        rgb = torch.clamp_min(_eval_sh(frame["sh_degree"], sh, d) + 0.5, 0.0)

    op = opacities.to(dtype).reshape(-1) * coef

    return dict(H=H, W=W, bg=bg, mx=mx, my=my, cA=cA, cB=cB, cC=cC, op=op, rgb=rgb, tz=tz, radius=radius, radii=radii,
                visible=visible, rminx=rminx, rminy=rminy, rmaxx=rmaxx, rmaxy=rmaxy, N=N, TX=TX, TY=TY)


def render_dense(frame, means3D, scales, rotations, opacities, colors_precomp=None, shs=None,
                 means2D=None, dtype=torch.float64):
    """frame: dict(H,W,tanfovx,tanfovy,kernel_size,scale_modifier,sh_degree,view,proj,campos,bg,
    subpix(optional [H,W,2]),depth_mode). Returns color[3,H,W], depth[1,H,W], alpha[1,H,W], radii[N]."""
    P = preprocess(frame, means3D, scales, rotations, opacities, colors_precomp, shs, means2D, dtype)
    H, W, bg, N = P["H"], P["W"], P["bg"], P["N"]
    mx, my, cA, cB, cC, op, rgb, tz = (P[k] for k in ("mx", "my", "cA", "cB", "cC", "op", "rgb", "tz"))
    visible, radii = P["visible"], P["radii"]
    rminx, rminy, rmaxx, rmaxy = P["rminx"], P["rminy"], P["rmaxx"], P["rmaxy"]

    # order: ascending (float32 depth bits, index) -- A.3
    depth32 = tz.detach().to(torch.float32)
    order = sorted(range(N), key=lambda i: (depth32[i].item(), i))

    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    sx = xs.to(dtype)
    sy = ys.to(dtype)
    if frame.get("subpix") is not None:
        sx = sx + frame["subpix"][..., 0].to(dtype)
        sy = sy + frame["subpix"][..., 1].to(dtype)
    tile_x = xs // TILE
    tile_y = ys // TILE

    Tr = torch.ones(H, W, dtype=dtype)
    Cacc = torch.zeros(3, H, W, dtype=dtype)
    Dacc = torch.zeros(H, W, dtype=dtype)
    done = torch.zeros(H, W, dtype=torch.bool)
    for i in order:
        if not bool(visible[i]):
            continue
        in_rect = (tile_x >= rminx[i]) & (tile_x < rmaxx[i]) & (tile_y >= rminy[i]) & (tile_y < rmaxy[i])
        if not bool(in_rect.any()):
            continue
        dx = mx[i] - sx
        dy = my[i] - sy
        power = -0.5 * (cA[i] * dx * dx + cC[i] * dy * dy) - cB[i] * dx * dy
        raw = op[i] * torch.exp(power)
        # min(0.99, .) with the clamp ignored in the derivative [UPSTREAM]
        alpha = torch.where(raw > 0.99, 0.99 + (raw - raw.detach()), raw)
        ok = in_rect & (~done) & (power <= 0) & (alpha >= 1.0 / 255.0)
        test_T = Tr * (1 - alpha)
        stop = ok & (test_T < 0.0001)
        done = done | stop
        use = ok & (~stop)
        w = torch.where(use, alpha * Tr, torch.zeros_like(Tr))
        Cacc = Cacc + rgb[i][:, None, None] * w[None]
        Dacc = Dacc + tz[i] * w
        Tr = torch.where(use, test_T, Tr)
    color = Cacc + Tr[None] * bg[:, None, None]
    alpha_out = (1 - Tr)[None]
    if frame.get("depth_mode", 0) == 0:
        a_ = 1 - Tr
        safe = torch.where(a_ > 0, a_, torch.ones_like(a_))  # keeps autograd finite where nothing hit
        depth = torch.where(a_ > 0, Dacc / safe, torch.full_like(a_, float("nan")))[None]
    else:
        depth = Dacc[None]
    return color, depth, alpha_out, radii
