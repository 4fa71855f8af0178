"""Seeded synthetic scenes of BASELINE.md section 4 / SURVEY 8(d) (host logic, CPU tensors).

Inputs are generated on the CPU with torch.Generator().manual_seed(seed) in float32 so that every
backend (oracle, HIP) sees bit-identical data; callers move them to the device.
"""
import math

import numpy as np
import torch

from .camera import fovy_from_fovx, make_frame


def _unit_quats(n, gen):
    q = torch.randn(n, 4, generator=gen)
    return q / q.norm(dim=1, keepdim=True)


def scene(n, W, H, seed=0, zrange=(250.0, 350.0), scale_range=(0.05, 0.6), fovx_deg=60.0, mode="precomp",
          sh_degree=1, xy_fill=1.0, pitch_deg=0.0, jitter=False, opacity_range=(0.05, 0.95), kernel_size=0.1):
    """cfg 2-style scene: pinhole at the origin, COLMAP axes, z ~ U(zrange), x,y fill the frustum,
    scales = exp(U(ln lo, ln hi)) per axis, unit quaternions, opacity ~ U(0.05,0.95).
    mode 'precomp' -> colors_precomp ~ U(0,1)^3 ; mode 'sh' -> shs deg `sh_degree`."""
    gen = torch.Generator().manual_seed(seed)
    fovx = math.radians(fovx_deg)
    fovy = fovy_from_fovx(fovx, W, H)
    z = torch.empty(n).uniform_(zrange[0], zrange[1], generator=gen)
    u = torch.empty(n, 2).uniform_(-1.0, 1.0, generator=gen) * xy_fill
    x = u[:, 0] * z * math.tan(fovx / 2)
    y = u[:, 1] * z * math.tan(fovy / 2)
    means = torch.stack([x, y, z], 1).contiguous()
    lo, hi = math.log(scale_range[0]), math.log(scale_range[1])
    scales = torch.exp(torch.empty(n, 3).uniform_(lo, hi, generator=gen))
    rots = _unit_quats(n, gen)
    opac = torch.empty(n, 1).uniform_(opacity_range[0], opacity_range[1], generator=gen)
    out = dict(means3D=means, scales=scales, rotations=rots, opacities=opac, colors_precomp=None, shs=None)
    if mode == "precomp":
        out["colors_precomp"] = torch.rand(n, 3, generator=gen)
    else:
        K = (sh_degree + 1) ** 2
        shs = torch.randn(n, K, 3, generator=gen)
        shs[:, 1:] *= 0.3
        out["shs"] = shs.contiguous()
    R = np.eye(3)
    if pitch_deg:
        # rotate the camera about its x axis (low-elevation variant): world points stay in front
        a = math.radians(pitch_deg)
        R = np.array([[1, 0, 0], [0, math.cos(a), -math.sin(a)], [0, math.sin(a), math.cos(a)]])
        out["means3D"] = (means @ torch.tensor(R.T, dtype=torch.float32)).contiguous()  # camera -> world
    subpix = None
    if jitter:
        subpix = (torch.rand(H, W, 2, generator=gen) - 0.5).contiguous()
    frame = make_frame(R, np.zeros(3), fovx, fovy, W, H, kernel_size=kernel_size,
                       sh_degree=sh_degree if mode == "sh" else 0, subpix=subpix)
    return frame, out


def orbit_scene(n, W, H, elevation_deg, seed=0, radius=420.0, extent=170.0, height=25.0, scale_range=(0.05, 0.6),
                fovx_deg=60.0, opacity_range=(0.05, 0.95), kernel_size=0.1):
    """A city-like slab (x, y in [-extent, extent], z in [0, height], z up) seen from an orbit camera that looks at the
    origin from `elevation_deg` above the horizon -- the pseudo-camera geometry of the reference's IDU stage
    (train.py:364-420: elevations 85..45 degrees, 25 in one schedule). colors_precomp mode."""
    gen = torch.Generator().manual_seed(seed)
    fovx = math.radians(fovx_deg)
    fovy = fovy_from_fovx(fovx, W, H)
    xy = torch.empty(n, 2).uniform_(-extent, extent, generator=gen)
    z = torch.empty(n, 1).uniform_(0.0, height, generator=gen)
    lo, hi = math.log(scale_range[0]), math.log(scale_range[1])
    out = dict(means3D=torch.cat([xy, z], 1).contiguous(),
               scales=torch.exp(torch.empty(n, 3).uniform_(lo, hi, generator=gen)), rotations=_unit_quats(n, gen),
               opacities=torch.empty(n, 1).uniform_(opacity_range[0], opacity_range[1], generator=gen),
               colors_precomp=torch.rand(n, 3, generator=gen), shs=None)
    e = math.radians(elevation_deg)
    C = np.array([radius * math.cos(e), 0.0, radius * math.sin(e)])
    f = -C / np.linalg.norm(C)                      # camera z: forward
    r = np.cross(f, np.array([0.0, 0.0, 1.0]))
    r = r / np.linalg.norm(r) if np.linalg.norm(r) > 1e-9 else np.array([0.0, 1.0, 0.0])   # camera x: right
    d = np.cross(f, r)                              # camera y: down
    R = np.stack([r, d, f], 1)                      # camera-to-world rotation (columns = camera axes)
    t = -R.T @ C                                    # world-to-camera translation
    return make_frame(R, t, fovx, fovy, W, H, kernel_size=kernel_size), out


def city_scene(n, W, H, elevation_deg, seed=0, radius=420.0, extent=170.0, n_buildings=300, splat=(0.25, 0.9),
               opacity_range=(0.6, 0.99), fovx_deg=60.0, kernel_size=0.1):
    """Opaque SURFACES instead of a transparent volume: a ground plane and `n_buildings` axis-aligned boxes (footprint
    8..30 m, height 5..45 m), covered with flat, mostly opaque disks (thickness 2 cm, normal = surface normal) -- what a
    trained urban scene looks like to the rasterizer: pixels saturate after a few splats and most of a tile's list lies
    BEHIND the last contributor. Same orbit camera as orbit_scene."""
    gen = torch.Generator().manual_seed(seed)
    frame, _ = orbit_scene(1, W, H, elevation_deg, seed, radius, extent, fovx_deg=fovx_deg, kernel_size=kernel_size)
    bx = torch.empty(n_buildings, 2).uniform_(-extent * 0.9, extent * 0.9, generator=gen)
    bs = torch.empty(n_buildings, 2).uniform_(8.0, 30.0, generator=gen)
    bh = torch.empty(n_buildings).uniform_(5.0, 45.0, generator=gen)
    # surface areas: ground, then per building roof + 4 walls
    areas = [torch.tensor([(2 * extent) ** 2])]
    areas.append(torch.stack([bs[:, 0] * bs[:, 1], bs[:, 0] * bh, bs[:, 0] * bh, bs[:, 1] * bh, bs[:, 1] * bh], 1).reshape(-1))
    areas = torch.cat(areas)
    face = torch.multinomial(areas / areas.sum(), n, replacement=True, generator=gen)
    u = torch.rand(n, 2, generator=gen)
    pos = torch.zeros(n, 3)
    normal_axis = torch.full((n,), 2, dtype=torch.long)      # ground / roofs: normal along z
    g0 = face == 0
    pos[g0, 0] = (u[g0, 0] * 2 - 1) * extent
    pos[g0, 1] = (u[g0, 1] * 2 - 1) * extent
    fb = (face - 1).clamp_min(0)
    b, k = fb // 5, fb % 5
    cx, cy, sx, sy, hh = bx[b, 0], bx[b, 1], bs[b, 0], bs[b, 1], bh[b]
    roof = (~g0) & (k == 0)
    pos[roof] = torch.stack([cx + (u[:, 0] - 0.5) * sx, cy + (u[:, 1] - 0.5) * sy, hh], 1)[roof]
    for kk, (axis, sign) in enumerate([(1, -1.0), (1, 1.0), (0, -1.0), (0, 1.0)], start=1):
        m = (~g0) & (k == kk)
        if axis == 1:   # wall facing -y / +y: spans x and z
            p = torch.stack([cx + (u[:, 0] - 0.5) * sx, cy + sign * 0.5 * sy, u[:, 1] * hh], 1)
        else:           # wall facing -x / +x: spans y and z
            p = torch.stack([cx + sign * 0.5 * sx, cy + (u[:, 0] - 0.5) * sy, u[:, 1] * hh], 1)
        pos[m] = p[m]
        normal_axis[m] = axis
    lo, hi = math.log(splat[0]), math.log(splat[1])
    scales = torch.exp(torch.empty(n, 3).uniform_(lo, hi, generator=gen))
    scales[torch.arange(n), normal_axis] = 0.02              # flat along the surface normal (axis-aligned: identity rotation)
    rots = torch.zeros(n, 4)
    rots[:, 0] = 1.0
    out = dict(means3D=pos.contiguous(), scales=scales, rotations=rots,
               opacities=torch.empty(n, 1).uniform_(opacity_range[0], opacity_range[1], generator=gen),
               colors_precomp=torch.rand(n, 3, generator=gen), shs=None)
    return frame, out


def morton_order(means3D, bits=10):
    """Permutation that sorts points along a Z-curve of their (x / z, y / z) direction, 2^bits cells per axis."""
    d = (means3D[:, :2] / means3D[:, 2:3]).double()
    lo, hi = d.min(0).values, d.max(0).values
    q = ((d - lo) / (hi - lo).clamp_min(1e-30) * (2 ** bits - 1)).long().clamp(0, 2 ** bits - 1)
    key = torch.zeros(means3D.shape[0], dtype=torch.int64)
    for b in range(bits):
        key |= ((q[:, 0] >> b) & 1) << (2 * b)
        key |= ((q[:, 1] >> b) & 1) << (2 * b + 1)
    return torch.argsort(key, stable=True)


def upstream_grads(W, H, seed=0):
    """dL/dimage, dL/ddepth ~ N(0,1)/P."""
    gen = torch.Generator().manual_seed(1000 + seed)
    P = W * H
    return torch.randn(3, H, W, generator=gen) / P, torch.randn(1, H, W, generator=gen) / P


# BASELINE.json configs -----------------------------------------------------------------------
def cfg1(seed=0, n=50_000):
    return scene(n, 800, 800, seed, zrange=(4.0, 8.0), scale_range=(0.005, 0.05))


def cfg2(seed=0, n=2_000_000, W=1920, H=1080):
    return scene(n, W, H, seed)


def cfg4(seed=0, n=5_000_000):
    return scene(n, 2560, 1440, seed, zrange=(500.0, 700.0))
