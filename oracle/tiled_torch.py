"""Tiled PyTorch restatement of the rasterizer with autograd -- TEST INFRASTRUCTURE and bench.py's CPU baseline.

SURVEY 8(d): the reference has no CPU render path (gaussian_renderer/__init__.py:27,38 hard-code "cuda" and the op
is CUDA-only), so "the reference's PyTorch CPU render path" is OUR PyTorch restatement of the same algorithm
(SURVEY Appendix A), timed with torch.set_num_threads(all host cores). oracle/dense_torch.py evaluates every
(Gaussian, pixel) pair and only scales to a few hundred Gaussians; this variant follows the tile structure of the
algorithm so that cfg 1 (50 k Gaussians, 800x800) runs in full:

  preprocess        vectorised over the Gaussians (oracle/dense_torch.preprocess: SURVEY A.2)
  binning + order   duplicates = repeat_interleave over each Gaussian's 16x16-tile rectangle, one argsort by
                    (tile, depth bits, index) (A.3)
  composite         per CHUNK of tiles of similar list length (padded with a null Gaussian): alpha[tiles, L, 256] for
                    the L entries and 256 pixels of every tile; transmittance by an exclusive
                    cumprod along the list; the "stop when T (1 - alpha) < 1e-4, splat not applied" rule as a
                    cumulative mask (A.4). Autograd supplies the backward (A.6), including the upstream convention
                    that the min(0.99, .) clamp is ignored in the derivative.

PARITY UNPINNED like the rest of oracle/ (the rasterizer source is an un-vendored submodule); checked against the C
oracle in tests/test_oracle_cpu.py. Only tests/ and bench.py's cpu_baseline leg import this file.
"""
import torch

from .dense_torch import TILE, preprocess


def render_tiled(frame, means3D, scales, rotations, opacities, colors_precomp=None, shs=None, means2D=None,
                 dtype=torch.float32, tile_subset=None, chunk_pairs=4_000_000):
    """Returns color[3,H,W], depth[1,H,W], alpha[1,H,W], radii[N], stats. tile_subset: optional iterable of tile ids to
    composite (bench.py's bounded sample of a large frame; the other tiles stay at background)."""
    P = preprocess(frame, means3D, scales, rotations, opacities, colors_precomp, shs, means2D, dtype)
    H, W, bg, N, TX, TY = P["H"], P["W"], P["bg"], P["N"], P["TX"], P["TY"]
    vis = P["visible"]
    idx = torch.nonzero(vis).reshape(-1)
    w_t = (P["rmaxx"] - P["rminx"])[idx]
    h_t = (P["rmaxy"] - P["rminy"])[idx]
    cnt = w_t * h_t
    D = int(cnt.sum())
    # one row per (Gaussian, tile) duplicate
    g_of = torch.repeat_interleave(idx, cnt)
    start = torch.cumsum(cnt, 0) - cnt
    local = torch.arange(D) - torch.repeat_interleave(start, cnt)
    wrep = torch.repeat_interleave(w_t, cnt)
    tx = P["rminx"][g_of] + local % wrep
    ty = P["rminy"][g_of] + local // wrep
    tile = ty * TX + tx
    depth_bits = P["tz"].detach().to(torch.float32).view(torch.int32).to(torch.int64)[g_of]
    key = (tile << 32) | depth_bits            # view-space z > 0.2: the bit pattern orders like the float
    order = torch.sort(key, stable=True).indices   # g_of ascends by construction: ties keep ascending index order
    g_sorted, tile_sorted = g_of[order], tile[order]
    bounds = torch.searchsorted(tile_sorted, torch.arange(TX * TY + 1))
    color = bg.reshape(3, 1, 1).expand(3, H, W).clone()
    depth = torch.full((1, H, W), float("nan"), dtype=dtype) if frame.get("depth_mode", 0) == 0 else torch.zeros(1, H, W, dtype=dtype)
    alpha_out = torch.zeros(1, H, W, dtype=dtype)
    sub = frame.get("subpix")
    tiles = torch.arange(TX * TY) if tile_subset is None else torch.as_tensor(list(tile_subset), dtype=torch.int64)
    lens = (bounds[tiles + 1] - bounds[tiles])
    keep = lens > 0
    tiles, lens = tiles[keep], lens[keep]
    by_len = torch.argsort(lens)
    tiles, lens = tiles[by_len], lens[by_len]
    # per-Gaussian tensors with one extra NULL row (index N: opacity 0) that pads the shorter lists of a chunk
    def padded(v, fill=0.0):
        return torch.cat([v, torch.full((1,) + tuple(v.shape[1:]), fill, dtype=v.dtype)], 0)
    mx, my, cA, cB, cC, op, tz = (padded(P[k]) for k in ("mx", "my", "cA", "cB", "cC", "op", "tz"))
    rgb = padded(P["rgb"])
    py, px = torch.meshgrid(torch.arange(TILE), torch.arange(TILE), indexing="ij")
    px, py = px.reshape(-1), py.reshape(-1)
    pairs = 0
    n_done = 0
    CH = max(1, int(chunk_pairs // (TILE * TILE)))   # list entries per chunk
    i = 0
    outs = []
    while i < tiles.numel():
        # tiles sorted by list length: a chunk of nt tiles padded to the longest of them
        Lmax_guess = int(lens[min(i + 8, tiles.numel() - 1)])
        nt = max(1, min(tiles.numel() - i, CH // max(Lmax_guess, 1)))
        tt, ll = tiles[i:i + nt], lens[i:i + nt]
        Lmax = int(ll.max())
        i += nt
        n_done += nt
        pairs += int(ll.sum()) * TILE * TILE
        ar = torch.arange(Lmax)[None, :]
        pos = bounds[tt][:, None] + ar                                        # [nt, Lmax]
        valid = ar < ll[:, None]
        g = torch.where(valid, g_sorted[pos.clamp(max=max(D - 1, 0))], torch.full_like(pos, N))
        X = ((tt % TX) * TILE)[:, None] + px[None]                            # [nt, 256] pixel coordinates
        Y = ((tt // TX) * TILE)[:, None] + py[None]
        inside = (X < W) & (Y < H)
        sx, sy = X.to(dtype), Y.to(dtype)
        if sub is not None:
            so = sub[Y.clamp(max=H - 1), X.clamp(max=W - 1)].to(dtype)
            sx, sy = sx + so[..., 0], sy + so[..., 1]
        dx = mx[g][:, :, None] - sx[:, None, :]                               # [nt, Lmax, 256]
        dy = my[g][:, :, None] - sy[:, None, :]
        power = -0.5 * (cA[g][:, :, None] * dx * dx + cC[g][:, :, None] * dy * dy) - cB[g][:, :, None] * dx * dy
        raw = op[g][:, :, None] * torch.exp(power)
        a = torch.where(raw > 0.99, 0.99 + (raw - raw.detach()), raw)         # clamp ignored in the derivative [UPSTREAM]
        ok = (power <= 0) & (a >= 1.0 / 255.0)
        a_eff = torch.where(ok, a, torch.zeros_like(a))
        T_incl = torch.cumprod(1 - a_eff, dim=1)                               # transmittance after each entry (skips: x1)
        T_excl = torch.cat([torch.ones_like(T_incl[:, :1]), T_incl[:, :-1]], 1)
        # the first entry with ok and T (1 - alpha) < 1e-4 stops the pixel and is itself not applied (a decision:
        # computed on detached products)
        stop = ok & (T_incl.detach() < 0.0001)
        stopped = torch.cummax(stop.to(torch.int8), dim=1).values.bool()
        use = ok & ~stopped
        w = torch.where(use, a * T_excl, torch.zeros_like(a))
        c = torch.einsum("tlp,tlc->tcp", w, rgb[g])                            # [nt, 3, 256]
        d = (w * tz[g][:, :, None]).sum(1)
        T_fin = torch.cumprod(torch.where(use, 1 - a, torch.ones_like(a)), dim=1)[:, -1]
        c = c + T_fin[:, None, :] * bg[None, :, None]
        al = 1 - T_fin
        if frame.get("depth_mode", 0) == 0:
            safe = torch.where(al > 0, al, torch.ones_like(al))
            d = torch.where(al > 0, d / safe, torch.full_like(al, float("nan")))
        outs.append((X[inside], Y[inside], c.permute(1, 0, 2)[:, inside], d[inside], al[inside]))
    for X, Y, c, d, al in outs:     # scatter the tiles into the frame (differentiable index_put)
        color[:, Y, X] = c
        depth[0, Y, X] = d
        alpha_out[0, Y, X] = al
    stats = dict(num_duplicates=D, num_visible=int(vis.sum()), pair_evaluations=int(pairs), tiles=int(n_done))
    return color, depth, alpha_out, P["radii"], stats
