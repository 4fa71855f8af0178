"""Restatement of the reference's densification DECISIONS (which Gaussians are cloned / split / pruned) -- TEST
INFRASTRUCTURE for the "densification indices bit-exactly" bar of BASELINE.json's north_star.

Follows scene/gaussian_model.py of the reference:
  densify_and_prune   :707-742   grads = accum / denom (NaN -> 0); ratio; Q = quantile(grads_abs, 1 - ratio) (0.99 when
                                 the tensor is empty / non-finite or torch.quantile raises, e.g. above 16 M elements)
  densify_and_clone   :686-705   (|grads| >= max_grad  or  |grads_abs| >= Q)  and  max(scaling) <= percent_dense*extent
  densify_and_split   :653-684   the same test on the post-clone set (gradients zero-padded for the clones) with
                                 max(scaling) > percent_dense * extent; N = 2 children with scaling / (0.8 N)
  final prune         :730-738   opacity < min_opacity  [or max_radii2D > max_screen_size or max(scaling) > 0.1 extent];
                                 densification_postfix (:626-651) has reset max_radii2D to zeros by then, so the screen
                                 size term never fires -- restated as the reference behaves.

The masks it returns are checked against masks captured from the REAL methods (tests/golden/make_golden_r2.py ->
tests/golden/reference_densify.npz) on every CPU run; the GPU test then feeds it the HIP path's gradients. Only the
decisions are restated: sampling of the split children is random in the reference and not a decision."""
import torch


def quantile_threshold(grads, grads_abs, max_grad):
    if grads_abs.numel() > 0 and not torch.isinf(grads_abs).any() and not torch.isnan(grads_abs).any():
        ratio = (torch.norm(grads, dim=-1) >= max_grad).float().mean()
        try:
            return torch.quantile(grads_abs.reshape(-1), 1 - ratio)
        except Exception:
            return 0.99
    return 0.99


def decisions(xyz_gradient_accum, xyz_gradient_accum_abs, denom, scaling, opacity, max_grad, min_opacity, extent,
              max_screen_size, percent_dense, n_split=2):
    """scaling = get_scaling [N,3] (activated), opacity = get_opacity [N,1] (activated, WITHOUT the 3D filter, as
    densify_and_prune reads them). Returns dict(clone[N], split[N + n_clone], prune[N + n_clone + n_split*n_split_sel
    - n_split_sel], Q)."""
    grads = xyz_gradient_accum / denom
    grads[grads.isnan()] = 0.0
    grads_abs = xyz_gradient_accum_abs / denom
    grads_abs[grads_abs.isnan()] = 0.0
    Q = quantile_threshold(grads, grads_abs, max_grad)
    smax = torch.max(scaling, dim=1).values
    sel = torch.logical_or(torch.norm(grads, dim=-1) >= max_grad, torch.norm(grads_abs, dim=-1) >= Q)
    clone = torch.logical_and(sel, smax <= percent_dense * extent)
    # after the clone: clones are appended with the SAME raw scaling / opacity
    scaling1 = torch.cat([scaling, scaling[clone]])
    opacity1 = torch.cat([opacity, opacity[clone]])
    n1 = scaling1.shape[0]
    pg = torch.zeros(n1, dtype=grads.dtype)
    pg[:grads.shape[0]] = grads.squeeze()
    pga = torch.zeros(n1, dtype=grads_abs.dtype)
    pga[:grads_abs.shape[0]] = grads_abs.squeeze()
    sel1 = torch.logical_or(pg >= max_grad, pga >= Q)
    split = torch.logical_and(sel1, torch.max(scaling1, dim=1).values > percent_dense * extent)
    # children: scaling / (0.8 N) (through log / exp in the reference: inverse activation then activation), same opacity
    child_scaling = torch.exp(torch.log(scaling1[split].repeat(n_split, 1) / (0.8 * n_split)))
    scaling2 = torch.cat([scaling1, child_scaling])[torch.cat([~split, torch.ones(child_scaling.shape[0], dtype=torch.bool)])]
    opacity2 = torch.cat([opacity1, opacity1[split].repeat(n_split, 1)])[
        torch.cat([~split, torch.ones(child_scaling.shape[0], dtype=torch.bool)])]
    prune = (opacity2 < min_opacity).squeeze(-1)
    if max_screen_size:
        big_vs = torch.zeros_like(prune)     # max_radii2D was reset by densification_postfix
        big_ws = scaling2.max(dim=1).values > 0.1 * extent
        prune = torch.logical_or(torch.logical_or(prune, big_vs), big_ws)
    return dict(clone=clone, split=split, prune=prune, Q=Q)


def margins(xyz_gradient_accum, xyz_gradient_accum_abs, denom, max_grad, Q):
    """Smallest relative distance of any Gaussian's statistic from the threshold it is compared with: decisions can only
    differ between two gradient implementations that agree to better than this."""
    grads = (xyz_gradient_accum / denom).nan_to_num(0.0).norm(dim=-1)
    grads_abs = (xyz_gradient_accum_abs / denom).nan_to_num(0.0).norm(dim=-1)
    m1 = ((grads - max_grad).abs() / max_grad).min()
    m2 = ((grads_abs - Q).abs() / Q).min() if float(Q) > 0 else torch.tensor(float("inf"))
    return float(m1), float(m2)
