"""Fused densify_and_prune (sfgs.densify, csrc/densify.hip) against the REAL method's outputs
(tests/golden/reference_densify_full.npz, made by tests/golden/make_golden_r2.py from scene/gaussian_model.py:603-742):
every parameter in the reference's final row order, the Adam moments (copied for survivors, zero for new rows, absent
for groups that never stepped), the reset statistics and the returned counts; and the radix-select quantile against
torch.quantile, including beyond torch's 16 M-element limit."""
import json
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_densify_full.npz")
GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "embeddings")
ATTR = dict(xyz="_xyz", f_dc="_features_dc", f_rest="_features_rest", opacity="_opacity", scaling="_scaling",
            rotation="_rotation", embeddings="_embeddings")


def _model(z, dev):
    from sfgs.adam import FusedAdam
    m = types.SimpleNamespace(appearance_enabled=True)
    groups = []
    for n in GROUPS:
        p = torch.nn.Parameter(torch.from_numpy(z["in_" + n]).to(dev))
        setattr(m, ATTR[n], p)
        groups.append(dict(params=[p], lr=float(z["lr_" + n]), name=n))
    groups.append(dict(params=[torch.nn.Parameter(torch.zeros(7, device=dev))], lr=1e-3, name="appearance_mlp"))
    m.optimizer = FusedAdam(groups, lr=0.0, eps=1e-15)
    for n in GROUPS:
        if "in_m_" + n in z.files:
            p = getattr(m, ATTR[n])
            m.optimizer.state[p] = dict(step=torch.tensor(1.0), exp_avg=torch.from_numpy(z["in_m_" + n]).to(dev),
                                        exp_avg_sq=torch.from_numpy(z["in_v_" + n]).to(dev))
    for k in ("denom", "xyz_gradient_accum", "xyz_gradient_accum_abs"):
        setattr(m, k, torch.from_numpy(z["in_" + k]).to(dev))
    m.xyz_gradient_accum_abs_max = torch.zeros_like(m.denom)
    m.max_radii2D = torch.zeros(m.denom.shape[0], device=dev)
    return m


def test_fused_densify_and_prune_matches_the_real_method():
    from sfgs import densify
    z = np.load(GOLD)
    cfg = json.loads(str(z["config"]))
    dev = torch.device("cuda:0")

    class M(types.SimpleNamespace):
        get_scaling = property(lambda self: torch.exp(self._scaling))
        get_opacity = property(lambda self: torch.sigmoid(self._opacity))
    m = _model(z, dev)
    m = M(**m.__dict__)
    m.percent_dense = cfg["percent_dense"]
    ret = densify.densify_and_prune(m, cfg["max_grad"], cfg["min_opacity"], cfg["extent"], cfg["max_screen_size"],
                                    samples=torch.from_numpy(z["samples"]))
    assert tuple(int(v) for v in ret) == tuple(int(v) for v in z["ret"])
    n_new = z["out_xyz"].shape[0]
    groups = {g["name"]: g for g in m.optimizer.param_groups}
    for n in GROUPS:
        p = getattr(m, ATTR[n])
        assert groups[n]["params"][0] is p and isinstance(p, torch.nn.Parameter) and p.requires_grad
        got, ref = p.detach().cpu().numpy(), z["out_" + n]
        assert got.shape == ref.shape, n
        if n in ("xyz", "scaling"):     # child rows are computed (R(q) sample + xyz; log(s / 1.6)): device libm vs torch's
            np.testing.assert_allclose(got, ref, rtol=2e-6, atol=2e-6, err_msg=n)
            differ = int((got != ref).any(axis=1).sum())
            assert differ <= 2 * int(z["ret"][1]), (n, differ)   # survivors and clones are copies: bit-identical rows
        else:
            np.testing.assert_array_equal(got, ref, err_msg=n)
        st = m.optimizer.state.get(p, None)
        if "out_m_" + n in z.files:
            np.testing.assert_array_equal(st["exp_avg"].cpu().numpy(), z["out_m_" + n], err_msg=n)
            np.testing.assert_array_equal(st["exp_avg_sq"].cpu().numpy(), z["out_v_" + n], err_msg=n)
        else:
            assert not st, n
    for k in ("denom", "xyz_gradient_accum", "xyz_gradient_accum_abs", "xyz_gradient_accum_abs_max", "max_radii2D"):
        got = getattr(m, k).cpu().numpy()
        assert got.shape == z["out_" + k].shape and not got.any(), k
    assert m._xyz.shape[0] == n_new


def test_default_samples_path_equals_explicit_std_times_z():
    """samples=None (what train.py gets): the kernel multiplies z ~ N(0, 1) by the parent's scaling itself (no host-side
    mask gather). Same generator state -> same z -> the result equals the explicit samples = z * std path."""
    from sfgs import densify
    z = np.load(GOLD)
    cfg = json.loads(str(z["config"]))
    dev = torch.device("cuda:0")

    class M(types.SimpleNamespace):
        get_scaling = property(lambda self: torch.exp(self._scaling))
        get_opacity = property(lambda self: torch.sigmoid(self._opacity))
    outs = []
    for explicit in (False, True):
        m = M(**_model(z, dev).__dict__)
        m.percent_dense = cfg["percent_dense"]
        raw_split = int(z["ret"][1])
        torch.manual_seed(77)
        samples = None
        if explicit:
            zz = torch.randn(2 * raw_split, 3, device=dev)
            import densify_rule
            d = densify_rule.decisions(m.xyz_gradient_accum.cpu(), m.xyz_gradient_accum_abs.cpu(), m.denom.cpu(),
                                       m.get_scaling.detach().cpu(), m.get_opacity.detach().cpu(), cfg["max_grad"],
                                       cfg["min_opacity"], cfg["extent"], cfg["max_screen_size"], cfg["percent_dense"])
            n0 = m._xyz.shape[0]
            split = d["split"][:n0].to(dev)    # clones are never split in the same pass (their gradients are padded zeros)
            assert int(split.sum()) == raw_split and not bool(d["split"][n0:].any())
            samples = zz * m.get_scaling.detach()[split].repeat(2, 1)
        ret = densify.densify_and_prune(m, cfg["max_grad"], cfg["min_opacity"], cfg["extent"], cfg["max_screen_size"],
                                        samples=samples)
        outs.append((ret, m._xyz.detach().clone(), m._scaling.detach().clone()))
    assert tuple(outs[0][0]) == tuple(outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


@pytest.mark.parametrize("n,q", [(1, 0.3), (2, 0.5), (1000, 0.0), (1000, 1.0), (100003, 0.85), (2_000_000, 0.9137)])
def test_quantile_matches_torch(n, q):
    from sfgs.densify import quantile_linear
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n)
    v = (torch.rand(n, generator=g) ** 3).to(dev)
    v[::7] = float(v[0])                # duplicates
    if n > 10:
        v[5] = 0.0
    qt = torch.tensor(q, device=dev)
    ref = torch.quantile(v, qt)
    got = quantile_linear(v, qt)
    assert float(got) == float(ref), (float(got), float(ref))


def test_quantile_beyond_torch_limit():
    from sfgs.densify import quantile_linear
    dev = torch.device("cuda:0")
    n = 20_000_000                      # torch.quantile raises above 16 M elements; the reference falls back to Q = 0.99
    v = torch.rand(n, device=dev)
    with pytest.raises(RuntimeError):
        torch.quantile(v, 0.9)
    got = float(quantile_linear(v, torch.tensor(0.9, device=dev)))
    srt = torch.sort(v).values
    rank = torch.tensor(0.9, device=dev) * (n - 1)
    lo, hi = int(rank.floor()), int(rank.ceil())
    ref = float(torch.lerp(srt[lo], srt[hi], rank - rank.floor()))
    assert got == ref
