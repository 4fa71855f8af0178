"""sfgs.densify.reorder_zcurve is a pure relabelling of the model (CPU test: it is plain torch indexing)."""
import types

import torch
from torch import nn

from sfgs import densify


def _model(n=500, seed=0):
    g = torch.Generator().manual_seed(seed)
    m = types.SimpleNamespace()
    m._xyz = nn.Parameter(torch.randn(n, 3, generator=g) * 10)
    m._features_dc = nn.Parameter(torch.randn(n, 1, 3, generator=g))
    m._features_rest = nn.Parameter(torch.randn(n, 3, 3, generator=g))
    m._opacity = nn.Parameter(torch.randn(n, 1, generator=g).double())     # float64 after reset_opacity
    m._scaling = nn.Parameter(torch.randn(n, 3, generator=g))
    m._rotation = nn.Parameter(torch.randn(n, 4, generator=g))
    shared = nn.Parameter(torch.randn(7, 5, generator=g))
    groups = [{"params": [m._xyz], "lr": 1e-3, "name": "xyz"}, {"params": [m._features_dc], "lr": 1e-3, "name": "f_dc"},
              {"params": [m._features_rest], "lr": 1e-3, "name": "f_rest"}, {"params": [m._opacity], "lr": 1e-3, "name": "opacity"},
              {"params": [m._scaling], "lr": 1e-3, "name": "scaling"}, {"params": [m._rotation], "lr": 1e-3, "name": "rotation"},
              {"params": [shared], "lr": 1e-3, "name": "appearance_mlp"}]
    m.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
    loss = sum((p ** 2).sum() * (i + 1) for i, p in enumerate([m._xyz, m._features_dc, m._features_rest, m._opacity.float(),
                                                                 m._scaling, m._rotation, shared]))
    loss.backward()
    m.optimizer.step()
    m.optimizer.zero_grad(set_to_none=True)
    for k in ("xyz_gradient_accum", "xyz_gradient_accum_abs", "xyz_gradient_accum_abs_max", "denom"):
        setattr(m, k, torch.rand(n, 1, generator=g))
    m.max_radii2D = torch.rand(n, generator=g)
    m.filter_3D = torch.rand(n, 1, generator=g).double()
    m.shared = shared
    return m


def test_reorder_is_a_consistent_relabelling():
    m = _model()
    before = {k: getattr(m, k).detach().clone() for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling",
                                                          "_rotation", "xyz_gradient_accum", "denom", "max_radii2D", "filter_3D")}
    st_before = {g["name"]: {k: v.clone() for k, v in m.optimizer.state[g["params"][0]].items() if torch.is_tensor(v) and v.dim()}
                 for g in m.optimizer.param_groups}
    shared_state = m.optimizer.state[m.shared]["exp_avg"].clone()
    perm = densify.reorder_zcurve(m)
    assert sorted(perm.tolist()) == list(range(500))
    for k, v in before.items():
        assert torch.equal(getattr(m, k).detach(), v[perm]), k
        assert getattr(m, k).dtype == v.dtype
    for g in m.optimizer.param_groups:
        p = g["params"][0]
        if g["name"] == "appearance_mlp":
            assert p is m.shared and torch.equal(m.optimizer.state[p]["exp_avg"], shared_state)
            continue
        assert p.requires_grad and isinstance(p, nn.Parameter)
        for k, v in st_before[g["name"]].items():
            assert torch.equal(m.optimizer.state[p][k], v[perm]), (g["name"], k)
    assert m._xyz is m.optimizer.param_groups[0]["params"][0]
    # the optimizer keeps working on the relabelled tensors
    (m._xyz ** 2).sum().backward()
    m.optimizer.step()


def test_zcurve_order_is_spatially_coherent():
    x = torch.rand(4096, 3)
    perm = densify.zcurve_permutation(x)
    d_sorted = (x[perm][1:] - x[perm][:-1]).norm(dim=1).mean()
    d_rand = (x[1:] - x[:-1]).norm(dim=1).mean()
    assert d_sorted < 0.25 * d_rand
