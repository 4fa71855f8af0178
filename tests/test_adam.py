"""Fused Adam against golden vectors produced by the REAL reference: GaussianModel.training_setup ->
torch.optim.Adam steps -> densification_postfix -> prune_points -> steps -> replace_tensor_to_optimizer -> step
(tests/golden/make_golden.py::make_optimizer_golden). The sequence is replayed (a) on the numpy oracle (CPU; pins
the oracle) and (b) on sfgs.adam.FusedAdam with the reference's optimizer surgery restated on torch tensors (GPU)."""
import os

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_optimizer.npz"))
GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "appearance_embeddings", "embeddings",
          "appearance_mlp")
PER_GAUSSIAN = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "embeddings")
# f32 rounding differs by a few ulp between torch's CPU kernels, numpy and the HIP kernel (fma contraction);
# parameters move by lr * O(1) per step, so errors are bounded by lr * few ulp
RTOL, ATOL = 2e-5, 2e-7


def _n_params(name):
    return len([k for k in G.files if k.startswith(f"opt_init_{name}_")])


def _grads(k, name):
    return [G[f"opt_s{k}_g_{name}_{i}"] if f"opt_s{k}_g_{name}_{i}" in G.files else None
            for i in range(_n_params(name))]


def _check_final(get):
    """get(name, i) -> (param, exp_avg, exp_avg_sq, step) as numpy."""
    for name in GROUPS:
        for i in range(_n_params(name)):
            p, m, v, step = get(name, i)
            assert step == float(G[f"opt_final_step_{name}_{i}"]), (name, i)
            np.testing.assert_allclose(p, G[f"opt_final_{name}_{i}"], rtol=RTOL, atol=ATOL, err_msg=f"{name}[{i}]")
            for got, key in ((m, "m"), (v, "v")):  # moments: cancellation near zero -> atol relative to the tensor
                want = G[f"opt_final_{key}_{name}_{i}"]
                np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-6 * float(np.abs(want).max(initial=0)),
                                           err_msg=f"{key} {name}[{i}]")


def test_oracle_replays_reference_optimizer_sequence():
    from oracle.adam_np import AdamOracle
    groups = [{"name": n, "lr": float(G[f"opt_lr_{n}"]), "weight_decay": float(G[f"opt_wd_{n}"]),
               "params": [G[f"opt_init_{n}_{i}"].copy() for i in range(_n_params(n))]} for n in GROUPS]
    by = {g["name"]: g for g in groups}
    assert by["appearance_embeddings"]["weight_decay"] > 0  # the golden exercises the L2 branch
    opt = AdamOracle(groups, betas=tuple(G["opt_betas"]), eps=float(G["opt_eps"]))

    def step(k):
        by["xyz"]["lr"] = float(G[f"opt_s{k}_xyzlr"])
        opt.step({n: _grads(k, n) for n in GROUPS})

    for k in range(3):
        step(k)
    keep = ~G["opt_prune_mask"]
    for n in PER_GAUSSIAN:  # cat_tensors_to_optimizer then _prune_optimizer (scene/gaussian_model.py:563-624)
        new = G[f"opt_cat_{n}"]
        st = opt.state[(n, 0)]
        by[n]["params"][0] = np.concatenate([by[n]["params"][0], new])[keep]
        st["m"] = np.concatenate([st["m"], np.zeros_like(new)])[keep]
        st["v"] = np.concatenate([st["v"], np.zeros_like(new)])[keep]
    step(3)
    step(4)
    by["opacity"]["params"][0] = G["opt_replace_opacity"].copy()  # replace_tensor_to_optimizer (:549-561)
    st = opt.state[("opacity", 0)]
    st["m"], st["v"] = np.zeros_like(st["m"]), np.zeros_like(st["v"])
    step(5)
    _check_final(lambda n, i: (by[n]["params"][i], opt.state[(n, i)]["m"], opt.state[(n, i)]["v"],
                               opt.state[(n, i)]["step"]))


def test_fused_adam_is_a_torch_adam_and_rejects_what_it_cannot_do():
    from sfgs.adam import FusedAdam
    p = torch.nn.Parameter(torch.zeros(4))
    opt = FusedAdam([{"params": [p], "lr": 0.1, "name": "xyz"}], lr=0.0, eps=1e-15)
    assert isinstance(opt, torch.optim.Adam) and opt.param_groups[0]["name"] == "xyz"
    assert opt.param_groups[0]["eps"] == 1e-15 and opt.param_groups[0]["lr"] == 0.1
    opt.step()  # no gradients anywhere: nothing to do, no state created (torch semantics)
    assert len(opt.state) == 0
    p.grad = torch.ones(4)
    with pytest.raises(ValueError, match="GPU"):
        opt.step()  # CPU parameter: loud failure, no silent fallback
    with pytest.raises(NotImplementedError):
        FusedAdam.from_adam(torch.optim.Adam([p], amsgrad=True))
    sd = opt.state_dict()
    assert sd["param_groups"][0]["name"] == "xyz"


# ---- GPU: the product path --------------------------------------------------------------------------------------
class _Model:
    """The reference's optimizer surgery restated on one-parameter groups (scene/gaussian_model.py:549-624)."""

    def __init__(self, opt):
        self.optimizer = opt

    def _group(self, name):
        return [g for g in self.optimizer.param_groups if g["name"] == name][0]

    def cat_then_prune(self, name, new, keep):
        grp = self._group(name)
        old = grp["params"][0]
        st = self.optimizer.state.get(old, None)
        st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(new)), dim=0)[keep]
        st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(new)), dim=0)[keep]
        del self.optimizer.state[old]
        grp["params"][0] = torch.nn.Parameter(torch.cat((old, new), dim=0)[keep].requires_grad_(True))
        self.optimizer.state[grp["params"][0]] = st

    def replace(self, name, tensor):
        grp = self._group(name)
        old = grp["params"][0]
        st = self.optimizer.state.get(old, None)
        st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(tensor), torch.zeros_like(tensor)
        del self.optimizer.state[old]
        grp["params"][0] = torch.nn.Parameter(tensor.requires_grad_(True))
        self.optimizer.state[grp["params"][0]] = st


def _build(dev, cls, **kw):
    groups = []
    for n in GROUPS:
        groups.append({"params": [torch.nn.Parameter(torch.tensor(G[f"opt_init_{n}_{i}"], device=dev))
                                  for i in range(_n_params(n))],
                       "lr": float(G[f"opt_lr_{n}"]), "name": n, "weight_decay": float(G[f"opt_wd_{n}"])})
    return cls(groups, lr=0.0, eps=float(G["opt_eps"]), **kw)


def _replay(model, dev):
    opt = model.optimizer

    def step(k):
        model._group("xyz")["lr"] = float(G[f"opt_s{k}_xyzlr"])
        for n in GROUPS:
            for prm, g in zip(model._group(n)["params"], _grads(k, n)):
                prm.grad = None if g is None else torch.tensor(g, device=dev)
        opt.step()
        opt.zero_grad(set_to_none=True)

    for k in range(3):
        step(k)
    keep = torch.tensor(~G["opt_prune_mask"], device=dev)
    for n in PER_GAUSSIAN:
        model.cat_then_prune(n, torch.tensor(G[f"opt_cat_{n}"], device=dev), keep)
    step(3)
    step(4)
    model.replace("opacity", torch.tensor(G["opt_replace_opacity"], device=dev))
    step(5)


@pytest.mark.gpu
def test_fused_adam_replays_reference_optimizer_sequence():
    from sfgs.adam import FusedAdam
    dev = torch.device("cuda:0")
    model = _Model(_build(dev, FusedAdam))
    _replay(model, dev)

    def get(n, i):
        prm = model._group(n)["params"][i]
        st = model.optimizer.state[prm]
        return prm.detach().cpu().numpy(), st["exp_avg"].cpu().numpy(), st["exp_avg_sq"].cpu().numpy(), float(st["step"])
    _check_final(get)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 3, 4095, 4096, 4097, 100_003, 1_000_000])
def test_fused_adam_matches_torch_adam_on_gpu(n):
    """Same device, same inputs: torch.optim.Adam (foreach path, i.e. the reference's) vs the fused launch, 4 steps,
    including an unaligned view (storage offset 1 float) and weight decay."""
    from sfgs.adam import FusedAdam
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(n)
    init = [torch.randn(n, 3, generator=gen), torch.randn(n + 1, generator=gen), torch.randn(max(n, 1), 7, generator=gen)]

    def make(cls):
        a = torch.nn.Parameter(init[0].to(dev))
        b = torch.nn.Parameter(init[1].to(dev)[1:])  # 4-byte-aligned only
        c = torch.nn.Parameter(init[2].to(dev))
        return cls([{"params": [a], "lr": 1e-2}, {"params": [b, c], "lr": 3e-3, "weight_decay": 0.1}], lr=0.0, eps=1e-15), [a, b, c]
    ref, pr = make(torch.optim.Adam)
    fus, pf = make(FusedAdam)
    for s in range(4):
        for x, y in zip(pr, pf):
            g = torch.randn(x.shape, generator=gen).to(dev) * 10.0 ** (s - 2)
            x.grad, y.grad = g.clone(), g.clone()
        ref.step(); fus.step()
    for x, y in zip(pr, pf):
        torch.testing.assert_close(y, x, rtol=RTOL, atol=ATOL)
        if x.numel():
            for k in ("exp_avg", "exp_avg_sq"):
                want = ref.state[x][k]
                torch.testing.assert_close(fus.state[y][k], want, rtol=RTOL, atol=1e-6 * float(want.abs().max()))
            assert float(fus.state[y]["step"]) == float(ref.state[x]["step"]) == 4.0


def test_dense_layout_rule():
    from sfgs.adam import _dense
    x = torch.zeros(7, 3)
    assert _dense(x) and _dense(x.t()) and _dense(torch.zeros(3, 7).t()) and _dense(torch.zeros(5, 1, 3).transpose(1, 2))
    assert not _dense(x[:, :2]) and not _dense(x[::2]) and not _dense(torch.zeros(4).expand(3, 4))


@pytest.mark.gpu
def test_fused_adam_steps_a_column_major_parameter_like_the_references_initial_xyz():
    """create_from_pcd builds `_xyz` from fetchPly's `np.vstack([x, y, z]).T` (scene/gaussian_model.py:316, dataset_readers.py:
    126-132): torch.tensor keeps the transposed strides, so until the first densification the reference's position parameter
    is COLUMN-major. torch.optim.Adam does not care; the fused step must not either (found by the real train.training())."""
    from sfgs.adam import FusedAdam
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(3)
    base = torch.randn(3, 5001, generator=gen)

    def make(cls):
        a = torch.nn.Parameter(base.to(dev).t())            # [5001, 3], strides (1, 5001)
        assert not a.is_contiguous()
        return cls([{"params": [a], "lr": 1e-2}], lr=0.0, eps=1e-15), a
    ref, x = make(torch.optim.Adam)
    fus, y = make(FusedAdam)
    for s in range(3):
        g = torch.randn(5001, 3, generator=gen).to(dev)
        # autograd hands a leaf a gradient in the leaf's own layout; a contiguous one must work too (step 2)
        x.grad = torch.empty_like(x).copy_(g) if s != 2 else g.clone()
        y.grad = torch.empty_like(y).copy_(g) if s != 2 else g.clone()
        ref.step(); fus.step()
    assert y.stride() == x.stride() == (1, 5001)
    torch.testing.assert_close(y, x, rtol=RTOL, atol=ATOL)
    for k in ("exp_avg", "exp_avg_sq"):
        torch.testing.assert_close(fus.state[y][k], ref.state[x][k], rtol=RTOL, atol=1e-6 * float(ref.state[x][k].abs().max()))


@pytest.mark.gpu
def test_install_rehomes_the_optimizer_training_setup_builds():
    from sfgs import adam

    class GaussianModel:  # shaped like scene/gaussian_model.py:350-382
        def training_setup(self, training_args):
            self._xyz = torch.nn.Parameter(torch.zeros(10, 3, device="cuda:0"))
            self.optimizer = torch.optim.Adam([{"params": [self._xyz], "lr": 0.5, "name": "xyz"}], lr=0.0, eps=1e-15)
    adam.install(GaussianModel)
    try:
        m = GaussianModel()
        m.training_setup(None)
        assert isinstance(m.optimizer, adam.FusedAdam) and m.optimizer.param_groups[0]["params"][0] is m._xyz
        m._xyz.grad = torch.ones_like(m._xyz)
        v0 = m._xyz._version
        m.optimizer.step()
        assert m._xyz._version > v0   # in-place update is visible to autograd / version-keyed caches (sfgs.prepass)
        torch.testing.assert_close(m._xyz.detach(), torch.full((10, 3), -0.5, device="cuda:0"), rtol=1e-6, atol=0)
        sd = m.optimizer.state_dict()     # capture()/restore() path (scene/gaussian_model.py:163-201)
        m2 = GaussianModel(); m2.training_setup(None); m2.optimizer.load_state_dict(sd)
        assert float(m2.optimizer.state[m2._xyz]["step"]) == 1.0
    finally:
        adam.uninstall(GaussianModel)
