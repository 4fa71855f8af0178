"""One-pass pruning: sfgs.compact against torch's `tensor[mask]` (bit-exact: pure data movement)
and against the golden sequence recorded from the REAL GaussianModel.prune_points (reference_optimizer.npz)."""
import numpy as np
import pytest
import torch

from test_adam import G, GROUPS, PER_GAUSSIAN, _build, _check_final, _grads


def test_contract_errors_without_gpu():
    from sfgs.compact import compact_rows
    with pytest.raises(ValueError, match="no CPU fallback"):
        compact_rows(torch.ones(4, dtype=torch.bool), [torch.zeros(4, 3)])
    with pytest.raises(ValueError, match="bool"):
        compact_rows(torch.ones(4), [torch.zeros(4, 3)])


@pytest.mark.gpu
@pytest.mark.parametrize("n,p_keep", [(0, 0.5), (1, 1.0), (1, 0.0), (777, 0.5), (4096, 0.9), (4097, 0.1),
                                      (100_003, 0.97), (2_000_000, 0.8), (300_000, 1.0), (300_000, 0.0)])
def test_compact_rows_is_bit_exact(n, p_keep):
    from sfgs.compact import compact_rows
    dev = "cuda:0"
    g = torch.Generator().manual_seed(n + int(p_keep * 100))
    keep = (torch.rand(n, generator=g) < p_keep).to(dev)
    ts = [torch.randn(n, 3, generator=g), torch.randn(n, 1, 3, generator=g), torch.randn(n, 3, 3, generator=g),
          torch.randn(n, 1, generator=g), torch.randn(n, 24, generator=g), torch.randn(n, generator=g),
          torch.randn(n, 1, generator=g, dtype=torch.float64), torch.randint(-5, 5, (n, 2), generator=g, dtype=torch.int32)]
    ts = [t.to(dev) for t in ts]
    ts[0][::5] = float("nan")  # payload bits must survive untouched
    outs = compact_rows(keep, ts)
    for t, o in zip(ts, outs):
        want = t[keep]
        assert o.shape == want.shape and o.dtype == want.dtype
        assert torch.equal(o.view(torch.uint8), want.view(torch.uint8))
    # byte-granular rows (bool / uint8 / int16 [N,1]) take the generic path
    odd = [(torch.rand(n, generator=g) < 0.5).to(dev), torch.randint(0, 255, (n, 3), generator=g, dtype=torch.uint8).to(dev),
           torch.randint(-9, 9, (n, 1), generator=g, dtype=torch.int16).to(dev), ts[0]]
    if n <= 5000:  # rows wider than one block's chunk
        wide = [torch.randn(n, 3001, generator=g).to(dev), torch.randn(n, 513, 2, generator=g).to(dev)]
        for t, o in zip(wide, compact_rows(keep, wide)):
            assert torch.equal(o, t[keep])
    for t, o in zip(odd, compact_rows(keep, odd)):
        assert torch.equal(o.view(torch.uint8), t[keep].view(torch.uint8))


class _RefShapedModel:
    """Attributes GaussianModel.prune_points touches (scene/gaussian_model.py:586-603)."""
    appearance_enabled = True

    def __init__(self, opt, n, dev):
        self.optimizer = opt
        for i, s in enumerate(("xyz_gradient_accum", "xyz_gradient_accum_abs", "xyz_gradient_accum_abs_max", "denom")):
            setattr(self, s, torch.arange(n, dtype=torch.float32, device=dev).reshape(n, 1) + 1000 * i)
        self.max_radii2D = torch.arange(n, dtype=torch.float32, device=dev)
        self.sync()

    def group(self, name):
        return [g for g in self.optimizer.param_groups if g["name"] == name][0]

    def prune_points(self, mask):
        raise AssertionError("install() must replace this")

    def sync(self):
        for attr, name in (("_xyz", "xyz"), ("_features_dc", "f_dc"), ("_features_rest", "f_rest"), ("_opacity", "opacity"),
                           ("_scaling", "scaling"), ("_rotation", "rotation"), ("_embeddings", "embeddings")):
            setattr(self, attr, self.group(name)["params"][0])


@pytest.mark.gpu
def test_prune_points_replays_the_reference_sequence():
    """Same golden as tests/test_adam.py, but the prune step goes through sfgs.compact.prune_points (installed on a
    class shaped like the reference's) and the steps through FusedAdam."""
    from sfgs import compact
    from sfgs.adam import FusedAdam
    dev = torch.device("cuda:0")
    opt = _build(dev, FusedAdam)
    n0 = G["opt_init_xyz_0"].shape[0]
    model = _RefShapedModel(opt, n0, dev)

    def step(k):
        model.group("xyz")["lr"] = float(G[f"opt_s{k}_xyzlr"])
        for n in GROUPS:
            for prm, g in zip(model.group(n)["params"], _grads(k, n)):
                prm.grad = None if g is None else torch.tensor(g, device=dev)
        opt.step()
        opt.zero_grad(set_to_none=True)

    for k in range(3):
        step(k)
    for n in PER_GAUSSIAN:  # cat_tensors_to_optimizer restated (scene/gaussian_model.py:605-624)
        grp = model.group(n)
        old, new = grp["params"][0], torch.tensor(G[f"opt_cat_{n}"], device=dev)
        st = opt.state[old]
        st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(new)))
        st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(new)))
        del opt.state[old]
        grp["params"][0] = torch.nn.Parameter(torch.cat((old, new)).requires_grad_(True))
        opt.state[grp["params"][0]] = st
    n1 = n0 + G["opt_cat_xyz"].shape[0]
    model.__init__(opt, n1, dev)  # densification_postfix re-creates the statistics tensors at the new size (:641-645)
    mask = torch.tensor(G["opt_prune_mask"], device=dev)
    compact.install(_RefShapedModel)
    try:
        model.prune_points(mask)
    finally:
        compact.uninstall(_RefShapedModel)
    kept = int((~mask).sum())
    assert model._xyz.shape == (kept, 3) and model._xyz is model.group("xyz")["params"][0] and model._xyz.requires_grad
    assert model._embeddings is model.group("embeddings")["params"][0]
    want_rows = torch.arange(n1, dtype=torch.float32, device=dev)[~mask]
    assert torch.equal(model.max_radii2D, want_rows) and torch.equal(model.denom, want_rows[:, None] + 3000)
    assert opt.state[model._opacity]["exp_avg"].shape == (kept, 1)
    step(3)
    step(4)
    new_op = torch.tensor(G["opt_replace_opacity"], device=dev)  # replace_tensor_to_optimizer (:549-561)
    grp = model.group("opacity")
    st = opt.state[grp["params"][0]]
    st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(new_op), torch.zeros_like(new_op)
    del opt.state[grp["params"][0]]
    grp["params"][0] = torch.nn.Parameter(new_op.requires_grad_(True))
    opt.state[grp["params"][0]] = st
    step(5)

    def get(n, i):
        prm = model.group(n)["params"][i]
        st = opt.state[prm]
        return prm.detach().cpu().numpy(), st["exp_avg"].cpu().numpy(), st["exp_avg_sq"].cpu().numpy(), float(st["step"])
    _check_final(get)
