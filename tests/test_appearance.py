"""sfgs.appearance: the appearance MLP's weight gradients as a split-K reduction (plain torch; host logic, runs on CPU).
The forward is F.linear's own; the weight gradient equals torch's up to the summation order of an fp32 reduction; the hook only
acts inside EmbeddingModel.forward and leaves the module, its parameters and its pickle untouched."""
import io
import types

import pytest
import torch
from torch import nn

from sfgs import appearance as ap


class _EmbeddingModel(nn.Module):   # the shape of scene/gaussian_model.py:44-58 (three Linear layers behind ReLUs)
    def __init__(self):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(59, 128), nn.ReLU(), nn.Linear(128, 128), nn.ReLU(), nn.Linear(128, 6))

    def forward(self, x):
        return self.mlp(x) * 0.01


def _fake_module():
    m = types.ModuleType("fake_gaussian_model")
    m.EmbeddingModel = _EmbeddingModel
    return m


@pytest.mark.parametrize("n", [70_001, 131_072])
def test_split_k_linear_is_f_linear_with_a_reordered_weight_reduction(n):
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, 59, generator=g, requires_grad=True)
    lin = nn.Linear(59, 128)
    up = torch.randn(n, 128, generator=g) / n
    want = lin(x)
    want.backward(up)
    ref = (x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    x.grad = None; lin.weight.grad = None; lin.bias.grad = None
    got = ap.SplitKLinear.apply(x, lin.weight, lin.bias)
    assert torch.equal(got, want)                                   # the forward is F.linear itself
    got.backward(up)
    assert torch.equal(x.grad, ref[0])                              # dX = dY W, the same product
    scale = float(ref[1].abs().max())
    assert float((lin.weight.grad - ref[1]).abs().max()) <= 2e-6 * scale
    assert float((lin.bias.grad - ref[2]).abs().max()) <= 2e-6 * float(ref[2].abs().max())


def test_hook_acts_only_inside_the_embedding_model_and_only_on_tall_inputs(monkeypatch):
    mod = _fake_module()
    calls = []
    orig_apply = ap.SplitKLinear.apply
    monkeypatch.setattr(ap.SplitKLinear, "apply", staticmethod(lambda *a: (calls.append(a[0].shape[0]), orig_apply(*a))[1]))
    plain_forward = nn.Linear.forward
    ap.install(mod)
    try:
        torch.manual_seed(0)
        e = mod.EmbeddingModel()
        tall, short = torch.randn(ap.MIN_ROWS, 59), torch.randn(100, 59)
        y = e(tall)
        assert calls == [ap.MIN_ROWS] * 3 and nn.Linear.forward is plain_forward      # three layers; the patch is gone again
        e(short)
        assert len(calls) == 3                                                        # a short input: F.linear
        with torch.no_grad():
            e(tall)
        assert len(calls) == 3                                                        # no graph: nothing to gain
        nn.Linear(59, 4)(tall)
        assert len(calls) == 3                                                        # outside EmbeddingModel.forward: untouched
        y.sum().backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in e.parameters())
        # the module pickles as before (capture() saves the module object, scene/gaussian_model.py:139)
        buf = io.BytesIO()
        torch.save(e.state_dict(), buf)
        assert set(e.state_dict()) == {"mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias", "mlp.4.weight", "mlp.4.bias"}
        assert "forward" not in e.__dict__ and all("forward" not in m.__dict__ for m in e.mlp)
    finally:
        ap.uninstall(mod)
    assert mod.EmbeddingModel.forward is _EmbeddingModel.forward
