"""sfgs.features.DeferredFeatures on CPU tensors: the handle answers render()'s cheap uses (shape, transposed view) without
running anything, and any other use sees exactly `torch.cat((_features_dc, _features_rest), dim=1)` with its autograd graph
(scene/gaussian_model.py:227-231; the uses: gaussian_renderer/__init__.py:110,114,121-122,127)."""
import torch

from sfgs import features


def _params(n=7, k=4, seed=0):
    gen = torch.Generator().manual_seed(seed)
    dc = torch.randn(n, 1, 3, generator=gen).requires_grad_(True)
    rest = torch.randn(n, k - 1, 3, generator=gen).requires_grad_(True)
    return dc, rest


def test_metadata_and_views_stay_handles():
    dc, rest = _params()
    h = features.DeferredFeatures(dc, rest)
    assert tuple(h.shape) == (7, 4, 3) and h.shape[0] == 7 and h.dtype == torch.float32 and h.requires_grad
    assert h.is_contiguous() and h.stride() == (12, 3, 1) and len(h) == 7 and h.dim() == 3
    t = h.transpose(1, 2)
    assert isinstance(t, features.DeferredFeatures) and tuple(t.shape) == (7, 3, 4) and t.stride() == (12, 1, 3)
    v = t.view(-1, 3, 4)                      # render(): pc.get_features.transpose(1, 2).view(-1, 3, (max_sh_degree+1)**2)
    assert v is t and t.view(7, 3, 4) is t and t.view((7, 3, 4)) is t
    assert h.float() is h
    assert h._sfgs_real is None and t._sfgs_real is None           # nothing ran
    assert features.split_parts(h)[2] is False and features.split_parts(t)[2] is True
    assert features.split_parts(h)[0].data_ptr() == dc.data_ptr() and features.split_parts(h)[1].data_ptr() == rest.data_ptr()
    back = t.transpose(2, 1)
    assert isinstance(back, features.DeferredFeatures) and tuple(back.shape) == (7, 4, 3)
    with torch.no_grad():
        assert not features.DeferredFeatures(dc, rest).requires_grad


def test_any_other_use_is_the_concatenation():
    dc, rest = _params(seed=1)
    ref = torch.cat((dc, rest), dim=1)
    h = features.DeferredFeatures(dc, rest)
    torch.testing.assert_close((h * 2.0 + 1.0).detach(), (ref * 2.0 + 1.0).detach(), rtol=0, atol=0)
    assert features.split_parts(h) is None                        # looked into: an ordinary tensor from now on
    torch.testing.assert_close(h[2:5].detach(), ref[2:5].detach(), rtol=0, atol=0)
    t = features.DeferredFeatures(dc, rest).transpose(1, 2)
    torch.testing.assert_close(t.reshape(7, 12).detach(), ref.transpose(1, 2).reshape(7, 12).detach(), rtol=0, atol=0)
    torch.testing.assert_close(features.DeferredFeatures(dc, rest).view(7, 12).detach(), ref.view(7, 12).detach(), rtol=0, atol=0)
    # gradients reach the two parameters through the materialised value
    w = torch.randn(7, 4, 3, generator=torch.Generator().manual_seed(3))
    (features.DeferredFeatures(dc, rest) * w).sum().backward()
    torch.testing.assert_close(dc.grad, w[:, :1], rtol=0, atol=0)
    torch.testing.assert_close(rest.grad, w[:, 1:], rtol=0, atol=0)
    # a module consuming it (the appearance MLP's position): flatten + linear
    lin = torch.nn.Linear(12, 2)
    torch.testing.assert_close(lin(features.DeferredFeatures(dc, rest).flatten(1)), lin(ref.flatten(1)), rtol=0, atol=0)


def test_install_patches_get_features_and_uninstall_restores():
    class Model:
        def __init__(self):
            self._features_dc, self._features_rest = _params(seed=2)

        @property
        def get_features(self):
            return torch.cat((self._features_dc, self._features_rest), dim=1)

    orig = Model.__dict__["get_features"]
    features.install(Model)
    try:
        m = Model()
        assert isinstance(m.get_features, features.DeferredFeatures) and m.get_features.shape[0] == 7
        torch.testing.assert_close(m.get_features + 0, torch.cat((m._features_dc, m._features_rest), dim=1), rtol=0, atol=0)
    finally:
        features.uninstall(Model)
    assert Model.__dict__["get_features"] is orig and not isinstance(Model().get_features, features.DeferredFeatures)

    class NoFeatures:
        pass
    features.install(NoFeatures)      # a class without the property: nothing to patch
    assert NoFeatures not in features._ORIG
