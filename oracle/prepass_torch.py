"""Oracle of the fused pre-pass (TEST INFRASTRUCTURE): the reference's three getters restated as plain torch
(scene/gaussian_model.py:207-213, 216-217, 237-249) followed by render()'s .float() casts
(gaussian_renderer/__init__.py:137-138). Pinned against golden vectors produced by the REAL GaussianModel
(tests/golden/make_golden.py)."""
import torch


def prepass_reference(scaling_raw, opacity_raw, rotation_raw, filter_3D):
    scales = torch.exp(scaling_raw)
    s_filt = torch.sqrt(torch.square(scales) + torch.square(filter_3D))
    opacity = torch.sigmoid(opacity_raw)
    scales_square = torch.square(scales)
    det1 = scales_square.prod(dim=1)
    det2 = (scales_square + torch.square(filter_3D)).prod(dim=1)
    coef = torch.sqrt(det1 / det2)
    o_filt = opacity * coef[..., None]
    rot = torch.nn.functional.normalize(rotation_raw)
    return s_filt.float(), o_filt.float(), rot
