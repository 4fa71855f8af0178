"""The appearance MLP's weight gradients as a split-K reduction (optional hook; plain torch, no kernel of ours).

`--appearance_enabled` -- how every shipped script trains (scripts/run_jax.py:22) -- runs `EmbeddingModel`
(scene/gaussian_model.py:44-69: Linear 59 -> 128 -> 128 -> 6 with ReLUs) over ALL N Gaussians every iteration
(gaussian_renderer/__init__.py:105-110). Its forward and input-gradient GEMMs are ordinary tall matrices times small
weights. Its WEIGHT gradients are not: dW[out, in] = dY[N, out]^T X[N, in] reduces over K = N = 2 000 000 rows into a
128 x 59 result, and the library kernel torch picks for that shape (fp32, hipBLASLt / rocBLAS alike: 32 x 32 x 256 tiles, a
handful of workgroups walking the whole K) takes 3.2 ms where the same reduction cut into 64 independent row blocks -- one
batched GEMM of [64][out, N/64] x [64][N/64, in] and a sum of the 64 partial results -- takes 0.38 ms
(profiles/r6_appearance_mlp_splitk.txt). Three layers: 10.3 ms of a 17.5 ms training iteration at 2 M Gaussians.

SURVEY 8 row a8 leaves the MLP to PyTorch ("dense -> MFMA via PyTorch, not ours"), and it stays there: this hook changes
WHICH torch GEMMs compute the weight gradient, nothing else. `install(GaussianModel)` wraps `EmbeddingModel.forward` so that,
while it runs, `nn.Linear.forward` on a 2-D input of at least MIN_ROWS rows goes through `SplitKLinear` -- forward
`addmm(bias, x, W^T)` exactly as `F.linear`, backward dX = dY W as torch's, dW = sum_c dY_c^T X_c, db = sum dY. The module,
its parameters, its state_dict and the checkpoint tuple (`capture()` pickles the module object) are untouched. Values:
the forward is bit-identical; dW differs from torch's by the summation order of an fp32 reduction (tests/test_appearance.py:
<= 2e-6 relative to the largest entry)."""
import sys

import torch
import torch.nn.functional as F
from torch import nn

__all__ = ["SplitKLinear", "install", "uninstall", "MIN_ROWS", "CHUNKS"]

MIN_ROWS = 65536   # below this the single GEMM is as fast
CHUNKS = 64        # independent row blocks of the reduction


class SplitKLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = dy @ weight
        if ctx.needs_input_grad[1]:
            n = x.shape[0]
            rows = n // CHUNKS
            n0 = rows * CHUNKS
            dyc = dy[:n0].reshape(CHUNKS, rows, dy.shape[1])
            xc = x[:n0].reshape(CHUNKS, rows, x.shape[1])
            dw = torch.bmm(dyc.transpose(1, 2), xc).sum(0)
            if n0 < n:
                dw = dw + dy[n0:].t() @ x[n0:]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(0)
        return dx, dw, db


def _linear_forward(self, input):
    if input.dim() == 2 and input.shape[0] >= MIN_ROWS and input.dtype == torch.float32 and torch.is_grad_enabled() \
            and (self.weight.requires_grad or input.requires_grad):
        return SplitKLinear.apply(input, self.weight, self.bias)
    return F.linear(input, self.weight, self.bias)


_ORIG = {}


def _embedding_model(target):
    """EmbeddingModel of the reference module that defines `target` (the GaussianModel class, or the module itself)."""
    mod = sys.modules[target.__module__] if isinstance(target, type) else target
    return getattr(mod, "EmbeddingModel", None)   # None: a model without the appearance MLP (nothing to do)


def install(target):
    """target: the reference's GaussianModel class (or its module, scene.gaussian_model)."""
    E = _embedding_model(target)
    if E is None or E in _ORIG:
        return
    orig = _ORIG[E] = E.forward

    def forward(self, *args, **kwargs):
        saved = nn.Linear.forward
        nn.Linear.forward = _linear_forward
        try:
            return orig(self, *args, **kwargs)
        finally:
            nn.Linear.forward = saved
    forward.__wrapped__ = orig
    E.forward = forward


def uninstall(target):
    E = _embedding_model(target)
    if E is not None and E in _ORIG:
        E.forward = _ORIG.pop(E)
