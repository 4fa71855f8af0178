"""Oracle of the fused Adam step (TEST INFRASTRUCTURE -- never imported by the product path).

The reference optimises every Gaussian tensor with `torch.optim.Adam(l, lr=0.0, eps=1e-15)`
(scene/gaussian_model.py:382; stepped at train.py:339,906). The algorithm therefore lives in a third-party
dependency, PyTorch (unpinned in the reference: README.md:67 installs the latest wheel), whose published update rule
(torch/optim/adam.py, `_single_tensor_adam` / `_multi_tensor_adam`, no amsgrad, no maximize, L2 weight decay) is
restated here in numpy float32 with the same rounding points:

    t   <- t + 1
    g'  = g + wd * p                       (if wd != 0)
    m   = m + (1-b1) * (g' - m)            (Tensor.lerp_, weight < 0.5 form)
    v   = v * b2 + (1-b2) * g' * g'        (mul_, addcmul_)
    den = sqrt(v) / sqrt(1 - b2^t) + eps
    p   = p - (lr / (1 - b1^t)) * (m / den)

Host scalars are formed in double and rounded to float32 when they meet the tensors, like the foreach kernels do.
Pinned by tests/test_adam.py against golden vectors produced by the REAL reference GaussianModel driving the real
torch.optim.Adam through training_setup / densification_postfix / prune_points / replace_tensor_to_optimizer
(tests/golden/make_golden.py -> reference_optimizer.npz)."""
import numpy as np

F = np.float32


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """One in-place Adam update of float32 arrays p, m, v with gradient g; `step` is the already incremented count."""
    assert p.dtype == m.dtype == v.dtype == F and g.dtype == F
    if weight_decay != 0:
        g = g + F(weight_decay) * p
    m += F(1 - beta1) * (g - m)
    v *= F(beta2)
    v += F(1 - beta2) * (g * g)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    den = np.sqrt(v) / F(bc2 ** 0.5) + F(eps)
    p += F(-(lr / bc1)) * (m / den)
    return p, m, v


class AdamOracle:
    """Minimal stand-in for the optimizer object the reference manipulates: groups of named float32 arrays with
    per-array state, so a recorded training/densification sequence can be replayed without torch."""

    def __init__(self, groups, betas=(0.9, 0.999), eps=1e-15):
        # groups: list of dicts {"name", "lr", "weight_decay", "params": [np arrays]}
        self.groups = groups
        self.betas, self.eps = betas, eps
        self.state = {}  # (group name, index) -> {"step", "m", "v"}

    def step(self, grads):
        """grads: {group name: [array or None per param]} (None -> skipped, like a parameter without .grad)."""
        for grp in self.groups:
            for i, p in enumerate(grp["params"]):
                g = grads.get(grp["name"], [None] * len(grp["params"]))[i]
                if g is None:
                    continue
                st = self.state.setdefault((grp["name"], i), {"step": 0, "m": np.zeros_like(p), "v": np.zeros_like(p)})
                st["step"] += 1
                adam_step(p, g.astype(F), st["m"], st["v"], st["step"], grp["lr"], self.betas[0], self.betas[1],
                          self.eps, grp.get("weight_decay", 0.0))
