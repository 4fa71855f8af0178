"""the reference's reset_opacity (scene/gaussian_model.py:483-501, every opacity_reset_interval
iterations -- first at iteration 3000, train.py:325) divides by a coefficient derived from the float64 filter_3D, so
`_opacity` and its Adam moments are float64 from then on. The fused pre-pass and the fused Adam must keep running.

Golden = the REAL GaussianModel: training_setup -> 2 x (getters -> loss -> backward -> Adam step) -> the real
reset_opacity -> 2 more iterations (tests/golden/make_golden_r2.py). The GPU test replays it through
sfgs.prepass.fused_activations + sfgs.adam.FusedAdam."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_reset_opacity.npz")
NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "embeddings")
ATTR = dict(xyz="_xyz", f_dc="_features_dc", f_rest="_features_rest", opacity="_opacity", scaling="_scaling",
            rotation="_rotation", embeddings="_embeddings")


def test_golden_records_the_dtype_change():
    z = np.load(GOLD)
    assert z["after_reset_opacity"].dtype == np.float64 and z["init_opacity"].dtype == np.float32
    assert z["final_opacity"].dtype == np.float64 and z["final_m_opacity"].dtype == np.float64
    assert str(z["s0_g_opacity_dtype"]) == "torch.float32" and str(z["s2_g_opacity_dtype"]) == "torch.float64"


@pytest.mark.gpu
def test_fused_hooks_follow_the_reference_through_reset_opacity():
    from sfgs.adam import FusedAdam
    from sfgs.prepass import fused_activations
    z = np.load(GOLD)
    dev = torch.device("cuda:0")
    P = {n: torch.nn.Parameter(torch.from_numpy(z["init" + ATTR[n]]).to(dev)) for n in NAMES}
    opt = FusedAdam([dict(params=[P[n]], lr=float(z["lr_" + n]), name=n) for n in NAMES], lr=0.0, eps=1e-15)
    filt = torch.from_numpy(z["filter_3D"]).to(dev)
    assert filt.dtype == torch.float64

    def step(k):
        sc, op, ro = fused_activations(P["scaling"], P["opacity"], P["rotation"], filt)
        np.testing.assert_allclose(op.detach().cpu().numpy(), z[f"s{k}_out_opacity"], rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(sc.detach().cpu().numpy(), z[f"s{k}_out_scales"], rtol=2e-6, atol=1e-9)
        w = lambda n: torch.from_numpy(z[f"s{k}_w_{n}"]).to(dev)
        loss = (sc * w("scales")).sum() + (op * w("opacity")).sum() + (ro * w("rotation")).sum()
        for n in ("xyz", "f_dc", "f_rest", "embeddings"):
            P[n].grad = torch.from_numpy(z[f"s{k}_g_{n}"]).to(dev)
        loss.backward()
        assert str(P["opacity"].grad.dtype) == str(z[f"s{k}_g_opacity_dtype"])
        opt.step()
        opt.zero_grad(set_to_none=True)

    step(0)
    step(1)
    # replace_tensor_to_optimizer (scene/gaussian_model.py:549-562) with the value the real reset_opacity produced
    new = torch.from_numpy(z["after_reset_opacity"]).to(dev)
    grp = [g for g in opt.param_groups if g["name"] == "opacity"][0]
    st = opt.state.pop(grp["params"][0])
    st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(new), torch.zeros_like(new)
    grp["params"][0] = P["opacity"] = torch.nn.Parameter(new.requires_grad_(True))
    opt.state[P["opacity"]] = st
    step(2)
    step(3)
    report, bad = [], []
    for n in NAMES:
        st = opt.state[P[n]]
        assert P[n].dtype == torch.from_numpy(z["final_" + n]).dtype
        # Adam normalises the gradient (m / sqrt(v)): an element whose float32 gradient chain rounds differently moves the
        # parameter by a fraction of lr, so parameters are compared in units of their learning rate and the moments
        # relative to their largest element
        lr = float(z["lr_" + n])
        dp = np.abs(P[n].detach().cpu().numpy().astype(np.float64) - z["final_" + n]).max() / lr
        dm = np.abs(st["exp_avg"].cpu().numpy() - z["final_m_" + n]).max() / np.abs(z["final_m_" + n]).max()
        dv = np.abs(st["exp_avg_sq"].cpu().numpy() - z["final_v_" + n]).max() / np.abs(z["final_v_" + n]).max()
        report.append((n, dp, dm, dv))
        if not (dp < 2e-3 and dm < 2e-5 and dv < 4e-5):
            bad.append(n)
    print("param error / lr, exp_avg rel, exp_avg_sq rel:", report)
    assert not bad, (bad, report)
