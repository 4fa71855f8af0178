"""tools/launch_scenes.py on the GPU with the FUSED hooks installed AND the shared-MLP protocol on (the
CPU launcher test runs --no-fused): two worker processes share the one GPU of the test box (--gpu-ids 0,0,
SFGS_DIST_BACKEND=gloo: RCCL cannot put two ranks on one device), the launcher installs prepass / filter3d /
densify_stats / FusedAdam / compact / densify on the model class and wraps training_setup for the shared appearance MLP.
/root/reference does not exist on the GPU box, so the class is a stand-in with the reference's method names
(scene/gaussian_model.py:350-392 training_setup: the same nine parameter groups). Checked: FusedAdam is the optimizer
that steps, the MLPs stay in bit-identical lock step while both scenes train, the longer scene finishes alone, and a
second training_setup (IDU episode, train.py:633) re-synchronises parameters and Adam moments."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MODEL = '''
import torch
from torch import nn


class GaussianModel:
    appearance_enabled = True
    max_sh_degree = 1
    active_sh_degree = 1

    def __init__(self, n, seed):
        g = torch.Generator().manual_seed(seed)
        dev = "cuda"
        r = lambda *s: torch.randn(*s, generator=g).to(dev)
        self._xyz = nn.Parameter(r(n, 3)); self._features_dc = nn.Parameter(r(n, 1, 3))
        self._features_rest = nn.Parameter(r(n, 3, 3)); self._opacity = nn.Parameter(r(n, 1))
        self._scaling = nn.Parameter(r(n, 3) - 2); self._rotation = nn.Parameter(r(n, 4))
        self._embeddings = nn.Parameter(r(n, 24)); self.appearance_embeddings = nn.Parameter(r(3, 32))
        torch.manual_seed(100 + seed)      # DIFFERENT initial MLPs per scene
        self.appearance_mlp = nn.Sequential(nn.Linear(59, 128), nn.ReLU(), nn.Linear(128, 128), nn.ReLU(),
                                            nn.Linear(128, 6)).to(dev)     # scene/gaussian_model.py:52-58
        self.filter_3D = torch.zeros(n, 1, device=dev)
        self.max_radii2D = torch.zeros(n, device=dev)

    get_xyz = property(lambda self: self._xyz)
    get_scaling_with_3D_filter = property(lambda self: torch.exp(self._scaling))
    get_opacity_with_3D_filter = property(lambda self: torch.sigmoid(self._opacity))
    get_rotation = property(lambda self: torch.nn.functional.normalize(self._rotation))

    def compute_3D_filter(self, cameras): pass
    def add_densification_stats(self, v, f): pass
    def prune_points(self, mask): pass
    def densify_and_prune(self, *a): pass

    def training_setup(self, training_args=None, num_train_cameras=0, from_scratch=True):   # :350-392
        l = [{"params": [self._xyz], "lr": 0.00016, "name": "xyz"}, {"params": [self._features_dc], "lr": 0.0025, "name": "f_dc"},
             {"params": [self._features_rest], "lr": 0.000125, "name": "f_rest"}, {"params": [self._opacity], "lr": 0.05, "name": "opacity"},
             {"params": [self._scaling], "lr": 0.005, "name": "scaling"}, {"params": [self._rotation], "lr": 0.001, "name": "rotation"},
             {"params": [self.appearance_embeddings], "lr": 0.001, "name": "appearance_embeddings", "weight_decay": 0.01},
             {"params": [self._embeddings], "lr": 0.005, "name": "embeddings"},
             {"params": list(self.appearance_mlp.parameters()), "lr": 0.0005, "name": "appearance_mlp"}]
        self.optimizer = torch.optim.Adam(l, lr=0.0, eps=1e-15)
'''

TRAIN = '''
import sys
import numpy as np
import torch
import gm_standin
scene, plan, out = sys.argv[1], PLANS[sys.argv[1]], sys.argv[2]
rank = {"A": 0, "B": 1}[scene]
m = gm_standin.GaussianModel(500, rank)
flat = lambda: torch.cat([p.detach().reshape(-1) for p in m.appearance_mlp.parameters()]).cpu().numpy().copy()
hist, k = [], 0
for ep, steps in enumerate(plan):
    m.training_setup(None, num_train_cameras=3, from_scratch=(ep == 0))
    assert type(m.optimizer).__name__ == "FusedAdam", type(m.optimizer)       # the launcher's fused hook re-homed it
    hist.append(flat())
    for _ in range(steps):
        for gi, group in enumerate(m.optimizer.param_groups):
            for i, p in enumerate(group["params"]):
                gg = torch.Generator().manual_seed(100000 * rank + 1000 * k + 10 * gi + i)
                p.grad = (torch.randn(*p.shape, generator=gg) * 1e-2).to(p.device)
        m.optimizer.step()
        m.optimizer.zero_grad(set_to_none=True)
        hist.append(flat())
        k += 1
xyz = m._xyz.detach().cpu().numpy()
assert np.isfinite(xyz).all()
np.save(out, np.stack(hist))
'''


def test_launcher_with_fused_hooks_and_shared_mlp_on_the_gpu(tmp_path):
    ref = tmp_path / "ref"
    ref.mkdir()
    (ref / "gm_standin.py").write_text(textwrap.dedent(MODEL))
    (ref / "fake_train.py").write_text("PLANS = {'A': [3], 'B': [4, 3]}\n" + textwrap.dedent(TRAIN))
    env = dict(os.environ, SFGS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "launch_scenes.py"), "--reference", str(ref), "--gpus", "2",
           "--gpu-ids", "0,0", "--scenes", "A", "B", "--shared-mlp", "--model-module", "gm_standin", "--",
           "fake_train.py", "{scene}", str(tmp_path / "{scene}.npy")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    A, B = np.load(str(tmp_path / "A.npy")), np.load(str(tmp_path / "B.npy"))
    assert A.shape == (4, 24966) and B.shape == (9, 24966)
    # rounds: 1 A.setup / B.setup (rank 0 seeds) . 2-4 both step . 5 rank 0 drains / B steps alone . 6 B.setup again (alone:
    # nobody trains, it keeps its own parameters) . 7-9 B steps alone
    for k in range(4):
        np.testing.assert_array_equal(A[k], B[k], err_msg=f"step {k}")
    assert not np.array_equal(B[3], B[4]) and np.array_equal(B[4], B[5]) and not np.array_equal(B[5], B[6])
    assert r.stdout.count("answered") == 2
