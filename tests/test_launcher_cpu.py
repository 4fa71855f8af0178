"""tools/launch_scenes.py (our replacement of the reference's process-per-scene farm, scripts/run_jax.py:52-87)
over gloo with world size 2, against the REAL GaussianModel class: the launcher's worker patches
GaussianModel.training_setup (scene/gaussian_model.py:350-382) so that the appearance MLP starts from rank 0's
initialisation and optimizer.step() first averages its gradients with one all-reduce; scenes of different length
finish independently (participation count + drain). Needs the reference tree (authoring container only): the product
code under test is tools/launch_scenes.py + sfgs/shard.py, the reference supplies the class being patched."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

SHIM = '''
import importlib.util, os, sys, types
sys.path.insert(0, {golden!r})
import make_golden as mg
mg._cpu_redirect()
sys.path.insert(0, {ref!r})
for name in ("plyfile", "simple_knn", "simple_knn._C"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
sys.modules["simple_knn._C"].distCUDA2 = None
_m = mg._load(os.path.join({ref!r}, "scene", "gaussian_model.py"), "ref_gaussian_model")
GaussianModel = _m.GaussianModel
'''

TRAIN = '''
import sys, types
import numpy as np
import torch
from torch import nn
import gm_shim
scene, plan, out = sys.argv[1], PLANS[sys.argv[2]], sys.argv[3]
rank = globals().get("RANKS", {"A": 0, "B": 1, "C": 2})[scene]
torch.manual_seed(100 + rank)          # DIFFERENT initial MLPs per scene: the launcher must broadcast rank 0's
orig_to = nn.Module.to
nn.Module.to = lambda self, *a, **k: self if (a and str(a[0]).startswith("cuda")) else orig_to(self, *a, **k)
m = gm_shim.GaussianModel(1, appearance_enabled=True, appearance_n_fourier_freqs=4, appearance_embedding_dim=32)
nn.Module.to = orig_to
n = 16
g = torch.Generator().manual_seed(rank)
m._xyz = nn.Parameter(torch.randn(n, 3, generator=g)); m._features_dc = nn.Parameter(torch.randn(n, 1, 3, generator=g))
m._features_rest = nn.Parameter(torch.randn(n, 3, 3, generator=g)); m._opacity = nn.Parameter(torch.randn(n, 1, generator=g))
m._scaling = nn.Parameter(torch.randn(n, 3, generator=g)); m._rotation = nn.Parameter(torch.randn(n, 4, generator=g))
m._embeddings = nn.Parameter(torch.randn(n, 24, generator=g)); m.max_radii2D = torch.zeros(n); m.spatial_lr_scale = 1.0
args = types.SimpleNamespace(percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016,
    position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005,
    rotation_lr=0.001, embedding_lr=0.005, appearance_embedding_lr=0.001, appearance_embedding_regularization=0.01,
    appearance_mlp_lr=0.0005, idu_position_lr_max_steps=10000)
flat = lambda: torch.cat([p.detach().reshape(-1) for p in m.appearance_mlp.parameters()]).numpy().copy()
hist, k = [], 0
for ep, steps in enumerate(plan):      # a second entry = training_setup again (the next IDU episode: train.py:633)
    m.training_setup(args, num_train_cameras=3, from_scratch=(ep == 0))
    hist.append(flat())
    for _ in range(steps):
        for i, p in enumerate(m.appearance_mlp.parameters()):
            gg = torch.Generator().manual_seed(1000 * rank + 10 * k + i)
            p.grad = torch.randn(*p.shape, generator=gg) * 1e-2
        m.optimizer.step()
        m.optimizer.zero_grad(set_to_none=True)
        hist.append(flat())
        k += 1
np.save(out, np.stack(hist))
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference tree (authoring container)")
def test_launcher_keeps_the_shared_mlp_in_step_and_lets_short_scenes_finish(tmp_path):
    ref = tmp_path / "ref"
    ref.mkdir()
    (ref / "gm_shim.py").write_text(SHIM.format(golden=os.path.join(ROOT, "tests", "golden"), ref=REF))
    outs = {s: str(tmp_path / f"{s}.npy") for s in "AB"}
    steps = {"A": 3, "B": 6}
    env = dict(os.environ, SFGS_TEST_OUT=str(tmp_path))
    # per-scene arguments through the {scene} placeholder; step counts differ -> the short scene must not block the long one
    cmd = [sys.executable, os.path.join(ROOT, "tools", "launch_scenes.py"), "--reference", str(ref), "--gpus", "2",
           "--scenes", "A", "B", "--no-fused", "--shared-mlp", "--model-module", "gm_shim", "--",
           "fake_train.py", "{scene}", "STEPS_{scene}", str(tmp_path / "{scene}.npy")]
    # STEPS_{scene} is resolved by a tiny wrapper: substitute before launching (the launcher only knows {scene})
    (ref / "fake_train.py").write_text("PLANS = {'STEPS_A': [3], 'STEPS_B': [6]}\n" + textwrap.dedent(TRAIN))
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    A, B = np.load(outs["A"]), np.load(outs["B"])
    assert A.shape == (steps["A"] + 1, 24966) and B.shape == (steps["B"] + 1, 24966)
    # broadcast: scene B (seeded differently) starts from rank 0's MLP; lock step while both train
    np.testing.assert_array_equal(A[0], B[0])
    for k in range(steps["A"] + 1):
        np.testing.assert_array_equal(A[k], B[k], err_msg=f"step {k}")
    assert not np.array_equal(B[steps["A"]], B[steps["A"] + 1])     # the long scene keeps training on its own
    assert "answered" in r.stdout                                    # the short scene drained instead of hanging
    # one-process cross-check of the first step: Adam on the MEAN of the two ranks' gradients
    import torch
    shapes = [(128, 59), (128,), (128, 128), (128,), (6, 128), (6,)]
    grads = []
    for rank in (0, 1):
        grads.append(torch.cat([torch.randn(*s, generator=torch.Generator().manual_seed(1000 * rank + i)).reshape(-1) * 1e-2
                                for i, s in enumerate(shapes)]))
    p = torch.nn.Parameter(torch.from_numpy(A[0]).clone())
    opt = torch.optim.Adam([p], lr=0.0005, eps=1e-15)
    p.grad = (grads[0] + grads[1]) / 2
    opt.step()
    np.testing.assert_allclose(p.detach().numpy(), A[1], rtol=0, atol=1e-9)


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference tree (authoring container)")
def test_three_scenes_on_two_ranks_with_a_repeated_training_setup(tmp_path):
    """the reference calls training_setup repeatedly (train.py:95, restore(), every IDU episode) and a
    rank may train several scenes. Rank 0 trains A (3 steps) then C (3 steps); rank 1 trains B: 4 steps, training_setup
    again, 4 steps. Every collective is the same all-reduce, so the rounds pair in order:
      1 A.setup / B.setup (rank 0 seeds) . 2-4 A, B step together . 5 C.setup / B step 4 . 6 C step 1 / B.setup
      7-8 C, B step together . 9-10 rank 0 drains, B steps alone.
    A rank that sets up while another trains adopts its parameters AND Adam state and performs that round's step."""
    ref = tmp_path / "ref"
    ref.mkdir()
    (ref / "gm_shim.py").write_text(SHIM.format(golden=os.path.join(ROOT, "tests", "golden"), ref=REF))
    (ref / "fake_train.py").write_text("PLANS = {'A': [3], 'B': [4, 4], 'C': [3]}\n" + textwrap.dedent(TRAIN))
    cmd = [sys.executable, os.path.join(ROOT, "tools", "launch_scenes.py"), "--reference", str(ref), "--gpus", "2",
           "--scenes", "A", "B", "C", "--no-fused", "--shared-mlp", "--model-module", "gm_shim", "--",
           "fake_train.py", "{scene}", "{scene}", str(tmp_path / "{scene}.npy")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    A, B, C = (np.load(str(tmp_path / f"{s}.npy")) for s in "ABC")
    assert A.shape[0] == 4 and B.shape[0] == 10 and C.shape[0] == 4
    for k in range(4):
        np.testing.assert_array_equal(A[k], B[k], err_msg=f"A/B step {k}")
    np.testing.assert_array_equal(C[0], B[4])      # C joined in the round of B's step 4: same parameters afterwards
    np.testing.assert_array_equal(C[1], B[5])      # B's second training_setup adopted C's state and took C's step
    np.testing.assert_array_equal(C[2], B[6])      # ... including the Adam moments: the following steps agree bit for bit
    np.testing.assert_array_equal(C[3], B[7])
    assert not np.array_equal(B[7], B[8]) and not np.array_equal(B[8], B[9])   # B finishes alone while rank 0 drains
    assert r.stdout.count("answered") == 2


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference tree (authoring container)")
def test_default_is_independent_scenes(tmp_path):
    """Without --shared-mlp the scenes train independently, as the reference's farm does (scripts/run_jax.py:52-87):
    no process group, different MLPs."""
    ref = tmp_path / "ref"
    ref.mkdir()
    (ref / "gm_shim.py").write_text(SHIM.format(golden=os.path.join(ROOT, "tests", "golden"), ref=REF))
    (ref / "fake_train.py").write_text("PLANS = {'A': [2], 'B': [2]}\n" + textwrap.dedent(TRAIN))
    cmd = [sys.executable, os.path.join(ROOT, "tools", "launch_scenes.py"), "--reference", str(ref), "--gpus", "2",
           "--scenes", "A", "B", "--no-fused", "--model-module", "gm_shim", "--",
           "fake_train.py", "{scene}", "{scene}", str(tmp_path / "{scene}.npy")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    A, B = (np.load(str(tmp_path / f"{s}.npy")) for s in "AB")
    assert not np.array_equal(A[0], B[0]) and "answered" not in r.stdout


JAX_SCENES = ["JAX_004", "JAX_068", "JAX_164", "JAX_168", "JAX_175", "JAX_214", "JAX_260", "JAX_264"]   # data/README.md


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference tree (authoring container)")
def test_eight_scenes_on_eight_ranks(tmp_path):
    """N = 8 before a node exists: the farm the launcher replaces runs the 8 JAX scenes on 8 GPUs
    (scripts/run_jax.py:52-87). Eight ranks over gloo, one scene each, scenes of DIFFERENT length (2 .. 6 steps), the
    shared-MLP protocol on: every scene starts from rank 0's MLP, all eight stay in lock step while all eight train, every
    shorter scene drains instead of hanging, the longest finishes alone."""
    ref = tmp_path / "ref"
    ref.mkdir()
    (ref / "gm_shim.py").write_text(SHIM.format(golden=os.path.join(ROOT, "tests", "golden"), ref=REF))
    steps = dict(zip(JAX_SCENES, [2, 3, 4, 5, 2, 3, 4, 6]))
    prelude = ("PLANS = %r\nRANKS = %r\n" % ({k: [v] for k, v in steps.items()}, {k: i for i, k in enumerate(JAX_SCENES)}))
    (ref / "fake_train.py").write_text(prelude + textwrap.dedent(TRAIN))
    cmd = [sys.executable, os.path.join(ROOT, "tools", "launch_scenes.py"), "--reference", str(ref), "--gpus", "8",
           "--scenes", *JAX_SCENES, "--no-fused", "--shared-mlp", "--model-module", "gm_shim", "--",
           "fake_train.py", "{scene}", "{scene}", str(tmp_path / "{scene}.npy")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    H = {s: np.load(str(tmp_path / f"{s}.npy")) for s in JAX_SCENES}
    for s in JAX_SCENES:
        assert H[s].shape == (steps[s] + 1, 24966), (s, H[s].shape)
        np.testing.assert_array_equal(H[s][0], H[JAX_SCENES[0]][0], err_msg=f"{s}: did not start from rank 0's MLP")
    for k in range(min(steps.values()) + 1):                       # all eight train: lock step, bit for bit
        for s in JAX_SCENES[1:]:
            np.testing.assert_array_equal(H[s][k], H[JAX_SCENES[0]][k], err_msg=f"{s} step {k}")
    # while a subset still trains, the subset stays in lock step (the average is over the ranks training in that round)
    for k in range(min(steps.values()) + 1, 5):
        alive = [s for s in JAX_SCENES if steps[s] >= k]
        for s in alive[1:]:
            np.testing.assert_array_equal(H[s][k], H[alive[0]][k], err_msg=f"{s} step {k} (ranks alive: {len(alive)})")
    longest = max(steps, key=steps.get)
    assert not np.array_equal(H[longest][-2], H[longest][-1])
    assert r.stdout.count("answered") == 8                         # every rank drained at its end (the last one: 0 rounds)
    for i, s in enumerate(JAX_SCENES):
        assert f"[launch_scenes rank {i}] {s}:" in r.stdout        # scene i ran on rank i, one scene per rank


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference tree (authoring container)")
def test_eight_independent_scenes_need_no_process_group(tmp_path):
    """The default (what the reference's farm does): 8 processes, no collective at all, 8 different MLPs."""
    ref = tmp_path / "ref"
    ref.mkdir()
    (ref / "gm_shim.py").write_text(SHIM.format(golden=os.path.join(ROOT, "tests", "golden"), ref=REF))
    prelude = ("PLANS = %r\nRANKS = %r\n" % ({k: [1] for k in JAX_SCENES}, {k: i for i, k in enumerate(JAX_SCENES)}))
    (ref / "fake_train.py").write_text(prelude + textwrap.dedent(TRAIN))
    cmd = [sys.executable, os.path.join(ROOT, "tools", "launch_scenes.py"), "--reference", str(ref), "--gpus", "8",
           "--scenes", *JAX_SCENES, "--no-fused", "--model-module", "gm_shim", "--",
           "fake_train.py", "{scene}", "{scene}", str(tmp_path / "{scene}.npy")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    first = [np.load(str(tmp_path / f"{s}.npy"))[0] for s in JAX_SCENES]
    assert all(not np.array_equal(first[0], f) for f in first[1:]) and "answered" not in r.stdout
