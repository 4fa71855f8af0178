"""GPU-less dry run of tests/ref_entry_driver.py: the reference's REAL train.training() (60 iterations: Scene + Satellite
loader, losses, one densify_and_prune, checkpoint capture -> restore) and render_video.render_sets() executed here, with the
rasterizer's backend seam swapped for the C oracle double and the reference's hard-coded "cuda" allocations redirected to
the CPU (the --backend oracle mode; see the driver's docstring). It pins the HARNESS -- scene writer, stand-ins for the
absent third-party packages, assertions -- so that the GPU run (tests/test_gpu_reference_entry.py) can only fail on the HIP
path. Needs the reference tree (authoring container only)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isfile("/root/reference/train.py"), reason="reference tree not present (GPU box)")
def test_real_training_and_render_sets_run_on_the_oracle_double(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ref_entry_driver.py"), "--backend", "oracle", "--work",
                        str(tmp_path / "work"), "--iters", "60"], capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0 and "REF-ENTRY OK" in r.stdout, r.stdout[-3000:] + "\n--- stderr ---\n" + r.stderr[-3000:]
    for stage in ('"stage": "training"', '"stage": "restore"', '"stage": "render_sets"', '"stage": "fused_ply_video"', '"stage": "idu_pseudo_cameras"'):
        assert stage in r.stdout
