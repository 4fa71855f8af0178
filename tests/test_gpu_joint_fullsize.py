"""BASELINE.json configs[4] AT SIZE against the oracle (VERDICT r4 "missing" item 4): eight 2 M-Gaussian scenes
(sfgs.synth.scene, seeds 0..7 -- SURVEY 8d cfg 5) concatenated into one 16 M-Gaussian frame at 1920x1080, rendered the two
ways the joint renderer shards it (sfgs.shard, ranks emulated in this process; the real-process versions on small scenes:
tests/test_gpu_joint_render.py):

  (A) replicated Gaussians, band-sharded: rank r of 8 renders its band of 8-pixel tile rows (settings.tile_rows);
  (B) SHARDED Gaussians: rank r holds only scene r, plans it for the whole frame, the exported records / coarse items are
      merged (sfgs_raster_plan_export / _merge) and every rank composites its band.

Until round 5 these routes were only compared with the single-pass HIP frame (HIP vs HIP), and the single pass was pinned
to the oracle up to 5 M Gaussians. Here both assembled frames are compared with the C ORACLE on two bands of 16 pixel rows
(oracle/sfgs_oracle.c: orc_forward_rows -- all 16 M Gaussians projected, lists and pixels for those tile rows only; the
full frame would take minutes): radii of all 16 M bit for bit, the bands' RGB / depth / alpha within the parity bars."""
import os

import numpy as np
import pytest
import torch

import parity
from oracle import oracle as orc
from sfgs.synth import scene

pytestmark = pytest.mark.gpu
W, H, SCENES, N_EACH = 1920, 1080, 8, 2_000_000
ORACLE_TILE_ROWS = ((20, 21), (51, 52))          # rows of 16x16 tiles: pixel rows 320..335 and 816..831


def test_joint_frame_of_eight_2M_scenes_matches_the_oracle_on_two_bands():
    from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer
    from sfgs import shard
    os.environ.setdefault("OMP_NUM_THREADS", str(os.cpu_count() or 1))
    dev = torch.device("cuda:0")
    parts_cpu, frame = [], None
    for r in range(SCENES):
        frame, g = scene(N_EACH, W, H, seed=r)
        parts_cpu.append(g)
    cat = {k: (torch.cat([p[k] for p in parts_cpu]).contiguous() if parts_cpu[0][k] is not None else None)
           for k in parts_cpu[0]}
    N = cat["means3D"].shape[0]
    assert N == SCENES * N_EACH
    settings = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=frame["tanfovx"], tanfovy=frame["tanfovy"], kernel_size=frame["kernel_size"],
        subpixel_offset=None, bg=frame["bg"].to(dev), scale_modifier=1.0, viewmatrix=frame["view"].to(dev),
        projmatrix=frame["proj"].to(dev), sh_degree=0, campos=frame["campos"].to(dev), prefiltered=False, debug=False)
    inputs = dict(means3D=cat["means3D"].to(dev), means2D=None, opacities=cat["opacities"].to(dev),
                  colors_precomp=cat["colors_precomp"].to(dev), scales=cat["scales"].to(dev),
                  rotations=cat["rotations"].to(dev))
    with torch.no_grad():
        # (A) every rank holds all 16 M Gaussians and renders its band
        A = torch.zeros(5, H, W, device=dev)
        radii = None
        for r in range(SCENES):
            t0, t1, a, b = shard.band_rows(H, SCENES, r)
            c, d, _, al, radii, _ = GaussianRasterizer(settings._replace(tile_rows=(t0, t1)))(**inputs)
            A[:, a:b] = torch.cat([c, d, al], 0)[:, a:b]
        # (B) rank r holds scene r only
        def mine(r):
            s = slice(r * N_EACH, (r + 1) * N_EACH)
            return {k: (v[s].contiguous() if v is not None else None) for k, v in inputs.items()}
        probe = [shard.plan_export(settings, mine(r), export_capacity=None) for r in range(SCENES)]
        C = max(p["max_coarse"] for p in probe)
        del probe
        parts = [shard.plan_export(settings, mine(r), export_capacity=C) for r in range(SCENES)]
        assert sum(p["N"] for p in parts) == N
        Bf = torch.zeros(5, H, W, device=dev)
        for r in range(SCENES):
            t0, t1, a, b = shard.band_rows(H, SCENES, r)
            c, d, al = shard.render_merged_parts(parts, settings._replace(tile_rows=(t0, t1)))
            Bf[:, a:b] = torch.cat([c, d, al], 0)[:, a:b]
    # the two shardings assemble the same frame bit for bit (route equality) ...
    assert torch.equal(torch.nan_to_num(A, nan=-1.0), torch.nan_to_num(Bf, nan=-1.0))
    A = A.cpu().numpy()
    radii = radii.cpu().numpy()
    # ... and that frame is the oracle's, on two bands
    report = []
    for row0, row1 in ORACLE_TILE_ROWS:
        R = orc.OracleRender(frame, cat["means3D"], cat["scales"], cat["rotations"], cat["opacities"],
                             colors_precomp=cat["colors_precomp"], tile_rows=(row0, row1))
        np.testing.assert_array_equal(radii, R.radii)                       # all 16 M, bit for bit
        rows = slice(16 * row0, min(16 * row1, H))
        for name, got, ref in (("color", A[0:3, rows], R.color[:, rows]), ("depth", A[3:4, rows], R.depth[:, rows]),
                               ("alpha", A[4:5, rows], R.alpha[:, rows])):
            rep = parity.assert_image_close(f"{name}[rows {rows.start}:{rows.stop}]", got, ref, borderline_min=4)
            report.append(rep)
        assert R.num_duplicates > 100_000 and float(R.alpha[:, rows].mean()) > 0.5   # a band with content: long lists
        R.close()
    print(report)
    root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        import json
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "joint_fullsize_parity.json"), "w") as f:
            json.dump(dict(N=N, W=W, H=H, export_capacity=int(C), bands=[list(t) for t in ORACLE_TILE_ROWS], report=report), f)
    except OSError:
        pass
