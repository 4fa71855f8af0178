"""The synthetic scene generators behind the regime sweep and the full-size parity cases (CPU: checked with the oracle)."""
import numpy as np
import torch

from oracle import oracle as orc
from sfgs.synth import city_scene, morton_order, orbit_scene, scene


def _finite(g):
    return all(torch.isfinite(v).all() for v in g.values() if v is not None)


def test_orbit_and_city_scenes_are_in_view_and_opaque_where_expected():
    for elev in (85.0, 45.0, 25.0):
        frame, g = orbit_scene(4000, 240, 135, elev, seed=1)
        assert _finite(g)
        R = orc.OracleRender(frame, **g)
        assert (R.radii > 0).mean() > 0.8            # the slab sits in front of the orbit camera
        frame, g = city_scene(30000, 240, 135, elev, seed=1, splat=(1.5, 4.0))
        assert _finite(g) and g["means3D"][:, 2].min() >= 0.0
        R = orc.OracleRender(frame, **g)
        assert (R.radii > 0).mean() > 0.8
        covered = R.alpha[0] > 0.5
        assert covered.mean() > 0.2                  # opaque surfaces: a good part of the frame saturates
        # saturating pixels stop early: their last contributor is far from the end of their tile's list
        assert np.median(R.n_contrib().reshape(covered.shape)[covered]) < 0.7 * R.max_tile_list


def test_morton_order_is_a_permutation_that_groups_neighbours():
    _, g = scene(5000, 320, 180, seed=3)
    perm = morton_order(g["means3D"])
    assert sorted(perm.tolist()) == list(range(5000))
    d = g["means3D"][:, :2] / g["means3D"][:, 2:3]
    step_sorted = (d[perm][1:] - d[perm][:-1]).norm(dim=1).mean()
    step_stored = (d[1:] - d[:-1]).norm(dim=1).mean()
    assert step_sorted < 0.2 * step_stored
