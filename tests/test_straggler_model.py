"""tools/straggler_model.py: the arithmetic of "eight scenes of different size on eight GPUs" -- pure host
code, runs everywhere. The model's inputs are measured single-GPU step times; what is pinned here is the model itself."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("straggler_model", os.path.join(ROOT, "tools", "straggler_model.py"))
sm = importlib.util.module_from_spec(spec)
spec.loader.exec_module(sm)


def test_equal_scenes_scale_by_the_world_size_and_the_straggler_sets_the_pace():
    t = sm.make_t({500_000: 0.4, 1_000_000: 0.6, 2_000_000: 1.0})
    assert abs(t(750_000) - 0.5) < 1e-12 and abs(t(1_500_000) - 0.8) < 1e-12 and abs(t(250_000) - 0.3) < 1e-12   # piecewise linear
    p = sm.predict([2_000_000] * 8, t)
    assert abs(p["speedup_independent"] - 8.0) < 1e-12 and abs(p["bench_value_gaussians_per_s"] - 16e6 / 1e-3) < 1e-3
    p = sm.predict([2_000_000] * 8, t, allreduce_ms=0.25)          # lock step pays the collective on every step
    assert abs(p["speedup_lockstep"] - 8.0 / 1.25) < 1e-12
    p = sm.predict([2_000_000] + [500_000] * 7, t)                 # one big scene among seven small ones
    assert p["straggler"] == 0 and abs(p["speedup_independent"] - (1.0 + 7 * 0.4) / 1.0) < 1e-12
    assert abs(p["bench_value_gaussians_per_s"] - 5.5e6 / 1e-3) < 1e-3


def test_critical_spread_is_where_mean_over_max_reaches_three_quarters():
    t = sm.make_t({500_000: 0.4, 2_000_000: 1.0})
    s = sm.critical_spread(t, 2_000_000, target=6.0)
    p = sm.predict(sm.spread_sizes(2_000_000, s), t)
    assert abs(p["speedup_independent"] - 6.0) < 1e-3 and abs(p["mean_over_max_t"] - 0.75) < 1e-3
    assert sm.predict(sm.spread_sizes(2_000_000, s + 0.05), t)["speedup_independent"] < 6.0
    # a step time with a large fixed part (small scenes are host- / launch-bound) tolerates a wider spread
    t_flat = sm.make_t({500_000: 0.9, 2_000_000: 1.0})
    assert sm.critical_spread(t_flat, 2_000_000) > s


def test_cli_reports_the_default_sweep():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "straggler_model.py"), "--json"], capture_output=True,
                       text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout)
    assert d["equal_scenes"]["speedup_independent"] == 8.0
    assert 0.0 < d["critical_spread_lockstep"] <= d["critical_spread_independent"] < 1.0
    eight = d["the_eight_sizes_0.5M_to_2M"]
    assert eight["sizes"][0] == 2_000_000 and eight["sizes"][-1] == 500_000 and eight["speedup_independent"] < 6.0
