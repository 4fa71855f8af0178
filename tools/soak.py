#!/usr/bin/env python
"""Extended random parity soak: tests/test_gpu_raster.py::test_random_configurations beyond its 40 committed seeds.
usage (GPU box): python tools/soak.py FIRST LAST   -> one line per failure, summary at the end"""
import os
import sys
import traceback

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "skyfall-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import test_gpu_raster as T  # noqa: E402

first, last = int(sys.argv[1]), int(sys.argv[2])
bad = []
for i in range(first, last):
    try:
        T.test_random_configurations(i)
    except Exception as e:  # noqa: BLE001
        bad.append(i)
        print(f"FAIL {i}: {T._random_config(i)}\n{traceback.format_exc(limit=2)}", flush=True)
print(f"soak {first}..{last}: {last - first - len(bad)} passed, {len(bad)} failed {bad}")
