import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skyfall-gs_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")
    # No test of this suite takes more than a few minutes. A hang (round 6: a rank-0-only block of bench.py that issued
    # collectives the other ranks never matched) must cost ONE test its verdict, not the whole GPU session its budget:
    # with pytest-timeout installed (this image), every test gets a ceiling unless the command line sets its own.
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 1500


def pytest_collection_modifyitems(config, items):
    # `-m gpu` tests must never silently pass without a GPU
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def sfgs_option():
    """set a process-wide route option of libsfgs.so (sfgs_set_option) for the duration of one test:
    `sfgs_option("sort", "split")`. Every option goes back to what it was when the test ends."""
    from sfgs import _lib as L
    saved = {}

    def set_(key, value):
        old = L.set_option(key, value)
        saved.setdefault(key, old)
    yield set_
    for key, old in saved.items():
        L.set_option(key, old)
