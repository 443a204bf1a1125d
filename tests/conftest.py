import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")
# hydro_debug_ke_fault (tests/test_error_paths_gpu.py) is refused unless this is set WHEN libhydro.so IS LOADED - which is
# once per process, by whichever test first makes an engine: hence here and not in that test
os.environ.setdefault("HYDRO_ENABLE_TEST_HOOKS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    return {k: z[k] for k in z.files}


def accel_of(fx):
    if "accel" in fx:
        return fx["accel"]
    return (fx["state"][:, 7:13].astype(np.float64) - fx["prev"].astype(np.float64)) / float(fx["dt"])


@pytest.fixture(scope="session")
def native_built():
    """Build (if stale) every native piece once per session; returns the repo root."""
    import __graft_entry__ as entry
    entry.build()
    return REPO


SCENE_FIXTURES = ["c2", "c3", "c4", "c5", "c4_adversarial", "ties"]
