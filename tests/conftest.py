import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the shared libraries are build products (git-ignored): in a fresh checkout build them once, exactly as
    # __graft_entry__.build() does (hipcc cross-compiles for gfx950 without a GPU; gcc for the C oracle)
    import subprocess
    for sub, so in (("star-gcn_amd/csrc", "libstargcn_hip.so"), ("oracle", "libseg_oracle.so")):
        if not os.path.exists(os.path.join(ROOT, sub, so)):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, sub), "-j8"], stdout=subprocess.DEVNULL)


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "seg_ops_golden.npz"))
