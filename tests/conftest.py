import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# the kernel-selection switches ($VIDIL_GEMM4W ...) are cached per process by the library unless this is set (core.hip):
# several tests flip them between launches
os.environ.setdefault("VIDIL_DEV_ENV", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build the HIP library (hipcc cross-compiles for
    gfx950 without a GPU) and the oracle's C part once, exactly as ``__graft_entry__.build()`` does."""
    lib = os.path.join(ROOT, "vidil_amd", "csrc", "libvidil_hip.so")
    ref = os.path.join(ROOT, "oracle", "_build", "libscan_ref.so")
    if not (os.path.exists(lib) and os.path.exists(ref)):
        import __graft_entry__

        __graft_entry__.build()


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
