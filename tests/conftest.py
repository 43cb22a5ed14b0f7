import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_unavailable_reason():
    """None when the -m gpu tests can run: a ROCm device is visible and the hipcc-built engine is in the tree."""
    try:
        import torch
        if not torch.cuda.is_available():
            return "no ROCm device visible (run with `-m gpu` on the MI355X box)"
    except Exception as exc:  # pragma: no cover
        return f"torch unavailable: {exc}"
    lib = os.path.join(REPO, "dtqn_amd", "csrc", "libdtqn_hip.so")
    if not os.path.exists(lib):
        return f"{lib} is missing (python -m dtqn_amd.build)"
    return None


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a host without a GPU skips the gpu-marked tests instead of erroring in their fixtures.
    On a GPU box nothing is skipped: a missing engine there is a failure, not a skip (engine.get_lib raises)."""
    reason = _gpu_unavailable_reason()
    if reason is None or os.environ.get("DTQN_FORCE_GPU_TESTS") == "1":
        return
    try:
        import torch
        if torch.cuda.is_available():
            return                      # GPU present but engine missing: let the tests fail loudly
    except Exception:
        pass
    skip = pytest.mark.skip(reason=reason)
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
