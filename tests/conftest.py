import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The fused weight-gradient + Adam launch of dtqn_td_update has a grid-wide barrier: every workgroup must be alive at
    # once.  The CPU emulation runs a handful of workgroups at a time, so without a GPU the one-call update takes the
    # unfused launches unless a test asks for the fused one (HIPEMU_COOP + DTQN_FUSED_ADAM=1, tests/test_emu_td.py).
    import torch
    if not torch.cuda.is_available():
        os.environ.setdefault("DTQN_FUSED_ADAM", "0")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
