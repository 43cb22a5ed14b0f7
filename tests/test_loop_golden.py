"""G12: the coupled actor / learner loop against THE REFERENCE's own run.py (tests/golden/make_golden.py gen_G12 drove
/root/reference/run.py's set_global_seed / get_agent / prepopulate / train / step / evaluate on the reference's CarFlag).
Here dtqn_amd's run.py drives the same loop with `--sampler reference --ref-quirks` semantics and the kernels on the test-only
HIP emulation; the -m gpu twin is tests/test_gpu_loop_golden.py."""
import os

import numpy as np
import pytest
import torch

from dtqn_amd import _binding as B

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G12_loop.npz")


@pytest.fixture(scope="module")
def emu():
    from emu import emu_build
    return B.load_library(emu_build.build())


@pytest.fixture(scope="module")
def fx():
    return dict(np.load(GOLDEN))


def test_small_loop_reproduces_the_reference_trace(emu, fx):
    from loop_harness import run_loop, compare_loop
    tr, agent, prepop = run_loop(fx, "small", torch.device("cpu"), test_lib=emu)
    s = compare_loop(fx, "small", tr, prepop, min_actions=150)
    print(s)
    assert s["updates_compared"] >= 50
