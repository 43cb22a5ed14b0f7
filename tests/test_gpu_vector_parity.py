"""-m gpu: dtqn_actor_forward_batch against the oracle on the MI355X (see test_vector_parity.py)."""
import pytest

from test_vector_parity import CASES, check_batch_argument_errors, check_batched_actor_vs_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from dtqn_amd import engine
    engine.require_gpu()
    return engine.get_lib()


@pytest.mark.parametrize("kw,sizes", CASES)
def test_batched_actor_forward_vs_oracle_on_the_gpu(lib, kw, sizes):
    from dtqn_amd import engine
    worst = check_batched_actor_vs_oracle(lib, kw, sizes, device="cuda", stream=engine.stream_ptr())
    assert worst <= 1e-4


def test_batched_actor_argument_checks_on_the_gpu(lib):
    from dtqn_amd import engine
    check_batch_argument_errors(lib, device="cuda", stream=engine.stream_ptr())
