"""-m gpu: the persistent-memory bag on the MI355X -- TD update vs the oracle, and the reference's own surface (DTQN.forward with
bag arguments, DtqnAgent.observe / get_action / train, sample_with_bag) vs the golden vectors generated from the reference
(tests/golden/G9_bag.npz)."""
import pytest

from oracle import dtqn_oracle as O

from helpers import make_td_case, check_td_updates
from test_bag_golden import NAMES, check_bag_surface, check_device_drawn_bags, check_vector_bag_rollout

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from dtqn_amd import engine
    engine.require_gpu()
    return engine.get_lib()


BAG_TD = [
    (dict(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50, bag_size=5), dict(batch=32, T=200, mask=-5, n_eps=40)),
    (dict(obs_dim=6, num_actions=5, inner_embed_size=64, num_heads=4, num_layers=1, history_len=70, discrete=True, vocab_sizes=9, action_dim=8,
          bag_size=7), dict(batch=4, T=90, mask=8, tuf=2)),
    (dict(obs_dim=3, num_actions=4, inner_embed_size=64, num_heads=2, num_layers=1, history_len=12, action_dim=4, bag_size=12, gate="gru"),
     dict(batch=3, T=20, mask=-5, history=6)),
    (dict(obs_dim=10, num_actions=10, inner_embed_size=128, num_heads=8, history_len=50, discrete=True, vocab_sizes=9, bag_size=10),
     dict(batch=8, T=50, mask=8, n_eps=20)),
    (dict(obs_dim=6, num_actions=6, inner_embed_size=128, num_heads=8, history_len=128, discrete=True, vocab_sizes=12, bag_size=30, action_dim=8),
     dict(batch=3, T=140, mask=11, n_eps=8, history=40)),
    (dict(obs_dim=1, num_actions=5, inner_embed_size=256, num_heads=8, history_len=100, discrete=True, vocab_sizes=22, num_layers=1, bag_size=64),
     dict(batch=2, T=120, mask=21, n_eps=5)),
    # identity-reordered layers; dropout (incl. the bag attention's own)
    (dict(obs_dim=3, num_actions=3, inner_embed_size=128, num_heads=8, history_len=50, bag_size=8, identity=True, pos="sin"), dict(batch=6, T=80, mask=-5, n_eps=12)),
    (dict(obs_dim=6, num_actions=5, inner_embed_size=64, num_heads=4, history_len=70, discrete=True, vocab_sizes=9, action_dim=8, bag_size=7, dropout=0.2),
     dict(batch=4, T=90, mask=8, tuf=2)),
]


@pytest.mark.parametrize("kw,run", BAG_TD)
def test_td_update_with_a_bag_vs_oracle(lib, kw, run):
    cfg = O.NetCfg(**kw)
    net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=35, batch=run["batch"], T=run["T"], n_eps=run.get("n_eps", 9), mask=run["mask"],
                                               history=run.get("history"), tuf=run.get("tuf", 10_000), device="cuda", test_lib=False)
    assert net.tiled == 1 and net.bag_size == cfg.bag_size
    eng.td.dropout_seed = 777
    check_td_updates(cfg, net, oracle, host, eng, rep, n_updates=3)


@pytest.mark.parametrize("name", NAMES)
def test_bag_surface_vs_the_reference(lib, name):
    check_bag_surface(None, name, device="cuda")


def test_device_drawn_bags(lib):
    check_device_drawn_bags(None, device="cuda")


@pytest.mark.parametrize("name", NAMES)
def test_vectorised_bag_rollout_on_the_device_vs_the_reference(lib, name):
    """The vectorised bag rollout (agents/vector.py: one batched dtqn_forward_bag per vector step, last rows through pinned memory
    behind an event) against G9's greedy rollout with bag evictions."""
    check_vector_bag_rollout(None, name, device="cuda")
