"""The replay PRODUCER under random call sequences, on the test-only emulation, against oracle/replay_oracle.py (itself pinned to the
reference buffer by G5): store_obs / store / flush in any legal order with commits at arbitrary points -- episodes of every length incl.
the maximum, slots reused after the ring wraps, an episode abandoned before its first store, several episodes inside one scatter launch
(dtqn_replay_apply: runs of stores found by all threads at once, ep_len written by the last store of a slot) and one episode split over
several launches.  Bit-exact arrays, episode lengths and write position after every sequence."""
import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from dtqn_amd import _binding as B
from oracle import replay_oracle as RO


@pytest.fixture(scope="module")
def emu():
    from emu import emu_build
    return B.load_library(emu_build.build())


def _episodes():
    # (length, commit-after-step set as a bitmask seed, abandon before the first store?)
    return st.lists(st.tuples(st.integers(0, 6), st.integers(0, 2 ** 16 - 1), st.booleans()), min_size=1, max_size=14)


@settings(max_examples=300, deadline=None, derandomize=True)
@given(_episodes(), st.sampled_from([(3, -5.0, False), (2, 7.0, True)]), st.integers(0, 2 ** 31 - 1))
def _run(emu, episodes, kind, seed):
    from dtqn_amd.buffers.replay_buffer import ReplayBuffer
    O, mask, discrete = kind
    T, E = 6, 4                                               # at most 6 steps per episode, 4 slots: the ring wraps after 4 episodes
    rb = ReplayBuffer(E * T, O, mask, T, context_len=3, device="cpu", lib=emu)
    shadow = RO.ReplayOracle(E * T, O, mask, T, 3)
    rng = np.random.default_rng(seed)
    draw = (lambda: rng.integers(0, 7, O).astype(np.float32)) if discrete else (lambda: rng.uniform(-1, 1, O).astype(np.float32))
    for length, commits, abandon in episodes:
        o = draw()
        rb.store_obs(o); shadow.store_obs(o)
        if commits & 1:
            rb.commit()
        if abandon or length == 0:
            # the reference's loop never abandons a slot, but store_obs twice in a row is legal for the buffer: the slot is cleansed again
            continue
        for t in range(length):
            o, a, r, d = draw(), int(rng.integers(0, 4)), float(rng.choice([0.0, 1.0, -1.0])), t == length - 1
            rb.store(o, a, r, d, t + 1); shadow.store(o, a, r, d, t + 1)
            if (commits >> (t + 1)) & 1:
                rb.commit()
        rb.flush(); shadow.flush()
        if (commits >> 8) & 1:
            rb.commit_finished()
    arrays = rb.export_arrays()                               # commits what is still staged
    assert np.array_equal(arrays["obss"], shadow.obss)
    assert np.array_equal(arrays["actions"], shadow.actions[:, :, 0])
    assert np.array_equal(arrays["rewards"], shadow.rewards[:, :, 0])
    assert np.array_equal(arrays["dones"].astype(bool), shadow.dones[:, :, 0])
    assert np.array_equal(arrays["eplens"], shadow.episode_lengths)
    assert np.array_equal(rb.dev.ep_len.numpy(), shadow.episode_lengths)
    assert list(rb.pos) == list(shadow.pos) and rb.can_sample(2) == shadow.can_sample(2)


def test_producer_matches_the_reference_buffer_under_random_call_sequences(emu):
    _run(emu)
