"""G12 on the MI355X: dtqn_amd's run.py loop (prepopulate -> step / flush / reset / train / anneal, evaluation every
eval_frequency steps) with `--sampler reference --ref-quirks` semantics against the trace THE REFERENCE's own run.py left
(tests/golden/make_golden.py gen_G12), at a small shape and at BASELINE config 1's shapes (ctx 50, d_model 64, batch 32).
Comparison rules: tests/loop_harness.py (exact events, absolute 1e-4 on the acting Q row, 2e-4 on the statistics; where the
trajectory is chaotic the bounds widen to a multiple of the reference's distance to its own one-thread twin)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G12_loop.npz")


@pytest.fixture(scope="module")
def fx():
    return dict(np.load(GOLDEN))


def test_small_loop_reproduces_the_reference_trace_on_the_device(fx):
    from loop_harness import run_loop, compare_loop
    tr, agent, prepop = run_loop(fx, "small", torch.device("cuda:0"))
    s = compare_loop(fx, "small", tr, prepop, min_actions=150)
    print(s)
    assert s["updates_compared"] >= 50


def test_cfg1_loop_follows_the_reference_trace_on_the_device(fx):
    from loop_harness import run_loop, compare_loop
    tr, agent, prepop = run_loop(fx, "cfg1", torch.device("cuda:0"))
    # drift_factor 8: measured 3.9 x the reference's own distance to its one-thread twin at the end of the matched prefix (chaos-bound)
    s = compare_loop(fx, "cfg1", tr, prepop, min_actions=100, drift_factor=8.0)
    print(s)
    # the first updates are the exact pin: before the trajectory's chaos has amplified anything the bounds are the absolute ones
    early = [u for u in range(10)]
    ref, got = fx["cfg1/ev/upd_stats"], np.array(tr.ev["upd_stats"])
    assert (np.abs(got[early] - ref[early]) / np.maximum(1.0, np.abs(ref[early]))).max() <= 2e-4
    # the measured multiple of the reference's own 8-thread-vs-1-thread distance (the envelope allows 8): recorded per run, so that a
    # regression towards the bound is visible.  Round 4: 3.9; round 5 (other summation orders in the backward): 1.0
    from helpers import parity_report
    parity_report("G12_cfg1_loop_device", {"drift_factor_measured_q": 8.0 * float(s["q_err_over_bound"]),
                                           "drift_factor_measured_stats": 8.0 * float(s["stat_err_over_bound"]), "drift_factor_allowed": 8.0,
                                           "first_divergence_action": s["first_divergence"], "divergence_gap": s.get("divergence_gap"),
                                           "updates_compared": s["updates_compared"], "q_abs_err_max": s["q_abs_err_max"]})
    # where the first greedy action flips is itself chaotic (a near-tie of the reference's own Q: compare_loop checks the gap against the
    # drift bound and demands >= 100 matching action events before it): rounds 4 / 5 saw it at action events ~190 / 149
    assert s["updates_compared"] >= 60
