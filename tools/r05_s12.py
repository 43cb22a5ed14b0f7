#!/usr/bin/env python3
"""round 5, session 12: BASELINE config 2 (and batch 192 / 384 of its shapes) with bench.py's other_configs protocol (10 warm-up + 500 timed
updates on a fresh agent), whole-sequence backward against two 32-row slices per sequence in the backward -- alternating, three times."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def rate(batch, env):
    for k in ("DTQN_ROW_SPLIT", "DTQN_FWD_SLICES"):
        os.environ.pop(k, None)
    os.environ.update(env)
    c = bench.CONFIGS[2]
    agent = bench.make_agent(c, batch, torch.device("cuda", 0), 0, "device")
    for _ in range(10):
        agent.train()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        agent.train()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 500
    agent._drain_stats(block=True)
    rs = agent.engine.row_split
    del agent
    torch.cuda.empty_cache()
    return 1.0 / dt, rs


for rep in range(3):
    for batch in (256, 192, 384):
        a, rsa = rate(batch, {})
        b, rsb = rate(batch, {"DTQN_ROW_SPLIT": "1", "DTQN_FWD_SLICES": "1"})
        print(f"rep {rep} batch {batch}: whole (row_split {rsa}) {a:8.1f}   backward in two slices (row_split {rsb}) {b:8.1f}   ratio {b / a:.3f}", flush=True)
