#!/usr/bin/env python3
"""cProfile of the train() loop of one shape (tests/perf/one_shape_trace.py's workload): where the HOST time of an update goes.
   python tests/perf/one_shape_hostprof.py <in-embed> <heads>"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

D, H = int(sys.argv[1]), int(sys.argv[2])
c = dict(bench.CONFIGS[1], D=D, H=H)
agent = bench.make_agent(c, 32, torch.device("cuda", 0), 0, "device", data_parallel=False)
for _ in range(100):
    agent.train()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    agent.train()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{D}/{H} pipeline={os.environ.get('DTQN_PIPELINE', '1')}: issue {1e6 * (t1 - t0) / 300:.1f} us per update, with the final sync {1e6 * (t2 - t0) / 300:.1f}")
if os.environ.get("HOSTPROF", "1") == "1":
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        agent.train()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
