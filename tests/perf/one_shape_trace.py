#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats -- python tests/perf/one_shape_trace.py <in-embed> <heads> [batch] [context]: a few hundred TD updates of one
shape on the cfg-1 workload (tests/perf/padded_rate.py's loop), for a per-kernel table of shapes off the BASELINE list."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

D, H = int(sys.argv[1]), int(sys.argv[2])
Bn = int(sys.argv[3]) if len(sys.argv) > 3 else 32
c = dict(bench.CONFIGS[1], D=D, H=H)
if len(sys.argv) > 4:
    c["L"] = int(sys.argv[4])
agent = bench.make_agent(c, Bn, torch.device("cuda", 0), 0, "device", data_parallel=False)
for _ in range(200):
    agent.train()
torch.cuda.synchronize()
agent._drain_stats(block=True)
