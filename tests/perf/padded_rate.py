#!/usr/bin/env python3
"""What width / head padding costs: TD-updates/s of cfg-1-like workloads (CarFlag shapes, context 50, 2 layers, batch 32,
synthetic replay) at shapes the kernels are instantiated for and at shapes that run zero-padded on the row-block path.
   python tests/perf/padded_rate.py            (on the GPU box; ~40 s)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402


def rate(D, H, steps=400, warm=60, force_tiled=False):
    if force_tiled:
        os.environ["DTQN_FORCE_TILED"] = "1"
    c = dict(bench.CONFIGS[1], D=D, H=H)
    agent = bench.make_agent(c, 32, torch.device("cuda", 0), 0, "device", data_parallel=False)
    os.environ.pop("DTQN_FORCE_TILED", None)
    for _ in range(warm):
        agent.train()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        agent.train()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    agent._drain_stats(block=True)
    net = agent.policy_network.net
    return steps / dt, dict(d_model=net.d_model, heads=net.num_heads, head_dim=net.head_dim, d_real=net.d_real, tiled=net.tiled)


if __name__ == "__main__":
    # one PROCESS per shape: every learner of the row-block path brings its own side stream, and a process's streams share
    # GPU_MAX_HW_QUEUES hardware queues -- the fifth learner of one process measured 856 updates/s where its own process gives 3.8 k
    import subprocess
    if len(sys.argv) == 4:
        D, H, ft = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] == "1"
        r, info = rate(D, H, force_tiled=ft)
        print(f"in-embed {D:3d} heads {H}{' (row-block path forced)' if ft else ''}: {r:8.1f} TD-updates/s   {info}", flush=True)
    else:
        for D, H, ft in ((64, 8, False), (64, 8, True), (48, 6, False), (48, 4, False), (64, 2, False), (64, 1, False), (128, 8, False), (96, 6, False),
                         (96, 8, False), (128, 4, False)):
            out = subprocess.run([sys.executable, os.path.abspath(__file__), str(D), str(H), "1" if ft else "0"], capture_output=True, text=True)
            print("\n".join(l for l in out.stdout.splitlines() if l.startswith("in-embed")) or out.stderr[-400:], flush=True)
