#!/usr/bin/env python3
"""What width / head padding costs: TD-updates/s of cfg-1-like workloads (CarFlag shapes, context 50, 2 layers, batch 32,
synthetic replay) at shapes the kernels are instantiated for, at head width 32 / zero-padded widths of d_model 64 (four-slice kernels since
round 5; DTQN_WS_LITE_OFF=1: the row-block path they ran on before) and at shapes that run zero-padded on the row-block path.
   python tests/perf/padded_rate.py            (on the GPU box; ~40 s)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402


def rate(D, H, steps=400, warm=60, force_tiled=False, lite_off=False):
    if force_tiled:
        os.environ["DTQN_FORCE_TILED"] = "1"
    if lite_off:       # A/B knob of dtqn_net_init: head width 32 / width-padded shapes on the row-block kernels as in rounds 1-4
        os.environ["DTQN_WS_LITE_OFF"] = "1"
    c = dict(bench.CONFIGS[1], D=D, H=H)
    agent = bench.make_agent(c, 32, torch.device("cuda", 0), 0, "device", data_parallel=False)
    os.environ.pop("DTQN_FORCE_TILED", None)
    os.environ.pop("DTQN_WS_LITE_OFF", None)
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
        D, H, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
        r, info = rate(D, H, force_tiled=mode == 1, lite_off=mode == 2)
        tag = {0: "", 1: " (row-block path forced)", 2: " (DTQN_WS_LITE_OFF=1: row-block path as in round 4)"}[mode]
        print(f"in-embed {D:3d} heads {H}{tag}: {r:8.1f} TD-updates/s   {info}", flush=True)
    else:
        # (what round 5 did not change -- 64/1, 128/8, 96/6, 128/4, 64/8 forced onto the row-block path -- is in profiles/r04_padded_shapes_rate.txt)
        for D, H, mode in ((64, 8, 0), (64, 4, 0), (64, 2, 0), (64, 2, 2), (48, 6, 0), (48, 6, 2), (48, 4, 0), (40, 2, 0), (32, 4, 0)):
            out = subprocess.run([sys.executable, os.path.abspath(__file__), str(D), str(H), str(mode)], capture_output=True, text=True)
            print("\n".join(l for l in out.stdout.splitlines() if l.startswith("in-embed")) or out.stderr[-400:], flush=True)
