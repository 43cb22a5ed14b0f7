"""env-steps/s of the coupled 1 : 1 loops (serial, two-stream) on live CarFlag, cfg-1 network -- the loop part of bench.py alone."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
c = bench.CONFIGS[1]
agent = bench.make_agent(c, c["B"], torch.device("cuda:0"), 0, "device")
for _ in range(500):
    agent.train()
torch.cuda.synchronize()
r = bench.env_step_rate(agent, seconds=float(sys.argv[1]) if len(sys.argv) > 1 else 3.0, vector_sizes=())
print({k: round(v, 1) for k, v in r.items()})
