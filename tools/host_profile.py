"""cProfile of the host side of agent.train() (bench loop) to see where Python time goes."""
import cProfile, pstats, sys, os, io, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from dtqn_amd import envs as dt_envs
from dtqn_amd.utils.agent_utils import get_agent
from dtqn_amd.utils.random import set_global_seed
c = bench.cfg1_shapes()
env = dt_envs.make("DiscreteCarFlag-v0"); set_global_seed(1, env)
agent = get_agent("DTQN", [env], 8, 0, c["D"], 500_000, torch.device("cuda"), 3e-4, 32, c["L"], -1, c["L"], 10_000, 0.99,
                  c["H"], c["NL"], 0.0, False, "res", "learned", 0, sampler="device", sample_seed=1)
bench.fill_synthetic_replay(agent, 1, c)
for _ in range(200): agent.train()
torch.cuda.synchronize()
# pure host cost: time N calls without waiting for the GPU (queue depth permitting)
t0 = time.perf_counter()
for _ in range(300): agent.train()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host issue time per train(): {(t1-t0)/300*1e6:.1f} us ; incl. drain {(t2-t0)/300*1e6:.1f} us")
pr = cProfile.Profile(); pr.enable()
for _ in range(500): agent.train()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:3500])
