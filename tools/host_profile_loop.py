"""cProfile of the coupled actor/learner loop (run.py step_overlapped + train) on the live CarFlag env."""
import cProfile, pstats, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import run as runpy
from dtqn_amd import envs
from dtqn_amd.utils.agent_utils import get_agent
from dtqn_amd.utils.epsilon_anneal import Constant
from dtqn_amd.utils.random import set_global_seed
env = envs.make("DiscreteCarFlag-v0")
set_global_seed(1, env)
agent = get_agent("DTQN", [env], 8, 0, 64, 500_000, torch.device("cuda"), 3e-4, 32, 50, -1, 50, 10_000, 0.99, 8, 2, 0.0, False, "res", "learned", 0,
                  sampler="device", sample_seed=1)
runpy.prepopulate(agent, 20_000, [env])
eps = Constant(0.1)
agent.context_reset(env.reset())
def loop(n):
    for _ in range(n):
        if runpy.step_overlapped(agent, env, eps):
            agent.replay_buffer.flush(); agent.context_reset(env.reset())
loop(500)
torch.cuda.synchronize()
t0 = time.perf_counter(); loop(3000); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"overlapped loop: {3000 / dt:.0f} steps/s ({dt / 3000 * 1e6:.1f} us/step)")
pr = cProfile.Profile(); pr.enable(); loop(2000); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
