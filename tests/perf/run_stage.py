"""Launch one stage of the TD update N times (for rocprofv3 --pmc passes).  usage: run_stage.py <stage> [iters] [batch]"""
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import dtqn_oracle as O
from helpers import make_td_case
from dtqn_amd import engine
lib = engine.get_lib(); engine.require_gpu()
stage = sys.argv[1]; iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20; Bn = int(sys.argv[3]) if len(sys.argv) > 3 else 32
cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50)
net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=1, batch=Bn, T=200, n_eps=300, mask=-5, device="cuda", test_lib=False)
eps, starts = host.sample_indices(Bn); eng.set_indices(eps, starts)
n, r, t, s = ctypes.byref(eng.net), ctypes.byref(rep.view), ctypes.byref(eng.td), eng._stream()
eng.forward_backward(rep); torch.cuda.synchronize()
fn = {"forward": lambda: lib.dtqn_td_forward(n, r, t, s), "backward": lambda: lib.dtqn_td_backward(n, r, t, s),
      "wgrad": lambda: lib.dtqn_td_wgrad(n, t, s), "reduce": lambda: lib.dtqn_td_reduce(n, t, s),
      "update": lambda: lib.dtqn_td_update(n, r, t, s)}[stage]
for _ in range(iters):
    fn(); torch.cuda.synchronize()
