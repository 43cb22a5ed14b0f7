"""Wall-clock of one TD update and of each stage (cfg-1/2 shapes, synthetic replay)."""
import ctypes, sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import dtqn_oracle as O
from helpers import make_td_case
from dtqn_amd import engine
lib = engine.get_lib(); engine.require_gpu()
res = {}
for Bn in (32, 256):
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50)
    net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=1, batch=Bn, T=200, n_eps=300, mask=-5, device="cuda", test_lib=False)
    eps, starts = host.sample_indices(Bn); eng.set_indices(eps, starts)
    n, r, t, s = ctypes.byref(eng.net), ctypes.byref(rep.view), ctypes.byref(eng.td), eng._stream()
    stages = {"forward": lambda: lib.dtqn_td_forward(n, r, t, s), "backward": lambda: lib.dtqn_td_backward(n, r, t, s),
              "wgrad": lambda: lib.dtqn_td_wgrad(n, t, s), "reduce": lambda: lib.dtqn_td_reduce(n, t, s),
              "clip_adam": lambda: lib.dtqn_td_clip_adam(n, t, s), "update": lambda: lib.dtqn_td_update(n, r, t, s)}
    for name, fn in stages.items():
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 10
        res[f"B{Bn}_{name}_us"] = us
        print(f"B={Bn} {name}: {us:.1f} us")
    print(f"B={Bn}: n_split={eng.n_split} wtiles={net.n_wtiles} workspace={eng.workspace_bytes()/1e6:.1f} MB")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/time_update.json", "w"), indent=1)
