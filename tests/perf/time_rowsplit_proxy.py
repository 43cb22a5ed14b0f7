"""Proxy for a 2-way row split of cfg 1: the same kernels on half-length windows (L=25 -> LP=32) at twice the batch,
8 waves per workgroup (DTQN_WAVES=8), against the real cfg-1 shape."""
import ctypes, sys, os
os.environ["DTQN_WAVES"] = "8"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import dtqn_oracle as O
from helpers import make_td_case
from dtqn_amd import engine
lib = engine.get_lib(); engine.require_gpu()
os.environ['DTQN_ROW_SPLIT'] = '0'
for L, Bn in ((50, 32), (32, 64), (16, 128)):
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=L)
    net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=1, batch=Bn, T=200, n_eps=300, mask=-5, device="cuda", test_lib=False)
    eps, starts = host.sample_indices(Bn); eng.set_indices(eps, starts)
    n, r, t, s = ctypes.byref(eng.net), ctypes.byref(rep.view), ctypes.byref(eng.td), eng._stream()
    stages = {"forward": lambda: lib.dtqn_td_forward(n, r, t, s), "backward": lambda: lib.dtqn_td_backward(n, r, t, s),
              "wgrad": lambda: lib.dtqn_td_wgrad(n, t, s), "update": lambda: lib.dtqn_td_update(n, r, t, s)}
    out = []
    for name, fn in stages.items():
        for _ in range(5): assert fn() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(f"{name} {e0.elapsed_time(e1) * 10:.1f}")
    print(f"L={L} (lp={net.lp}) B={Bn}: " + "  ".join(out))
