"""Experiment: the five launches of one TD update captured in a HIP graph vs issued one by one."""
import ctypes, sys, os, json, time
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
    n, r, t = ctypes.byref(eng.net), ctypes.byref(rep.view), ctypes.byref(eng.td)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        s = ctypes.c_void_p(side.cuda_stream)
        for _ in range(5): assert lib.dtqn_td_update(n, r, t, s) == 0
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            s2 = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            assert lib.dtqn_td_update(n, r, t, s2) == 0
    torch.cuda.synchronize()
    def timeit(fn, iters=300):
        for _ in range(10): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e6
    cur = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    res[f"B{Bn}_eager_us"] = timeit(lambda: lib.dtqn_td_update(n, r, t, cur))
    res[f"B{Bn}_graph_us"] = timeit(lambda: g.replay())
    # host cost of issuing only
    t0 = time.perf_counter()
    for _ in range(200): lib.dtqn_td_update(n, r, t, cur)
    res[f"B{Bn}_eager_issue_us"] = (time.perf_counter() - t0) / 200 * 1e6
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): g.replay()
    res[f"B{Bn}_graph_issue_us"] = (time.perf_counter() - t0) / 200 * 1e6
    torch.cuda.synchronize()
    print({k: round(v, 1) for k, v in res.items() if k.startswith(f"B{Bn}")})
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/graph_update.json", "w"), indent=1)
