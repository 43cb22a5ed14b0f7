"""Wall-clock of one TD update and of each stage on the row-block tiled path (BASELINE cfg-4 / cfg-5 shapes)."""
import ctypes, sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import dtqn_oracle as O
from helpers import make_td_case
from dtqn_amd import engine
lib = engine.get_lib(); engine.require_gpu()
res = {}
SHAPES = {
    "cfg4": dict(obs_dim=6, num_actions=6, inner_embed_size=128, num_heads=8, num_layers=2, history_len=128, discrete=True, vocab_sizes=12),
    "cfg5": dict(obs_dim=1, num_actions=5, inner_embed_size=256, num_heads=8, num_layers=2, history_len=256, discrete=True, vocab_sizes=22),
}
for tag, kw in SHAPES.items():
    for Bn in (32,):
        cfg = O.NetCfg(**kw)
        L = cfg.history_len
        net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=1, batch=Bn, T=L + 40, n_eps=64, mask=kw["vocab_sizes"] - 1,
                                                   device="cuda", test_lib=False)
        eps, starts = host.sample_indices(Bn); eng.set_indices(eps, starts)
        n, r, t, s = ctypes.byref(eng.net), ctypes.byref(rep.view), ctypes.byref(eng.td), eng._stream()
        stages = {"forward": lambda: lib.dtqn_td_forward(n, r, t, s), "backward": lambda: lib.dtqn_td_backward(n, r, t, s),
                  "wgrad": lambda: lib.dtqn_td_wgrad(n, t, s), "reduce": lambda: lib.dtqn_td_reduce(n, t, s),
                  "clip_adam": lambda: lib.dtqn_td_clip_adam(n, t, s), "update": lambda: lib.dtqn_td_update(n, r, t, s)}
        for name, fn in stages.items():
            for _ in range(3): assert fn() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 50
            res[f"{tag}_B{Bn}_{name}_us"] = us
            print(f"{tag} B={Bn} {name}: {us:.1f} us")
        D, NL = cfg.inner_embed_size, cfg.num_layers
        flops = 3 * Bn * L * (2 * 12 * D * D * NL + 2 * D * D) + 2 * 2 * Bn * L * (12 * D * D * NL + D * D)   # GEMM flops only
        res[f"{tag}_B{Bn}_gemm_tflops"] = flops / (res[f"{tag}_B{Bn}_update_us"] * 1e-6) / 1e12
        print(f"{tag} B={Bn}: GEMM-only {res[f'{tag}_B{Bn}_gemm_tflops']:.1f} TFLOP/s  n_split={eng.n_split} workspace={eng.workspace_bytes()/1e6:.1f} MB")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/time_update_tiled.json", "w"), indent=1)
