"""Stage times (forward x3 / backward chain / weight gradients / reduce / clip + Adam / whole update) of the row-block path at the
BASELINE config 3 / 4 / 5 shapes and per-GPU batches.   python tests/perf/time_stages_cfg.py [3 4 5] [--out file.json]"""
import ctypes, sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import dtqn_oracle as O
from helpers import make_td_case
from dtqn_amd import engine
lib = engine.get_lib(); engine.require_gpu()
SHAPES = {
    3: (dict(obs_dim=10, num_actions=10, inner_embed_size=128, num_heads=8, num_layers=2, history_len=50, discrete=True, vocab_sizes=9), 512),
    4: (dict(obs_dim=6, num_actions=6, inner_embed_size=128, num_heads=8, num_layers=2, history_len=128, discrete=True, vocab_sizes=12), 128),
    5: (dict(obs_dim=1, num_actions=5, inner_embed_size=256, num_heads=8, num_layers=2, history_len=256, discrete=True, vocab_sizes=22), 32),
}
out = os.path.join(ROOT, "gpurun_out", "time_stages_cfg.json")
argv = sys.argv[1:]
if "--out" in argv:
    out = argv[argv.index("--out") + 1]
    del argv[argv.index("--out"):argv.index("--out") + 2]
cids = [int(a) for a in argv] or [3, 4, 5]
res = {}
for cid in cids:
    kw, Bn = SHAPES[cid]
    cfg = O.NetCfg(**kw)
    L = cfg.history_len
    net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=1, batch=Bn, T=L + 40, n_eps=2 * Bn + 8, mask=kw["vocab_sizes"] - 1,
                                               device="cuda", test_lib=False)
    eps, starts = host.sample_indices(Bn); eng.set_indices(eps, starts)
    n, r, t, s = ctypes.byref(eng.net), ctypes.byref(rep.view), ctypes.byref(eng.td), eng._stream()
    stages = {"forward": lambda: lib.dtqn_td_forward(n, r, t, s), "backward": lambda: lib.dtqn_td_backward(n, r, t, s),
              "wgrad": lambda: lib.dtqn_td_wgrad(n, t, s), "reduce": lambda: lib.dtqn_td_reduce(n, t, s),
              "clip_adam": lambda: lib.dtqn_td_clip_adam(n, t, s), "update": lambda: lib.dtqn_td_update(n, r, t, s)}
    row = {}
    for name, fn in stages.items():
        for _ in range(5): assert fn() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): fn()
        e1.record(); torch.cuda.synchronize()
        row[name + "_us"] = e0.elapsed_time(e1) * 1e3 / 30
    D, NL, A = cfg.inner_embed_size, cfg.num_layers, cfg.num_actions
    e_in = kw["obs_dim"] * 8
    ftok = 2 * e_in * D + NL * (24 * D * D + 4 * L * D) + 2 * D * D + 2 * D * A
    row["gflop"] = 5 * Bn * L * ftok / 1e9
    row["frac_mfma"] = row["gflop"] / row["update_us"] * 1e3 / 157.3
    res[f"cfg{cid}"] = row
    print(f"cfg{cid} B={Bn}: " + "  ".join(f"{k}={v:.1f}" if k.endswith("_us") else f"{k}={v:.4f}" for k, v in row.items()), flush=True)
    del net, oracle, host, eng, rep
    torch.cuda.empty_cache()
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(res, open(out, "w"), indent=1)
