"""TD update time of the GRU-gated cfg-1 network (D = 64, context 50, batch 32) with and without latency mode."""
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
for split in ("0", "4"):
    os.environ["DTQN_ROW_SPLIT"] = split
    from oracle import dtqn_oracle as O
    from helpers import make_td_case
    from dtqn_amd import engine
    lib = engine.get_lib(); engine.require_gpu()
    cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50, gate="gru")
    net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=1, batch=32, T=200, n_eps=300, mask=-5, device="cuda", test_lib=False)
    eps, starts = host.sample_indices(32); eng.set_indices(eps, starts)
    n, r, t, s = ctypes.byref(eng.net), ctypes.byref(rep.view), ctypes.byref(eng.td), eng._stream()
    for _ in range(10): assert lib.dtqn_td_update(n, r, t, s) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): lib.dtqn_td_update(n, r, t, s)
    e1.record(); torch.cuda.synchronize()
    print(f"GRU cfg-1 shapes, row_split={eng.row_split}: {e0.elapsed_time(e1) * 5:.1f} us / update")

# GRU gates / identity layers at D = 128 (row-block tiled path): cfg-3 shapes, batch 512, and cfg-4 shapes, batch 128
os.environ.pop("DTQN_ROW_SPLIT", None)
import json
out = {}
for name, kw, Bn, T, mask in (
        ("cfg3_res", dict(obs_dim=10, num_actions=10, inner_embed_size=128, num_heads=8, num_layers=2, history_len=50, discrete=True, vocab_sizes=9), 512, 50, 8),
        ("cfg3_gru", dict(obs_dim=10, num_actions=10, inner_embed_size=128, num_heads=8, num_layers=2, history_len=50, discrete=True, vocab_sizes=9, gate="gru"), 512, 50, 8),
        ("cfg3_identity", dict(obs_dim=10, num_actions=10, inner_embed_size=128, num_heads=8, num_layers=2, history_len=50, discrete=True, vocab_sizes=9, identity=True), 512, 50, 8),
        ("cfg4_gru", dict(obs_dim=6, num_actions=6, inner_embed_size=128, num_heads=8, num_layers=2, history_len=128, discrete=True, vocab_sizes=12, gate="gru"), 128, 250, 11)):
    cfg = O.NetCfg(**kw)
    net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=1, batch=Bn, T=T, n_eps=600, mask=mask, device="cuda", test_lib=False)
    eps, starts = host.sample_indices(Bn); eng.set_indices(eps, starts)
    n, r, t, s = ctypes.byref(eng.net), ctypes.byref(rep.view), ctypes.byref(eng.td), eng._stream()
    for _ in range(3): assert lib.dtqn_td_update(n, r, t, s) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): lib.dtqn_td_update(n, r, t, s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    D, L = cfg.inner_embed_size, cfg.history_len
    ftok = 2 * cfg.obs_dim * 8 * D + 2 * (24 * D * D + 4 * L * D + (24 * D * D if cfg.gate == "gru" else 0)) + 2 * D * D + 2 * D * cfg.num_actions
    out[name] = {"ms_per_update": ms, "tiled": int(net.tiled), "alg_tflops": 5 * Bn * L * ftok / ms / 1e9}
    print(f"{name}: tiled={net.tiled} {ms:.3f} ms / update, {out[name]['alg_tflops']:.1f} algorithmic TFLOP/s")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/time_variants.json", "w"), indent=1)
