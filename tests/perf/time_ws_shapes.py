"""TD-update time of whole-sequence shapes at d_model 64 (config-1 network, residual and GRU gates) over batches that exercise the one-, two- and
four-slice backward instantiations.   python tests/perf/time_ws_shapes.py"""
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import dtqn_oracle as O
from helpers import make_td_case
from dtqn_amd import engine
lib = engine.get_lib(); engine.require_gpu()
for gate in ("res", "gru"):
    for Bn in (32, 64, 96, 128, 192, 256):
        cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50, gate=gate)
        net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=1, batch=Bn, T=200, n_eps=2 * Bn + 40, mask=-5, device="cuda", test_lib=False)
        eps, starts = host.sample_indices(Bn); eng.set_indices(eps, starts)
        n, r, t, s = ctypes.byref(eng.net), ctypes.byref(rep.view), ctypes.byref(eng.td), eng._stream()
        for _ in range(10): assert lib.dtqn_td_update(n, r, t, s) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): lib.dtqn_td_update(n, r, t, s)
        e1.record(); torch.cuda.synchronize()
        print(f"{gate} batch {Bn:3d} row_split {eng.row_split}: {e0.elapsed_time(e1) * 5:.1f} us / update", flush=True)
        del net, oracle, host, eng, rep
