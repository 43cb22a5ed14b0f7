import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import dtqn_oracle as O
from helpers import net_from_cfg
from dtqn_amd import engine
from dtqn_amd.learner import TdEngine, DeviceReplay
lib = engine.get_lib(); engine.require_gpu()
cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50)
net = net_from_cfg(lib, cfg)
for Bn in (32, 256):
    for ns in (2, 4, 8, 16, 32):
        if ns > Bn: continue
        eng = TdEngine(net, Bn, n_split=ns)
        n, t, s = ctypes.byref(eng.net), ctypes.byref(eng.td), eng._stream()
        for name, fn in (("wgrad", lambda: lib.dtqn_td_wgrad(n, t, s)), ("reduce", lambda: lib.dtqn_td_reduce(n, t, s))):
            for _ in range(5): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(30):
                e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            print(f"B={Bn} n_split={ns} {name}: median {ts[len(ts)//2]:.1f} us  min {ts[0]:.1f}")
