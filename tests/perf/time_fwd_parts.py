"""HIP-event timing of the TD forward launched whole and in parts (dtqn_td_forward_part): the policy passes as 2 B 4 = 256
workgroups of 16 rows, the target pass alone, against the three-pass launch of two 32-row slices.  cfg-1 shapes."""
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import dtqn_oracle as O
from helpers import make_td_case
from dtqn_amd import engine
lib = engine.get_lib(); engine.require_gpu()
Bn = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=64, num_heads=8, num_layers=2, history_len=50)
net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=1, batch=Bn, T=200, n_eps=300, mask=-5, device="cuda", test_lib=False)
eng.sample_in_forward(300, -1, 5)
stream = torch.cuda.current_stream(); s = ctypes.c_void_p(stream.cuda_stream)
n, r, t = eng._net_ref, rep.view_ref, eng._td_ref
cases = {"whole 3 passes x 2 slices": lambda: lib.dtqn_td_forward(n, r, t, s),
         "passes 0-1 x 4 slices (256 wg)": lambda: lib.dtqn_td_forward_part(n, r, t, 0, 2, 4, -1, s),
         "passes 0-1 x 2 slices (128 wg)": lambda: lib.dtqn_td_forward_part(n, r, t, 0, 2, 2, -1, s),
         "pass 2 x 2 slices (64 wg)": lambda: lib.dtqn_td_forward_part(n, r, t, 2, 1, 2, -1, s),
         "pass 2 x 4 slices (128 wg)": lambda: lib.dtqn_td_forward_part(n, r, t, 2, 1, 4, -1, s),
         "3 passes x 4 slices (384 wg)": lambda: lib.dtqn_td_forward_part(n, r, t, 0, 3, 4, -1, s),
         "backward": lambda: lib.dtqn_td_backward(n, r, t, s)}
ref_q = None
for name, fn in cases.items():
    rc = fn()
    assert rc == 0, (name, rc)
    for _ in range(5): fn()
    stream.synchronize()
    ts = []
    for _ in range(60):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream); fn(); e1.record(stream); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"{name:34s} {np.median(ts):7.2f} us (p10 {np.percentile(ts, 10):6.2f}, p90 {np.percentile(ts, 90):6.2f})")
# the pieces leave the same Q as the whole launch
lib.dtqn_td_forward(n, r, t, s); torch.cuda.synchronize(); q_whole = eng.q3.clone()
eng.q3.zero_(); lib.dtqn_td_forward_part(n, r, t, 0, 2, 4, -1, s); lib.dtqn_td_forward_part(n, r, t, 2, 1, 2, -1, s); torch.cuda.synchronize()
print("parts == whole:", bool(torch.equal(q_whole, eng.q3)), float((q_whole - eng.q3).abs().max()))
