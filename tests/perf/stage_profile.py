"""Per-stage wall-clock breakdown of workgroups 0 / 1 of the TD forward / backward kernels (debug aid).
Needs a library built with the stage clocks: DTQN_BUILD_PROF=1 python -m dtqn_amd.build (the product build has none)."""
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
eps, starts = host.sample_indices(Bn); eng.set_indices(eps, starts)
prof = torch.zeros(128, dtype=torch.int64, device="cuda")
lib.dtqn_debug_set_profile_buffer(ctypes.c_void_p(prof.data_ptr()))
for _ in range(5): eng.forward_backward(rep)
torch.cuda.synchronize()
acc = np.zeros(128)
N = 20
for _ in range(N):
    prof.zero_(); eng.forward_backward(rep); torch.cuda.synchronize()
    p = prof.cpu().numpy().astype(np.float64)
    for base in (0, 32, 64, 96):
        seg = p[base:base + 24]; n = int(np.argmax(seg <= 0)) if (seg <= 0).any() else 24      # the contiguous stage marks (24..31: finer marks)
        if n < 2:
            continue
        acc[base + 1:base + n] += np.diff(seg[:n]) / 100.0     # 100 MHz -> us
        acc[base] += (seg[n - 1] - seg[0]) / 100.0
fw = ["total", "embed"] + [f"L{l}:{s}" for l in range(2) for s in ("qkv", "attn", "outproj", "LN1", "FFN")] + ["LN2(last)", "head+Q"]
bw = ["total", "loss", "head"] + [f"L{l}:{s}" for l in (1, 0) for s in ("LN2b", "FFNb", "LN1b", "dO", "attnb", "dqkvWin")] + ["tail"]
print(f"row_split = {eng.row_split}")
for name, base, labels in (("forward", 0, fw), ("backward", 64, bw)):
    print(f"== {name} (B={Bn}, workgroups 0 | 1, mean of {N})")
    for i, lab in enumerate(labels):
        print(f"  {lab:12s} {acc[base + i] / N:8.2f} us   {acc[base + 32 + i] / N:8.2f} us")
# finer marks of the backward's loss stage and of the first LayerNorm backward (slots 24..31 of workgroups 0 | 1), us from the kernel's first mark
p = prof.cpu().numpy().astype(np.float64)
lab = ["prefetches issued", "dq zeroed + barrier", "loss wave done", "barrier", "L_top: before s2 -> LDS", "s2 in LDS", "barrier", "LN2 backward body"]
print("== backward, finer marks of the LAST run (us since the kernel's first mark; workgroups 0 | 1)")
for i, name in enumerate(lab):
    a, b = p[64 + 24 + i], p[96 + 24 + i]
    print(f"  {name:26s} {(a - p[64]) / 100.0 if a > 0 else float('nan'):8.2f}   {(b - p[96]) / 100.0 if b > 0 else float('nan'):8.2f}")
print("  (loss done / head done at %.2f / %.2f | %.2f / %.2f)" % ((p[65] - p[64]) / 100, (p[66] - p[64]) / 100, (p[97] - p[96]) / 100, (p[98] - p[96]) / 100))
lib.dtqn_debug_set_profile_buffer(None)
