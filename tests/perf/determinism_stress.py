"""Stress: the TD gradient of a fixed batch must be bit-identical over many repetitions (no atomics, fixed reduction
orders) in every latency-mode configuration -- a cheap detector for races in the row-slice hand-overs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
res = []
for split, D, disc in (("4", 64, False), ("4", 128, False), ("4", 128, True), ("1", 128, True), ("1", 64, False), ("0", 64, False)):
    os.environ["DTQN_ROW_SPLIT"] = split
    from dtqn_amd import engine
    from oracle import dtqn_oracle as O
    from helpers import make_td_case
    lib = engine.get_lib(); engine.require_gpu()
    kw = dict(obs_dim=10 if disc else 3, num_actions=5, inner_embed_size=D, num_heads=8, history_len=50, discrete=disc, vocab_sizes=9 if disc else None)
    cfg = O.NetCfg(**kw)
    Bn = 32 if D == 64 else 16
    net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=3, batch=Bn, T=120, n_eps=40, mask=8 if disc else -5, device="cuda", test_lib=False)
    eps, starts = host.sample_indices(Bn); eng.set_indices(eps, starts)
    eng.forward_backward(rep); torch.cuda.synchronize()
    ref = eng.grad.clone(); refq = eng.q3.clone()
    bad = 0
    for it in range(300):
        eng.forward_backward(rep)
        if it % 10 == 9:
            torch.cuda.synchronize()
            if not (torch.equal(eng.grad, ref) and torch.equal(eng.q3, refq)): bad += 1
    finite = bool(torch.isfinite(ref).all())
    print(f"row_split={eng.row_split} D={D} discrete={disc} B={Bn}: mismatching checks {bad}/30, finite {finite}, flags {int(eng.xflags.sum())}")
    res.append(bad == 0 and finite)
print("ALL DETERMINISTIC" if all(res) else "NONDETERMINISM FOUND")
