"""Which tensors of the backward's gradient record differ between two builds of the engine (DTQN_HIP_LIB=...)?

    DTQN_HIP_LIB=<so> python tests/perf/guard_fold_probe.py dump gpurun_out/probe_<tag>.npz [repeats]
    python tests/perf/guard_fold_probe.py compare gpurun_out/probe_a.npz gpurun_out/probe_b.npz

dump: <D = 128, 16-row slices> backward (batch 8, row_split 4) on a fixed batch; stores grad, the grd / small records and
how many of `repeats` repetitions were bit-identical to the first (a race shows up as run-to-run differences, a
miscompile as a stable difference between the builds)."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch


def dump(path, repeats):
    from dtqn_amd import engine
    from oracle import dtqn_oracle as O
    from helpers import make_td_case
    lib = engine.get_lib()
    out = {"build": lib.dtqn_build_info().decode()}
    for name, kw, Bn in (("d128_disc", dict(obs_dim=10, num_actions=10, inner_embed_size=128, num_heads=8, history_len=50, discrete=True, vocab_sizes=9), 8),
                         ("d128_box", dict(obs_dim=3, num_actions=5, inner_embed_size=128, num_heads=8, history_len=50), 16)):
        cfg = O.NetCfg(**kw)
        net, oracle, host, eng, rep = make_td_case(lib, cfg, seed=21, batch=Bn, T=120, n_eps=40, mask=8 if cfg.discrete else -5, device="cuda", test_lib=False)
        assert eng.row_split == 4, eng.row_split
        eps, starts = host.sample_indices(Bn)
        eng.set_indices(eps, starts)
        eng.forward_backward(rep)
        torch.cuda.synchronize()
        g0, r0 = eng.grad.clone(), eng.grd.clone()
        same = 0
        for _ in range(repeats):
            eng.forward_backward(rep)
            torch.cuda.synchronize()
            same += int(torch.equal(eng.grad, g0) and torch.equal(eng.grd, r0))
        out[name + "/grad"] = g0.cpu().numpy()
        out[name + "/grd"] = r0.cpu().numpy()
        out[name + "/small"] = eng.small.cpu().numpy()
        out[name + "/q3"] = eng.q3.cpu().numpy()
        out[name + "/same"] = same
        out[name + "/repeats"] = repeats
        fields = {k: getattr(net, k) for k in ("grd_stride", "go_dx0", "go_layer0", "grd_layer_stride", "go_dhh", "go_dq", "gl_dqkv", "gl_da", "gl_dhp", "gl_df",
                                               "lp", "d_model", "num_layers", "ap", "sp_stride", "so_ln")}
        out[name + "/net"] = json.dumps(fields)
        out[name + "/batch"] = Bn
        print(name, "bit-identical repetitions:", same, "/", repeats, "| build:", out["build"])
    np.savez(path, **out)


def compare(a, b):
    za, zb = np.load(a), np.load(b)
    print("A:", za["build"], "\nB:", zb["build"])
    for name in ("d128_disc", "d128_box"):
        n = json.loads(str(za[name + "/net"]))
        Bn, LP, D = int(za[name + "/batch"]), n["lp"], n["d_model"]
        print(f"== {name}: run-to-run identical A {int(za[name + '/same'])}/{int(za[name + '/repeats'])}, B {int(zb[name + '/same'])}/{int(zb[name + '/repeats'])}")
        ga, gb = za[name + "/grad"], zb[name + "/grad"]
        print("  grad  max|A-B| / max|A| =", np.abs(ga - gb).max() / np.abs(ga).max())
        ra = za[name + "/grd"].reshape(Bn, n["grd_stride"]); rb = zb[name + "/grd"].reshape(Bn, n["grd_stride"])
        secs = [("dx0", n["go_dx0"], LP * D), ("dhh", n["go_dhh"], LP * D), ("dq", n["go_dq"], LP * n["ap"])]
        for l in range(n["num_layers"]):
            base = n["go_layer0"] + l * n["grd_layer_stride"]
            secs += [(f"L{l}.dqkv", base + n["gl_dqkv"], LP * 3 * D), (f"L{l}.da", base + n["gl_da"], LP * D),
                     (f"L{l}.dhp", base + n["gl_dhp"], LP * 4 * D), (f"L{l}.df", base + n["gl_df"], LP * D)]
        for nm, off, sz in secs:
            xa, xb = ra[:, off:off + sz], rb[:, off:off + sz]
            d = np.abs(xa - xb)
            if d.max() > 0:
                w = xa.shape[1] // LP
                rows = np.unique(np.argwhere(d.reshape(Bn, LP, w) > 0)[:, 1])
                cols = np.unique(np.argwhere(d.reshape(Bn, LP, w) > 0)[:, 2])
                print(f"  {nm:9s} max|A-B| = {d.max():.3e} (max|A| {np.abs(xa).max():.3e}); rows {rows.min()}..{rows.max()} ({len(rows)}), cols {cols.min()}..{cols.max()} ({len(cols)}), seqs {np.unique(np.argwhere(d > 0)[:, 0]).tolist()}")
            else:
                print(f"  {nm:9s} identical")
        sa, sb = za[name + "/small"], zb[name + "/small"]
        print("  small max|A-B| =", np.abs(sa - sb).max(), " q3 max|A-B| =", np.abs(za[name + "/q3"] - zb[name + "/q3"]).max())


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 30)
    else:
        compare(sys.argv[2], sys.argv[3])
