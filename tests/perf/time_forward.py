"""Quick wall-clock of the forward kernel at cfg-1/2 shapes (3*B workgroups, like the TD forward)."""
import ctypes, sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from dtqn_amd import engine, _binding as B
from oracle import dtqn_oracle as O
from helpers import net_from_cfg, pack_theta, ptr
lib = engine.get_lib(); engine.require_gpu()
out = {}
for D, H, O_, A, disc, V in ((64, 8, 3, 3, False, 0), (128, 8, 10, 10, True, 9)):
    cfg = O.NetCfg(obs_dim=O_, num_actions=A, inner_embed_size=D, num_heads=H, history_len=50, discrete=disc, vocab_sizes=V)
    net = net_from_cfg(lib, cfg)
    theta = torch.from_numpy(pack_theta(net, O.init_params(cfg, 1))).cuda()
    for Bn in (32, 96, 256, 768, 1536):
        obs = (torch.randint(0, max(V, 1), (Bn, 50, O_)).float() if disc else torch.rand(Bn, 50, O_) * 2 - 1).cuda()
        act = torch.zeros(Bn, 50, dtype=torch.uint8).cuda()
        q = torch.empty(Bn, 50, A).cuda()
        s = engine.stream_ptr()
        for _ in range(5):
            lib.dtqn_forward(ctypes.byref(net), ptr(theta), ptr(obs), ptr(act), Bn, 50, ptr(q), s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            lib.dtqn_forward(ctypes.byref(net), ptr(theta), ptr(obs), ptr(act), Bn, 50, ptr(q), s)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 50
        flop = Bn * 64 * (2 * D * 3 * D + 2 * D * D + 16 * D * D + 2 * D * D) * 2
        out[f"D{D}_B{Bn}"] = {"us": us, "mfma_tflops_padded": flop / us / 1e6}
        print(f"D={D} B={Bn}: {us:.1f} us/launch  ({flop/us/1e6:.1f} TFLOP/s on padded GEMM flops)")
# tiled path: BASELINE config 4 / 5 shapes
for D, H, L, O_, A, V, Bs in ((128, 8, 128, 6, 6, 12, (128, 384)), (256, 8, 256, 1, 5, 22, (32, 96))):
    cfg = O.NetCfg(obs_dim=O_, num_actions=A, inner_embed_size=D, num_heads=H, history_len=L, discrete=True, vocab_sizes=V)
    net = net_from_cfg(lib, cfg)
    theta = torch.from_numpy(pack_theta(net, O.init_params(cfg, 1))).cuda()
    for Bn in Bs:
        obs = torch.randint(0, V, (Bn, L, O_)).float().cuda(); act = torch.zeros(Bn, L, dtype=torch.uint8).cuda()
        q = torch.empty(Bn, L, A).cuda(); ws = torch.empty(lib.dtqn_forward_workspace_floats(ctypes.byref(net), Bn)).cuda()
        s = engine.stream_ptr()
        fn = lambda: lib.dtqn_forward_tiled(ctypes.byref(net), ptr(theta), ptr(obs), ptr(act), Bn, L, ptr(q), ptr(ws), s)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 20
        ftok = 2 * O_ * 8 * D + 2 * (6 * D * D + 2 * D * D + 16 * D * D + 4 * L * D) + 2 * D * D + 2 * D * A
        out[f"tiled_D{D}_L{L}_B{Bn}"] = {"us": us, "alg_tflops": Bn * L * ftok / us / 1e6}
        print(f"tiled D={D} L={L} B={Bn}: {us:.1f} us/forward  ({Bn * L * ftok / us / 1e6:.1f} algorithmic TFLOP/s)")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/time_forward.json", "w"), indent=1)
