"""TD-updates/s of agent.train() (the bench's other_configs protocol: wall clock over `steps` updates, pipelined where the engine
pipelines) at BASELINE configs 2-5.   python tests/perf/time_agent_cfg.py [3 4 5] [--steps 300]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
argv = sys.argv[1:]
steps = 300
if "--steps" in argv:
    steps = int(argv[argv.index("--steps") + 1]); del argv[argv.index("--steps"):argv.index("--steps") + 2]
cids = [int(a) for a in argv] or [3, 4, 5]
device = torch.device("cuda:0")
for cid in cids:
    c = bench.CONFIGS[cid]
    agent = bench.make_agent(c, c["B"], device, 0, "device")
    for _ in range(10): agent.train()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): agent.train()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    agent._drain_stats(block=True)
    gflop = 5 * c["B"] * c["L"] * bench.f_tok(c) / 1e9
    pipe = getattr(agent.engine, "_pipe", None)
    print(f"cfg{cid} B={c['B']}: {1/dt:.1f} updates/s  {dt*1e6:.1f} us  frac_mfma={gflop/dt/1e3/bench.MFMA_F32_PEAK_TFLOPS:.4f}  "
          f"pipe={'none' if pipe is None else ('ride' if pipe['ride'] else 'side-stream')} used={0 if pipe is None else pipe['used']}", flush=True)
    del agent
    torch.cuda.empty_cache()
