#!/usr/bin/env python3
"""Benchmark of the DTQN TD-update hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: under torch.distributed.run (RANK / WORLD_SIZE in the environment) every process is one rank; started plainly, bench.py
re-executes itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` on 127.0.0.1 (one rank per GPU over RCCL).

One "step" = one DtqnAgent.train() = one TD update (sample windows -> 3 forwards -> double-DQN loss
-> backward -> clip -> Adam; four launches at batch 32, the window draw is part of the forward kernel) on a device-resident
synthetic replay of the shape SURVEY.md section 8d prescribes.  Workload at every N: BASELINE.json's metric configuration, DiscreteCarFlag-v0 shapes,
context 50, d_model 64, 8 heads, 2 layers, batch 32 PER GPU (weak scaling: each rank owns its
replay shard and batch; the only exchange is the flat-gradient all-reduce over RCCL).
Rank 0 prints ONE JSON line (kept under 4 KB: the contract keys first, then `roofline` and `cpu_baseline`, then one-number
summaries) and writes everything else it measured -- per-update latency distributions, per-config details, counter dumps --
to gpurun_out/bench_detail.json.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

# This one process builds a learner per BASELINE config, each with its own streams; HIP multiplexes a process's streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams that share a queue do not overlap.  A training process has three
# streams (update, actor, the row-block path's side stream); give the benchmark process room for all of its learners'.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from dtqn_amd import dist as ddp                                  # noqa: E402
from dtqn_amd.utils.agent_utils import get_agent                  # noqa: E402
from dtqn_amd import envs as dt_envs                              # noqa: E402

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3   # same guide: dense f32-input MFMA peak


# BASELINE.json `configs` as shapes.  B = per-GPU batch.  Config 1 is the one the
# metric is quoted on; 2-5 are reported as extra lines ("other_configs") and drive the parity tests.
CONFIGS = {
    1: dict(name="DiscreteCarFlag-v0", kind="box", O=3, A=3, T=200, L=50, D=64, H=8, NL=2, B=32),
    2: dict(name="DiscreteCarFlag-v0", kind="box", O=3, A=3, T=200, L=50, D=64, H=8, NL=2, B=256),
    3: dict(name="Memory-5-v0 shapes", kind="multidiscrete", nvec=7, O=10, A=10, T=50, L=50, D=128, H=8, NL=2, B=512),
    4: dict(name="gv_memory.7x7 shapes", kind="multidiscrete", nvec=10, O=6, A=6, T=250, L=128, D=128, H=8, NL=2, B=128),
    5: dict(name="POMDP-hallway shapes", kind="discrete", nvec=21, O=1, A=5, T=256, L=256, D=256, H=8, NL=2, B=32),
}


def cfg1_shapes():
    return CONFIGS[1]


class SyntheticEnv:
    """Spaces + episode limit of a BASELINE config whose simulator is not available (gridverse, gym-pomdps): only
    what get_agent() introspects.  Never stepped."""

    def __init__(self, c):
        from dtqn_amd.envs import spaces
        if c["kind"] == "box":
            self.observation_space = spaces.Box(low=-1.1, high=1.1, shape=(c["O"],), dtype=np.float32)
        elif c["kind"] == "multidiscrete":
            self.observation_space = spaces.MultiDiscrete([c["nvec"]] * c["O"])
        elif c["kind"] == "image":
            self.observation_space = spaces.Box(low=0, high=255, shape=tuple(c["image"]), dtype=np.uint8)
            self._shape = tuple(c["image"])
        else:
            self.observation_space = spaces.Discrete(c["nvec"])
        self.action_space = spaces.Discrete(c["A"])
        self._max_episode_steps = c["T"]

    def reset(self):          # get_agent probes a sample observation like the reference does (utils/env_processing.py:63-66, 86-87)
        shape = getattr(self, "_shape", None)
        return np.zeros(shape, dtype=np.uint8) if shape is not None else np.zeros(getattr(self.observation_space, "shape", None) or (1,))


def f_tok(c):
    """Algorithmic forward FLOPs per token: embed + NL*(in-proj, out-proj, FFN,
    dense attention) + Q head."""
    D, L = c["D"], c["L"]
    e_in = c["O"] if c["kind"] == "box" else c["O"] * 8
    return 2 * e_in * D + c["NL"] * (6 * D * D + 2 * D * D + 16 * D * D + 4 * L * D) + 2 * D * D + 2 * D * c["A"]


def fill_synthetic_replay(agent, seed: int, c) -> None:
    """SURVEY.md section 8d synthetic replay: every slot a finished episode, lengths U{5..T}, obs U(-1,1)^3,
    actions U{0..A-1}, rewards from {0,0,0,+1,-1}, done on the last step, reference padding elsewhere."""
    rb = agent.replay_buffer
    E, T, O = rb.max_size, c["T"], c["O"]
    rng = np.random.Generator(np.random.PCG64(seed))
    lens = rng.integers(5, T + 1, size=E)
    if c["kind"] == "box":
        obs = rng.uniform(-1, 1, size=(E, T + 1, O)).astype(np.float32)
    else:
        obs = rng.integers(0, c["nvec"], size=(E, T + 1, O)).astype(np.float32)      # tokens, stored as f32 on the device
    act = rng.integers(0, c["A"], size=(E, T + 1)).astype(np.uint8)
    rew = rng.choice(np.array([0, 0, 0, 1, -1], dtype=np.float32), size=(E, T))
    done = np.ones((E, T), dtype=np.uint8)
    t_idx = np.arange(T)[None, :]
    live = t_idx < lens[:, None]
    done[live] = 0
    done[np.arange(E), lens - 1] = 1
    rew[~live] = 0.0
    act[:, :T][~live] = 0
    act[np.arange(E), T] = 0
    pad_obs = np.arange(T + 1)[None, :] > lens[:, None]
    obs[pad_obs] = rb.obs_mask
    rb.import_arrays(dict(obss=obs, actions=act, rewards=rew, dones=done, eplens=lens.astype(np.int32)))
    rb.pos = [E + 1, 0]            # all slots finished (slot (E+1) % E is "in progress" and excluded, like the reference)


def _event_pair_overhead(stream, iters: int = 200) -> float:
    """Microseconds an EMPTY HIP event pair reads on this stream (record, record, synchronise): what every one-launch-at-a-time
    timing below carries on top of the kernel.  Subtracted, so that the per-kernel figures add up to a step."""
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        e1.record(stream)
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))


def time_kernels(agent, iters: int = 50) -> dict:
    """Average duration of each launch of one update, HIP events on the launch stream, one launch at a time, minus the reading of an
    empty event pair.  Latency mode with the pipelined update (learner.py): the launches of a step are the two policy passes (four
    16-row slices), the backward launch that also carries the NEXT update's target pass, weight gradients, clip + Adam; the target
    pass by itself (the inline fallback after a replay write or a target sync) is listed apart (`..._target_inline`)."""
    eng, rep = agent.engine, agent.replay_buffer.dev
    lib = eng.lib
    n, r, t = ctypes.byref(eng.net), ctypes.byref(rep.view), ctypes.byref(eng.td)
    stream = torch.cuda.current_stream()
    s = ctypes.c_void_p(stream.cuda_stream)
    pipelined = getattr(eng, "_pipe", None) is not None and bool(eng.td.sample_in_kernel)
    stages = {"dtqn_forward_kernel": lambda: lib.dtqn_td_forward(n, r, t, s),
              "dtqn_backward_kernel": lambda: lib.dtqn_td_backward(n, r, t, s),
              "dtqn_wgrad_kernel": lambda: lib.dtqn_td_wgrad(n, t, s),
              "dtqn_reduce_kernel": lambda: lib.dtqn_td_reduce(n, t, s),
              "dtqn_clip_adam_kernel": lambda: lib.dtqn_td_clip_adam(n, t, s)}
    if lib.dtqn_td_wgrad_is_direct(n, eng.batch):
        stages["dtqn_wgrad_direct_kernel"] = stages.pop("dtqn_wgrad_kernel")
        del stages["dtqn_reduce_kernel"]
        stages["dtqn_clip_adam_kernel"] = stages.pop("dtqn_clip_adam_kernel")      # keep launch order
    if pipelined:
        # what an update launches in this mode: the policy passes as four slices, and a backward launch that carries the next
        # update's target pass; the target pass by itself (the inline fallback) is listed apart, it is not part of a steady-state step
        d, _, nxt = eng._pipe_begin(rep)
        if nxt is None:
            d, _, nxt = eng._pipe_begin(rep)
        lib.dtqn_td_forward_part(n, r, t, 2, 1, 4, d, s)
        stages["dtqn_forward_kernel"] = lambda: lib.dtqn_td_forward_part(n, r, t, 0, 2, 4, d, s)
        stages["dtqn_backward_kernel"] = lambda: lib.dtqn_td_backward_ahead(n, r, t, nxt, d + 1, s)
        stages["dtqn_forward_kernel_target_inline"] = lambda: lib.dtqn_td_forward_part(n, r, t, 2, 1, 4, d, s)
    empty = _event_pair_overhead(stream)
    out = {}
    for name, fn in stages.items():
        for _ in range(3):
            fn()
        stream.synchronize()
        # one event pair per launch, drained in between: the same quantity rocprofv3 --kernel-trace reports
        # per dispatch (back-to-back launches of a 32-workgroup kernel would overlap their ramp-up/drain)
        ts = []
        for _ in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            fn()
            e1.record(stream)
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        out[name] = max(0.0, float(np.mean(ts)) - empty)          # us
    out["_event_pair_us"] = empty
    agent._calls_issued += 3 + iters            # the statistics ring counts optimizer launches: keep the host's count in step
    agent._drain_stats(block=True)
    agent.engine.pipeline_reset()               # ... and the pipelined forward's mirror of the optimizer step
    return out


def time_kernels_in_stream(agent, plain_step_us=None, updates: int = 120) -> dict:
    """The launches of the pipelined update timed INSIDE a running stream of updates: one HIP event behind EVERY launch
    of every update, nothing synchronised until the end.  The interval between two consecutive events is the launch behind the second one
    INCLUDING its kernel boundary -- the quantity whose sum over an update is the step, and what rocprofv3 --kernel-trace reports per
    dispatch when the launches run back to back -- plus the cost of one event, which is measured in the same run as
    (instrumented step - plain step) / launches per update (the plain step: the same calls without events, timed right before).
    A launch timed by itself behind a drained stream is not the kernel the pipeline runs: round 4's clip + Adam read 3.9 us that way
    against 7.0 us in the trace of the same run, and an event PAIR around one launch minus an empty pair (this round's first version) still
    read 2 us low on every kernel -- the empty pair contains the boundary the subtraction then removes.  The one-at-a-time readings stay in
    the detail file (`kernels_us_isolated`)."""
    eng, rep = agent.engine, agent.replay_buffer.dev
    lib = eng.lib
    n, t = eng._net_ref, eng._td_ref
    stream = torch.cuda.current_stream()
    s = ctypes.c_void_p(stream.cuda_stream)
    n_valid, exclude = agent.replay_buffer.valid_range()
    eng.sample_in_forward(n_valid, exclude, agent.sample_seed)
    names = ["dtqn_forward_kernel", "dtqn_backward_kernel", "dtqn_wgrad_direct_kernel", "dtqn_clip_adam_kernel"]
    # the stage list below leaves dtqn_td_reduce out: right only while the weight-gradient launch writes the flat gradient itself
    if not lib.dtqn_td_wgrad_is_direct(n, eng.batch):
        raise RuntimeError("time_kernels_in_stream: dtqn_td_wgrad is not direct for this batch -- add the dtqn_td_reduce stage")
    stages = [lambda: eng._forward_stage(rep, s), lambda: eng._backward_stage(rep, s),
              lambda: eng._check(lib.dtqn_td_wgrad(n, t, s), "dtqn_td_wgrad"), lambda: eng.clip_adam()]
    inline0 = eng._pipe["inline"]
    # the plain step of THIS stream of launches (same calls, no events inside): two events around `updates` updates
    for _ in range(10):
        for fn in stages:
            fn()
        agent._calls_issued += 1
    p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    p0.record(stream)
    for _ in range(updates):
        for fn in stages:
            fn()
        agent._calls_issued += 1
    p1.record(stream)
    stream.synchronize()
    agent._drain_stats(block=True)
    plain_step_us = p0.elapsed_time(p1) * 1e3 / updates if plain_step_us is None else plain_step_us
    evs = []
    for u in range(-10, updates):                    # the first updates only fill the queue (and take the one inline target pass)
        row = []
        for fn in stages:
            fn()
            if u >= -1:
                e = torch.cuda.Event(enable_timing=True)
                e.record(stream)
                row.append(e)
        if row:
            evs.append(row)
        agent._calls_issued += 1
    stream.synchronize()
    agent._drain_stats(block=True)
    iv = np.zeros((len(evs) - 1, 4))
    for u in range(1, len(evs)):
        prev = evs[u - 1][3]
        for j in range(4):
            iv[u - 1, j] = prev.elapsed_time(evs[u][j]) * 1e3
            prev = evs[u][j]
    step_instr = float(iv.sum(axis=1).mean())
    per_event = max(0.0, (step_instr - plain_step_us) / 4.0)
    out = {k: max(0.0, float(iv[:, j].mean()) - per_event) for j, k in enumerate(names)}
    out["_event_us"] = per_event
    out["_step_instrumented_us"] = step_instr
    out["_step_plain_us"] = plain_step_us
    out["_inline_target_passes"] = int(eng._pipe["inline"] - inline0)      # 1: the first update of the run (nothing was computed ahead yet)
    return out


def _round_profiles(kind: str, cid: int):
    """profiles/r<NN><suffix>_<kind>_cfg<N>.json, newest round and suffix first ('r02' < 'r02b' < 'r03' ...)."""
    import glob
    return sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]*_{kind}_cfg{cid}.json")), reverse=True)


def build_digest() -> str:
    """Source digest of the engine that is loaded (dtqn_build_info: 'dtqn_hip gfx950 src=<digest>')."""
    from dtqn_amd import engine
    info = engine.get_lib().dtqn_build_info().decode()
    return info.split("src=")[-1].split()[0] if "src=" in info else info


def pmc_traffic(kernel: str, batch: int, cid: int = 1):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 --pmc passes of this config (FETCH_SIZE and
    WRITE_SIZE collected in separate passes, in KB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).
    tools/profile_round3.sh is the only writer of those files and stamps them with the engine's source digest and the git
    commit (`_meta`).  Returns (bytes, source) -- source names the file, its stamp and whether the stamp equals the engine
    loaded now (false: the counters were taken on other kernels than the ones being timed) -- or (None, None)."""
    cands = _round_profiles("pmc_traffic", cid) if CONFIGS[cid]["B"] == batch else []
    cands.append(os.path.join(ROOT, "profiles", f"r01_pmc_traffic_B{batch}.json"))
    for path in cands:
        if not os.path.exists(path):
            continue
        d = json.load(open(path))
        meta = d.get("_meta", {})
        for k, v in d.items():
            if k != "_meta" and kernel in k and v.get("FETCH_SIZE") is not None and v.get("WRITE_SIZE") is not None:
                src = {"file": os.path.relpath(path, ROOT), "src": meta.get("src"), "git": meta.get("git"),
                       "matches_build": meta.get("src") == build_digest()}
                return int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024), src
    return None, None


def _oracle_learner(c, batch):
    from oracle import dtqn_oracle as O
    from oracle.replay_oracle import ReplayOracle, synth_fill
    disc = c["kind"] != "box"
    # reference mask / vocabulary (utils/env_processing.py:100-112, agent_utils.py:92-95): Discrete(n): mask n, V = n + 1;
    # MultiDiscrete(nvec): mask max(nvec) + 1, V = max(nvec) + 2 (cfg 3: mask 8, V = 9)
    mask = (c["nvec"] + 1 if c["kind"] == "multidiscrete" else c["nvec"]) if disc else -5
    vocab = mask + 1 if disc else 0
    cfg = O.NetCfg(obs_dim=c["O"], num_actions=c["A"], inner_embed_size=c["D"], num_heads=c["H"], num_layers=c["NL"],
                   history_len=c["L"], discrete=disc, vocab_sizes=vocab)
    learner = O.OracleLearner(cfg, O.init_params(cfg, seed=1))
    n_eps = max(58, batch + 8) if c["T"] <= 64 else 58
    buf = ReplayOracle((n_eps + 2) * c["T"], c["O"], mask, c["T"], c["L"])
    synth_fill(buf, np.random.Generator(np.random.PCG64(1)), n_eps, disc, vocab, c["A"], min_len=5)
    ot = torch.long if disc else torch.float32

    def one():
        o, a, r, no, na, d, _ = buf.sample(batch)
        learner.update(O.Batch(torch.as_tensor(o, dtype=ot), torch.as_tensor(a, dtype=torch.long), torch.as_tensor(r),
                               torch.as_tensor(no, dtype=ot), torch.as_tensor(na, dtype=torch.long), torch.as_tensor(d, dtype=torch.long)))
    return one


def cpu_baseline(c, batch: int, budget_s: float = 15.0) -> dict:
    """The oracle (un-fused PyTorch-CPU eager port of the reference's path, parity-locked to the
    reference by tests/test_oracle_golden.py) timed on this host's cores, same shapes."""
    one = _oracle_learner(c, batch)
    for _ in range(3):
        one()
    # tiny-op workloads do not scale to every host core: sweep a few thread counts inside the budget
    # and report the best one with the thread count that produced it
    trials = {}
    avail = os.cpu_count() or 1
    for threads in sorted({1, min(8, avail), min(32, avail)}):
        torch.set_num_threads(threads)
        one()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s / 3:
            one()
            n += 1
        trials[threads] = n / (time.perf_counter() - t0)
    best = max(trials, key=trials.get)
    torch.set_num_threads(min(avail, 8))
    return {"value": trials[best], "unit": "TD-updates/s", "cores": best, "kind": "port",
            "sample": f"~{budget_s:.0f} s of TD updates of the same workload (B={batch}, L={c['L']}, D={c['D']}) on the oracle; "
                      f"updates/s by torch thread count: {json.dumps({str(k): round(v, 2) for k, v in trials.items()})} "
                      f"on a {avail}-core host",
            "reference_in_build_container": reference_cpu_numbers(1)}


def cpu_baseline_other(cid: int, budget_s: float = 6.0) -> dict:
    """BASELINE configs 2-5: a bounded sample of oracle updates (one warm-up, then whole updates until the budget is
    spent, at least one) at the config's own batch, 8 threads (or all cores if fewer)."""
    c = CONFIGS[cid]
    avail = os.cpu_count() or 1
    threads = min(8, avail)
    torch.set_num_threads(threads)
    one = _oracle_learner(c, c["B"])
    one()
    n, t0 = 0, time.perf_counter()
    while n < 1 or time.perf_counter() - t0 < budget_s:
        one()
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": 1.0 / dt, "unit": "TD-updates/s", "cores": threads, "kind": "port",
            "sample": f"{n} oracle update(s) of B={c['B']}, L={c['L']}, D={c['D']} ({dt * 1e3:.0f} ms each) on a {avail}-core host",
            "reference_in_build_container": reference_cpu_numbers(cid)}


def reference_cpu_numbers(cid: int):
    """The REFERENCE's own DtqnAgent.train() timed in the build container (tests/golden/make_golden.py time ->
    tests/golden/ref_cpu_timing.json: a committed data file; the reference itself never runs on the GPU box)."""
    path = os.path.join(ROOT, "tests", "golden", "ref_cpu_timing.json")
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    runs = [r for r in d["runs"] if r["config"] == f"config{cid}"]
    return {"cpu": d.get("cpu", ""), "nproc": d["nproc"], "torch": d["stamp"]["torch"],
            "runs": [{k: r[k] for k in ("threads", "updates", "ms_median", "ms_p10", "ms_p90", "td_updates_per_s")} for r in runs]}


def make_agent(c, batch, device, rank, sampler, data_parallel=True):
    env = dt_envs.make("DiscreteCarFlag-v0") if c["kind"] == "box" else SyntheticEnv(c)
    from dtqn_amd.utils.random import set_global_seed
    if c["kind"] == "box":
        set_global_seed(1 + rank, env)
    agent = get_agent("DTQN", [env], 8, 0, c["D"], 500_000, device, 3e-4, batch, c["L"], c["T"], c["L"], 10_000, 0.99,
                      c["H"], c["NL"], 0.0, False, "res", "learned", 0, sampler=sampler, sample_seed=1 + rank, data_parallel=data_parallel)
    fill_synthetic_replay(agent, seed=1 + rank, c=c)
    return agent


def time_hbm_kernels(agent, c, kern: dict, iters: int = 50) -> dict:
    """The HBM-bound launches of the path, each priced as algorithmic bytes per launch / average
    launch duration (HIP events on the launch stream) / 8 TB/s:
      dtqn_clip_adam_kernel      28 * P_t bytes (read p, g, m, v; write p, m, v)
      dtqn_replay_sample_kernel  B draws: episode length read + (episode, start) written = 12 * B bytes
      dtqn_replay_apply_kernel   256 producer records of an env-step stream: 32-byte record + observation row read from pinned
                                 staging, obs row + action + reward + done + episode length written (a cleanse of a slot,
                                 (T + 1) * (4 * O + 1) + 5 * T bytes, once per episode start)
    At these sizes (0.4 - 7 MB, a few KB for the replay kernels) a launch is a few microseconds of launch floor; the fractions
    are small by construction and reported because north_star asks for them."""
    eng, rb = agent.engine, agent.replay_buffer
    lib, rep = eng.lib, rb.dev
    stream = torch.cuda.current_stream()
    s = ctypes.c_void_p(stream.cuda_stream)
    p_t = eng.net.n_trainable
    out = {"dtqn_clip_adam_kernel": {"bytes": 28 * p_t, "us": kern["dtqn_clip_adam_kernel"]}}

    def timed(fn):
        for _ in range(3):
            fn()
        stream.synchronize()
        ts = []
        for _ in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            fn()
            e1.record(stream)
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        return max(0.0, float(np.mean(ts)) - empty)

    empty = kern.get("_event_pair_isolated_us") or kern.get("_event_pair_us", 0.0)      # these two are timed one launch at a time
    n_valid, exclude = rb.valid_range()
    out["dtqn_replay_sample_kernel"] = {"bytes": 12 * eng.batch,
                                        "us": timed(lambda: eng.sample_on_device(rep, n_valid, exclude, 1, stream=s))}
    # producer: commits of 256 records (the staging capacity) written into the slot in progress; the slot is restored afterwards
    E, T, O = rb.max_size, c["T"], c["O"]
    slot = exclude
    keep = {k: getattr(rep, k)[slot].clone() for k in ("obs", "actions", "rewards", "dones")}
    keep_len, keep_pos = int(rb.episode_lengths[slot]), list(rb.pos)
    n_rec = rb.STAGE_CAPACITY
    obs_rows = np.random.Generator(np.random.PCG64(3)).uniform(-1, 1, size=(n_rec, O)).astype(np.float32)
    ts = []
    for i in range(8 + 3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        rb.pos = [keep_pos[0], 0]
        rb.store_obs(obs_rows[0])
        for j in range(1, n_rec):
            t = (j - 1) % T
            rb.store(obs_rows[j], 1, 0.5, False, t + 1)
            if t == T - 1:
                rb.pos = [keep_pos[0], 0]
        e0.record(stream)
        rb.commit(stream_ptr=s)
        e1.record(stream)
        e1.synchronize()
        if i >= 3:
            ts.append(e0.elapsed_time(e1) * 1e3)
    apply_bytes = n_rec * (32 + 4 * O) + (n_rec - 1) * (4 * O + 1 + 4 + 1 + 4) + ((T + 1) * (4 * O + 1) + 5 * T)
    us_apply = max(0.0, float(np.mean(ts)) - empty)
    out["dtqn_replay_apply_kernel"] = {"bytes": apply_bytes, "us": us_apply, "records": n_rec, "us_per_record": us_apply / n_rec}
    for k in ("obs", "actions", "rewards", "dones"):
        getattr(rep, k)[slot].copy_(keep[k])
    rb.episode_lengths[slot] = keep_len
    rep.ep_len.copy_(torch.from_numpy(rb.episode_lengths))
    rb.pos = keep_pos
    torch.cuda.synchronize()
    for v in out.values():
        v["GBs"] = v["bytes"] / (v["us"] * 1e-6) / 1e9
        v["frac"] = v["GBs"] / HBM_PEAK_GBS
    return out


def update_latency(agent, n: int = 500) -> dict:
    """Distribution of the GPU time of single updates: one HIP event pair per update on the launch stream, n updates
    issued back to back (no host synchronisation in between), median / p10 / p90 in microseconds."""
    stream = torch.cuda.current_stream()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    for _ in range(20):
        agent.train()
    evs[0].record(stream)
    for i in range(n):
        agent.train()
        evs[i + 1].record(stream)
    evs[-1].synchronize()
    agent._drain_stats(block=True)
    ts = np.array([evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(n)])
    return {"updates": n, "us_median": float(np.median(ts)), "us_p10": float(np.percentile(ts, 10)), "us_p90": float(np.percentile(ts, 90)),
            "us_mean": float(ts.mean())}


def mfma_counters(cid: int):
    """Hardware MFMA utilisation from the newest committed rocprofv3 --pmc pass of this config (per kernel:
    SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES and the MFMA op count).  Goes to the detail file, never into the line."""
    for path in _round_profiles("pmc_mfma", cid):
        d = json.load(open(path))
        if d:
            return {"file": os.path.relpath(path, ROOT), "counters": d}
    return None


def other_configs(device, steps: int = 500, with_cpu: bool = True) -> dict:
    """BASELINE configs 2-5 at their per-GPU batch on this GPU: TD-updates/s (wall clock over `steps` updates), the
    per-update latency distribution, the achieved algorithmic FP32 FLOP rate of the whole update (5 * B * L * F_tok,
    SURVEY.md section 8d) against the MFMA peak, and the CPU baseline."""
    out = {}
    for cid in (2, 3, 4, 5):
        c = CONFIGS[cid]
        agent = make_agent(c, c["B"], device, 0, "device")
        for _ in range(10):
            agent.train()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            agent.train()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        agent._drain_stats(block=True)
        gflop = 5 * c["B"] * c["L"] * f_tok(c) / 1e9
        kern = time_kernels(agent, 20) if not agent.engine.net.tiled else {}
        out[f"config{cid}"] = {"workload": f"{c['name']}: ctx={c['L']}, d_model={c['D']}, {c['H']} heads, {c['NL']} layers, batch {c['B']}",
                               "td_updates_per_s": 1.0 / dt, "ms_per_update": dt * 1e3, "samples_per_s": c["B"] / dt,
                               "update_latency_us": update_latency(agent, 300),
                               "algorithmic_gflop_per_update": gflop, "achieved_tflops": gflop / dt / 1e3,
                               "frac_of_f32_mfma_peak": gflop / dt / 1e3 / MFMA_F32_PEAK_TFLOPS,
                               "kernels_us": kern,
                               "clip_adam_hbm": ({"bytes": 28 * agent.engine.net.n_trainable, "us": kern["dtqn_clip_adam_kernel"],
                                                  "frac": 28 * agent.engine.net.n_trainable / (kern["dtqn_clip_adam_kernel"] * 1e-6) / 1e9 / HBM_PEAK_GBS}
                                                 if kern else time_clip_adam(agent)),
                               "kernel_path": "row-block tiled" if agent.engine.net.tiled else "whole-sequence (LDS-resident)"}
        del agent
        torch.cuda.empty_cache()
        if with_cpu:
            out[f"config{cid}"]["cpu_baseline"] = cpu_baseline_other(cid)
    return out


IMAGE_CFG = dict(name="MiniHack pixel-crop shapes", kind="image", image=(3, 144, 144), A=8, T=100, L=50, D=64, H=8, NL=2, B=32)


def f_img(c) -> int:
    """Algorithmic forward FLOPs per token of the convolutional observation embedding (representations.py:77-130): 2 * 9 * Cin * Cout
    per output pixel of the five 3x3 convolutions (strides 2, 1, 2, 1, 2, padding 1) + the Linear(128 h5 w5, D)."""
    C, h, w = c["image"]
    half = lambda x: (x - 1) // 2 + 1
    h1, w1 = half(h), half(w); h3, w3 = half(h1), half(w1); h5, w5 = half(h3), half(w3)
    conv = h1 * w1 * 64 * 9 * C + h1 * w1 * 64 * 9 * 64 + h3 * w3 * 64 * 9 * 64 + h3 * w3 * 128 * 9 * 64 + h5 * w5 * 128 * 9 * 128
    return 2 * conv + 2 * 128 * h5 * w5 * c["D"]


def image_config(device, steps: int = 8) -> dict:
    """The image-observation variant of the path (SURVEY.md section 8f rank 4 tail) at MiniHack's 3 x 144 x 144 pixel crop, cfg-1
    network and batch: uint8 replay on the device, the convolutional embedding of B (L + 1) policy rows and B L target rows in front
    of the row-block update, its backward behind it.  Algorithmic FLOPs: encoder forward of both token lists + 2 x the policy
    list for the backward, + the transformer's 5 B L F_tok."""
    c = IMAGE_CFG
    env = SyntheticEnv(c)
    agent = get_agent("DTQN", [env], 8, 0, c["D"], 4000, device, 3e-4, c["B"], c["L"], c["T"], c["L"], 10_000, 0.99, c["H"], c["NL"], 0.0, False,
                      "res", "learned", 0, sampler="device", sample_seed=1)
    rb = agent.replay_buffer
    E, T = rb.max_size, c["T"]
    g = torch.Generator(device="cpu").manual_seed(1)
    rb.dev.obs.copy_(torch.randint(0, 256, tuple(rb.dev.obs.shape), dtype=torch.uint8, generator=g))
    rb.dev.actions.copy_(torch.randint(0, c["A"], tuple(rb.dev.actions.shape), dtype=torch.uint8, generator=g))
    rb.dev.rewards.copy_((torch.randint(0, 3, tuple(rb.dev.rewards.shape), generator=g) - 1).float())
    rb.dev.dones.zero_(); rb.dev.dones[:, -1] = 1
    rb.episode_lengths[:] = T
    rb.dev.ep_len.fill_(T)
    rb.pos = [E + 1, 0]
    for _ in range(2):
        agent.train()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        agent.train()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    agent._drain_stats(block=True)
    tokens_pol, tokens_tgt = c["B"] * (c["L"] + 1), c["B"] * c["L"]
    ftok = 2 * 0 + c["NL"] * (24 * c["D"] ** 2 + 4 * c["L"] * c["D"]) + 2 * c["D"] ** 2 + 2 * c["D"] * c["A"]
    flops = f_img(c) * (tokens_pol + tokens_tgt + 2 * tokens_pol) + 5 * c["B"] * c["L"] * ftok
    out = {"workload": f"{c['name']}: obs {c['image']} uint8, ctx={c['L']}, d_model={c['D']}, batch {c['B']} ({tokens_pol} + {tokens_tgt} encoder tokens)",
           "td_updates_per_s": 1.0 / dt, "ms_per_update": dt * 1e3, "algorithmic_tflop_per_update": flops / 1e12,
           "achieved_tflops": flops / dt / 1e12, "frac_of_f32_mfma_peak": flops / dt / 1e12 / MFMA_F32_PEAK_TFLOPS,
           "encoder_gflop_per_token_forward": f_img(c) / 1e9}
    del agent
    torch.cuda.empty_cache()
    return out


def time_clip_adam(agent, iters: int = 20) -> dict:
    """The optimizer launch by itself (HIP events), priced against HBM: 28 * P_t bytes per launch."""
    eng = agent.engine
    stream = torch.cuda.current_stream()
    s = ctypes.c_void_p(stream.cuda_stream)
    n, t = ctypes.byref(eng.net), ctypes.byref(eng.td)
    for _ in range(3):
        eng.lib.dtqn_td_clip_adam(n, t, s)
    stream.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        eng.lib.dtqn_td_clip_adam(n, t, s)
        e1.record(stream)
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    us = float(np.mean(ts))
    agent._calls_issued += 3 + iters          # the statistics ring counts optimizer launches
    agent._drain_stats(block=True)
    eng.pipeline_reset()
    b = 28 * eng.net.n_trainable
    return {"bytes": b, "us": us, "frac": b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}


def env_step_rate(agent, seconds: float = 3.0, env_id: str = "DiscreteCarFlag-v0", vector_sizes=(8, 32)) -> dict:
    """Live actor loop on the host cores: epsilon-greedy get_action (GPU forward of the rolling
    context) + env step + observe, and the reference's coupled 1 env step : 1 update loop."""
    import run as runpy
    from dtqn_amd.utils.epsilon_anneal import Constant
    from dtqn_amd.utils.random import set_global_seed
    env = dt_envs.make(env_id)
    set_global_seed(1, env)
    eps = Constant(0.1)
    out = {}
    for mode in ("actor_only", "coupled_1to1", "coupled_1to1_overlapped"):
        agent.context_reset(env.reset())
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            if mode == "coupled_1to1_overlapped":
                done = runpy.step_overlapped(agent, env, eps)      # actor forward || TD update on two streams
            else:
                done = runpy.step(agent, env, eps)
            if done:
                agent.replay_buffer.flush()
                agent.context_reset(env.reset())
            if mode == "coupled_1to1":
                agent.train()
            n += 1
        torch.cuda.synchronize()
        out[mode] = n / (time.perf_counter() - t0)
    # vectorised rollout: N host environments, ONE batched actor launch per vector step, N updates per vector step (1 : 1)
    from dtqn_amd.agents.vector import VectorActor
    for N in vector_sizes:
        venvs = [dt_envs.make(env_id) for _ in range(N)]
        for k, e in enumerate(venvs):
            e.seed(100 + k)
        vec = VectorActor(agent, venvs)
        for mode in ("actor_only", "coupled_1to1"):
            vec.reset_all()
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < seconds:
                # coupled: the N updates are queued behind the actor forward and run while the host steps the N envs
                vec.step_all(eps.val, updates=N if mode == "coupled_1to1" else 0)
                n += N
            torch.cuda.synchronize()
            out[f"vector{N}_{mode}"] = n / (time.perf_counter() - t0)
    agent._drain_stats(block=True)
    return out


def _self_launch(args) -> None:
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks ourselves (same interpreter, one rank per
    GPU, rendezvous on 127.0.0.1) and hand the process over to the launcher; rank 0 of the job prints the line."""
    import socket
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and not ddp.same_device():
        raise SystemExit(f"--gpus {args.gpus} but only {n_dev} GPU(s) are visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execvpe(sys.executable, cmd, env)


def _r(x, nd=4):
    """Round floats for the line (the detail file keeps full precision)."""
    if isinstance(x, float):
        return float(f"{x:.{nd}g}")
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


LINE_LIMIT = 4000      # bytes: the driver keeps an 8 KB stdout tail; round 2's 37 KB line was cut and never parsed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS), help="BASELINE.json configs index + 1 (metric: 1)")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the config's; BASELINE.json metric: 32)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of BASELINE configs 2-5")
    ap.add_argument("--sampler", default="device", choices=["device", "reference"])
    ap.add_argument("--prewarm", type=int, default=3000, help="untimed updates on a scratch learner before the W warm-up steps (clock ramp, code loading)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-env-rate", action="store_true", help="skip the live env-steps/s loops (cleaner rocprofv3 traces)")
    ap.add_argument("--detail", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="where the full measurement record goes (the printed line only carries summaries)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _self_launch(args)                                # does not return
    rank, world, local = ddp.init_from_env("cuda")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    c = CONFIGS[args.config]
    if args.batch is None:
        args.batch = c["B"]
    agent = make_agent(c, args.batch, device, rank, args.sampler)
    if world > 1:
        # self-diagnosing start-up of a multi-GPU run (stderr; the JSON line stays the one line on stdout): which gradient exchange
        # the start-up check selected and why, and who can reach whom -- row r = can_device_access_peer(rank r's device, d)
        row = torch.tensor(ddp.peer_access_row(device) + [0] * max(0, 16 - torch.cuda.device_count()), dtype=torch.int32,
                           device="cpu" if ddp.same_device() else device)[:16]
        rows = [torch.zeros_like(row) for _ in range(world)]
        torch.distributed.all_gather(rows, row)
        if rank == 0:
            sel = agent.dp.selection if agent.dp is not None else {"kind": "none", "validated": False, "reason": "not distributed"}
            n_dev = torch.cuda.device_count()
            print(f"[bench] world={world} devices_visible={n_dev} exchange={sel['kind']} validated={sel['validated']} reason={sel['reason']!r}",
                  file=sys.stderr)
            for r, t in enumerate(rows):
                print(f"[bench] peer access from rank {r}: {t.tolist()[:n_dev]}", file=sys.stderr)
            sys.stderr.flush()
    # Device pre-warm, untimed and on a scratch learner of the same shape: a fresh box idles at low clocks and loads every code object
    # on first use; W warm-up steps of 0.1 ms each are over before either has settled.  The measured agent still takes its W steps.
    if args.prewarm > 0:
        scratch = make_agent(c, args.batch, device, rank, args.sampler, data_parallel=False)
        for _ in range(args.prewarm):
            scratch.train()
        torch.cuda.synchronize()
        scratch._drain_stats(block=True)
        del scratch

    done_ev = torch.cuda.Event()

    def sync_all():
        # a blocking synchronise wakes the host tens of microseconds after the GPU went idle -- a visible share of a 20-step (2 ms) timed
        # region; polling an event first makes the synchronise below return as soon as the last kernel has retired
        done_ev.record(torch.cuda.current_stream(device))
        while not done_ev.query():
            pass
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        agent.train()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        agent.train()
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax.item())
    agent._drain_stats(block=True)
    exchange = None
    if world > 1:
        # the one exchange step of an update, timed by itself (every rank takes part; rank 0 reports), HIP events on the launch
        # stream: the exchange the start-up check SELECTED (dist.DataParallel.selection) and, beside it, the library collective
        pct = lambda ts: None if not ts else {"median": float(np.median(ts)), "p10": float(np.percentile(ts, 10)), "p90": float(np.percentile(ts, 90))}
        sel = agent.dp.selection
        exchange = {"kind": sel["kind"], "validated": sel["validated"], "reason": sel["reason"], "bytes": int(agent.engine.grad.numel() * 4),
                    "selected_us": pct(agent.dp.time_exchange("selected")),
                    "rccl_us": pct(agent.dp.time_exchange("rccl")) if sel["kind"] != "rccl" else None}
        if exchange["rccl_us"] is None and sel["kind"] == "rccl":
            exchange["rccl_us"] = exchange["selected_us"]
        sync_all()
        # the one-GPU rate, measured by THIS job on this rank's own GPU (a solo learner: no exchange), so that the line can state the
        # weak-scaling efficiency against a number from the same run; every rank does it at the same time
        solo = make_agent(c, args.batch, device, rank, args.sampler, data_parallel=False)
        for _ in range(max(args.warmup, 20)):
            solo.train()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            solo.train()
        torch.cuda.synchronize()
        solo_rate = args.steps / (time.perf_counter() - t1)
        solo._drain_stats(block=True)
        # per-rank kernel sums (HIP events, one launch at a time): the slowest rank's kernels bound the data-parallel step
        ksum = torch.tensor([float(sum(v for k, v in time_kernels(solo, 20).items() if not k.startswith("_") and not k.endswith("_target_inline")))],
                            dtype=torch.float64, device="cpu" if ddp.same_device() else device)      # gloo (same-device smoke mode) moves host tensors
        allk = [torch.zeros_like(ksum) for _ in range(world)]
        torch.distributed.all_gather(allk, ksum)
        exchange["solo_updates_per_s_rank0"] = solo_rate
        exchange["kernels_us_sum_per_rank"] = [float(k.item()) for k in allk]
        del solo
        sync_all()
    # per-update latency distribution: agent.train() takes part in the gradient exchange, so EVERY rank runs it (rank 0 reports)
    lat = update_latency(agent)
    sync_all()

    if rank == 0:
        ms = elapsed * 1e3 / args.steps
        ups = world * args.steps / elapsed                     # whole-job TD updates / s (each rank does one per step)
        ft = f_tok(c)
        tokens = args.batch * c["L"]
        tiled = bool(agent.engine.net.tiled)
        kern = time_kernels(agent)
        kern_isolated = None
        pipe = getattr(agent.engine, "_pipe", None)
        pipe_counts = None if pipe is None else {"used": int(pipe["used"]), "inline": int(pipe["inline"])}      # of everything this agent ran so far
        if pipe is not None and pipe["ride"] and "dtqn_forward_kernel_target_inline" in kern and agent.dp is None:
            # pipelined update on one GPU: the line carries the IN-STREAM durations (what rocprofv3 --kernel-trace reports per dispatch)
            kern_isolated = kern
            ins = time_kernels_in_stream(agent)
            kern = dict(kern_isolated, **{k: v for k, v in ins.items() if not k.startswith("_")})
            kern["_event_us"] = ins["_event_us"]
            kern["_step_instrumented_us"] = ins["_step_instrumented_us"]
            kern["_step_plain_us"] = ins["_step_plain_us"]
            kern["_event_pair_isolated_us"] = kern_isolated.get("_event_pair_us")
        dom = max(("dtqn_forward_kernel", "dtqn_backward_kernel"), key=lambda k: kern[k])
        # algorithmic FLOPs per launch: forward kernel = 3 forwards; backward kernel = data-gradient half
        # of the backward (~ 1x forward; the weight-gradient half runs in dtqn_wgrad_kernel)
        # (pipelined forward: the launch on the update's stream carries the two policy passes)
        piped = "dtqn_forward_kernel_target_inline" in kern
        stage_flops = {"dtqn_forward_kernel": (2 if piped else 3) * tokens * ft, "dtqn_backward_kernel": (2 if piped else 1) * tokens * ft}
        flops = stage_flops[dom]
        ach = flops / (kern[dom] * 1e-6) / 1e12
        p_t = agent.engine.net.n_trainable
        p_all = agent.engine.net.n_theta
        gather_bytes = args.batch * ((c["L"] + 1) * 4 * c["O"] + (c["L"] + 1) + c["L"] * 4 + c["L"])
        alg_bytes = gather_bytes + 4 * 2 * p_all + 4 * p_t + 28 * p_t
        traffic, traffic_src = (None, None) if tiled else pmc_traffic(dom, args.batch, args.config)
        whole_frac = 5 * tokens * ft / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS
        crit = {k: v for k, v in kern.items() if not k.startswith("_") and not k.endswith("_target_inline")}
        detail = {"build": build_digest(), "kernels_us": kern, "kernels_us_sum": float(sum(crit.values())), "kernels_us_isolated": kern_isolated,
                  "kernels_us_method": "intervals between in-stream events behind every launch (nothing synchronised), minus the per-event cost = (instrumented step - plain step) / 4"
                  if kern_isolated is not None else "one launch at a time behind a drained stream, minus the empty event pair",
                  "forward": "policy passes as 2 x B x 4 workgroups of 16 rows; the next update's target pass rides in the backward launch"
                  if "dtqn_forward_kernel_target_inline" in kern else "three passes in one launch"}
        # -------- the line: contract keys first, then roofline and cpu_baseline, then one-number summaries ---------------
        line = {
            "metric": "env-steps/sec + TD-updates/sec, DiscreteCarFlag-v0 ctx=50 b=32, 1/2/4/8 GPU",
            "value": ups, "unit": "TD-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic replay (SURVEY.md 8d shapes), random-init weights", "prewarm_updates": args.prewarm,
            "config": {"workload": f"{c['name']} shapes (BASELINE config {args.config}): ctx={c['L']}, d_model={c['D']}, {c['H']} heads, "
                                   f"{c['NL']} layers, obs {c['O']} {'f32' if c['kind'] == 'box' else 'tokens'}, {c['A']} actions, "
                                   f"batch {args.batch} per GPU, history {c['L']}, device replay {500_000 // c['T']} x {c['T']} steps",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}", "sampler": args.sampler},
            "roofline": {"bound": "mfma", "kernel": dom if not tiled else dom.replace("_kernel", "") + " stage (row-block tl_* kernels)",
                         "achieved": ach, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK_TFLOPS,
                         "traffic": traffic, "traffic_src": traffic_src, "launch_us": kern[dom], "flops_per_launch": flops,
                         # both stages and the whole update priced the same way (backward = data-gradient half: 1x one
                         # forward pass; weight gradients are their own kernel)
                         "stage_frac": {k.replace("dtqn_", "").replace("_kernel", ""): f / (kern[k] * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS
                                        for k, f in stage_flops.items()},
                         "whole_update_frac": whole_frac},
        }
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(c, args.batch) if args.config == 1 else cpu_baseline_other(args.config, 12.0)
            detail["cpu_baseline"] = cb
            ref = cb.get("reference_in_build_container") or {}
            line["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                    "sample": cb["sample"],
                                    "reference_train_in_build_container_updates_per_s":
                                        {str(r["threads"]): r["td_updates_per_s"] for r in ref.get("runs", [])}}
        line["kernels_us"] = {k.replace("dtqn_", "").replace("_kernel", ""): v for k, v in kern.items() if not k.startswith("_")}
        line["kernels_us"]["sum_on_stream"] = detail["kernels_us_sum"]
        line["kernels_us"]["method"] = "in_stream" if kern_isolated is not None else "isolated"
        if pipe_counts is not None:
            line["pipeline"] = pipe_counts
        line["hbm_view"] = {"algorithmic_bytes_per_update": alg_bytes, "achieved_GBs": alg_bytes / (ms * 1e-3) / 1e9,
                            "frac": alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if not tiled:
            hb = time_hbm_kernels(agent, c, kern)
            detail["hbm_kernels"] = hb
            line["hbm_kernels"] = {k.replace("dtqn_", "").replace("_kernel", ""): {"bytes": v["bytes"], "us": v["us"], "frac_of_8TBs": v["frac"],
                                                                                    **({"us_per_record": v["us_per_record"]} if "us_per_record" in v else {})}
                                   for k, v in hb.items()}
        detail["update_latency_us"] = lat
        line["update_us_median"] = lat["us_median"]
        if world > 1:
            exchange["weak_scaling_efficiency"] = (ups / world) / exchange["solo_updates_per_s_rank0"]
            line["exchange"] = exchange
        if world == 1 and args.config == 1 and not args.no_other_configs:
            oc = other_configs(device, with_cpu=not args.no_cpu_baseline)
            detail["other_configs"] = {k: dict(v, mfma_counters=mfma_counters(int(k[-1]))) for k, v in oc.items()}
            img = image_config(device)
            detail["image_config"] = img
            line["image_obs_config"] = {"updates_per_s": img["td_updates_per_s"], "frac_mfma": img["frac_of_f32_mfma_peak"],
                                        "tflop_per_update": img["algorithmic_tflop_per_update"]}
            line["other_configs"] = {k[-1]: {"B": CONFIGS[int(k[-1])]["B"], "updates_per_s": v["td_updates_per_s"],
                                             "frac_mfma": v["frac_of_f32_mfma_peak"], "clip_adam_frac_hbm": v["clip_adam_hbm"]["frac"],
                                             "cpu_updates_per_s": v.get("cpu_baseline", {}).get("value")} for k, v in oc.items()}
        if world == 1 and c["kind"] == "box" and not args.no_env_rate:
            rates = env_step_rate(agent)
            detail["env_steps_per_sec"] = rates
            line["env_steps_per_sec"] = {"actor_only": rates["actor_only"], "coupled_1to1": rates["coupled_1to1"],
                                         "coupled_1to1_overlap": rates["coupled_1to1_overlapped"],
                                         **{k: v for k, v in rates.items() if k.startswith("vector")}}
        if world == 1 and args.config == 1 and not args.no_env_rate:
            # live Memory-5-v0 (BASELINE config 3's env) on the host cores against the cfg-3 network on the GPU
            c3 = CONFIGS[3]
            env = dt_envs.make("Memory-5-v0")
            from dtqn_amd.utils.random import set_global_seed
            set_global_seed(1, env)
            a3 = get_agent("DTQN", [env], 8, 0, c3["D"], 500_000, device, 3e-4, c3["B"], c3["L"], -1, c3["L"], 10_000, 0.99,
                           c3["H"], c3["NL"], 0.0, False, "res", "learned", 0, sampler="device", sample_seed=1)
            import run as runpy
            runpy.prepopulate(a3, 30_000, [env])
            r3 = env_step_rate(a3, 2.0, "Memory-5-v0", vector_sizes=(8,))
            detail["env_steps_per_sec_config3"] = dict(r3, env="Memory-5-v0 (live, host cores)", batch=c3["B"])
            line["env_steps_per_sec_cfg3"] = {"actor_only": r3["actor_only"], "coupled_1to1": r3["coupled_1to1"]}
            del a3
        detail["mfma_counters"] = mfma_counters(args.config)
        line["detail_file"] = os.path.relpath(args.detail, ROOT)
        line = _r(line)
        line["value"], line["ms_per_step"] = ups, ms              # the two the driver cross-checks: full precision
        text = json.dumps(line, separators=(",", ":"))
        for k in ("env_steps_per_sec_cfg3", "image_obs_config", "hbm_view", "kernels_us", "hbm_kernels", "other_configs", "env_steps_per_sec"):
            if len(text) <= LINE_LIMIT:
                break
            line.pop(k, None)                                     # never reached with today's keys; the contract keys always stay
            text = json.dumps(line, separators=(",", ":"))
        try:
            os.makedirs(os.path.dirname(args.detail), exist_ok=True)
            with open(args.detail, "w") as f:
                json.dump(dict(detail, line=line), f, indent=1)
        except OSError:
            pass
        print(text, flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
