"""G12 harness: drive dtqn_amd's own run.py loop (prepopulate / train, hence step / evaluate) exactly as the reference's run.py
was driven when tests/golden/make_golden.py wrote G12_loop.npz, log the same events from the outside, and compare.

Shared by the CPU-emulation test (tests/test_loop_golden.py) and the -m gpu test (tests/test_gpu_loop_golden.py)."""
import json
import random

import numpy as np
import torch

from dtqn_amd import _binding as B
from oracle import dtqn_oracle as O


class LoopTrace:
    def __init__(self):
        self.ev = {k: [] for k in ("act_mode", "act_eps", "act_action", "act_greedy", "act_q", "obs_mode", "obs_obs", "obs_action",
                                   "obs_reward", "obs_done", "upd_ep", "upd_start", "upd_stats", "upd_after_act", "flush_after_obs")}
        self.rows, self.log_steps = [], []


def run_loop(fx, name, device, test_lib=None):
    """Returns (LoopTrace, agent, prepopulated buffer arrays) for case `name` of the fixture dict `fx`."""
    import run as runpy
    from dtqn_amd import envs
    from dtqn_amd.agents.dtqn import TrainMode
    from dtqn_amd.networks.dtqn import DTQN
    from dtqn_amd.utils import agent_utils, epsilon_anneal
    from dtqn_amd.utils.logging_utils import RunningAverage
    from dtqn_amd.utils.random import set_global_seed

    g = lambda k: fx[f"{name}/{k}"]
    cfg = json.loads(str(g("cfg")))
    D, H, NL, L = cfg["inner_embed_size"], cfg["num_heads"], cfg["num_layers"], cfg["history_len"]
    seed, Bn, steps = int(g("seed")), int(g("B")), int(g("steps"))
    train_envs, eval_envs = [envs.make("DiscreteCarFlag-v0")], [envs.make("DiscreteCarFlag-v0")]
    set_global_seed(seed, *(train_envs + eval_envs))
    eps = epsilon_anneal.LinearAnneal(1.0, 0.1, steps // 10)
    if test_lib is not None:          # CPU kernel emulation: the network factory needs the test library
        orig = agent_utils.MODEL_MAP["DTQN"]

        def emu_dtqn(*a, **k):
            m = DTQN(*a, _test_lib=test_lib, **k)
            m._allow_cpu = True
            return m
        agent_utils.MODEL_MAP["DTQN"] = emu_dtqn
    try:
        agent = agent_utils.get_agent("DTQN", train_envs, 8, 0, D, int(g("buf_size")), device, float(g("lr")), Bn, L, -1, L, int(g("tuf")),
                                      0.99, H, NL, 0.0, False, "res", "learned", 0, sampler="reference", ref_quirks=True, sample_seed=seed)
    finally:
        if test_lib is not None:
            agent_utils.MODEL_MAP["DTQN"] = orig
    ocfg = O.NetCfg(obs_dim=3, num_actions=3, inner_embed_size=D, num_heads=H, num_layers=NL, history_len=L)
    pol = O.init_params(ocfg, seed=int(g("pol_seed")), perturb=True)
    cs = O.param_checksum(pol)                 # the weights the reference run started from (generator drift guard)
    assert np.isfinite(cs) and abs(cs - float(g("pol_checksum"))) <= 1e-12 * cs
    agent.policy_network.load_state_dict({k: v.clone() for k, v in pol.items()})
    agent.target_update()
    runpy.prepopulate(agent, int(g("prepop")), train_envs)
    rb = agent.replay_buffer
    n_used = min(rb.pos[0] + 1, rb.max_size)
    arrays = rb.export_arrays()
    # copies: on the CPU emulation export_arrays() hands out views of the live buffer
    prepop = {"pos": np.array(rb.pos), "obss": arrays["obss"][:n_used].copy(), "actions": arrays["actions"][:n_used].copy(),
              "rewards": arrays["rewards"][:n_used].copy(), "dones": arrays["dones"][:n_used].astype(bool), "eplens": arrays["eplens"][:n_used].copy()}
    from dtqn_amd.utils.random import RNG
    prepop["rng_probe"] = RNG.rng.bit_generator.state["state"]["state"] % (1 << 53)

    tr = LoopTrace()
    ev = tr.ev
    orig_get, orig_obs, orig_train, orig_flush, orig_idx = agent.get_action, agent.observe, agent.train, rb.flush, rb.sample_indices
    is_eval = lambda: int(agent.train_mode == TrainMode.EVAL)
    draws = {}

    def get_action(epsilon=0.0):
        calls0 = agent._actor_calls
        a = orig_get(epsilon=epsilon)
        greedy = agent._actor_calls != calls0
        ev["act_mode"].append(is_eval()); ev["act_eps"].append(float(epsilon)); ev["act_action"].append(int(a)); ev["act_greedy"].append(greedy)
        ev["act_q"].append(agent._q_np.copy() if greedy else np.full(3, np.nan, dtype=np.float32))
        return a

    def observe(obs, action, reward, done):
        ev["obs_mode"].append(is_eval()); ev["obs_obs"].append(np.asarray(obs, dtype=np.float64).copy()); ev["obs_action"].append(int(action))
        ev["obs_reward"].append(float(reward)); ev["obs_done"].append(bool(done))
        return orig_obs(obs, action, reward, done)

    def sample_indices(bs):
        e, s = orig_idx(bs)
        draws["e"], draws["s"] = np.array(e), np.array(s)
        return e, s

    def train():
        n0 = agent.num_train_steps
        orig_train()
        if agent.num_train_steps != n0:
            ev["upd_ep"].append(draws["e"]); ev["upd_start"].append(draws["s"]); ev["upd_after_act"].append(len(ev["act_action"]))
            agent._drain_stats(block=True)
            ev["upd_stats"].append([agent.td_errors.q[-1], agent.grad_norms.q[-1], agent.qvalue_max.q[-1], agent.qvalue_mean.q[-1],
                                    agent.qvalue_min.q[-1], agent.target_max.q[-1], agent.target_mean.q[-1], agent.target_min.q[-1]])

    def flush():
        ev["flush_after_obs"].append(len(ev["obs_action"]))
        return orig_flush()

    agent.get_action, agent.observe, agent.train, rb.flush, rb.sample_indices = get_action, observe, train, flush, sample_indices

    class Logger:
        def log(self, results, step):
            tr.log_steps.append(int(step))
            tr.rows.append({k: float(v) for k, v in results.items() if k != "losses/hours"})

    runpy.train(agent, train_envs, eval_envs, ["DiscreteCarFlag-v0"], steps, eps, int(g("eval_frequency")), int(g("eval_episodes")),
                "/nonexistent/policy", False, Logger(), RunningAverage(10), RunningAverage(10), RunningAverage(10), None, False)
    tr.final = {"pos": np.array(rb.pos), "eps": float(eps.val), "num_train_steps": int(agent.num_train_steps),
                "rng_probe": RNG.rng.bit_generator.state["state"]["state"] % (1 << 53)}
    return tr, agent, prepop


def _envelope(fx, name):
    """Running maximum, per action event / per update, of the reference's distance to its own one-thread twin (`<name>_t1/*`):
    how far two correct fp32 executions of this trajectory are apart by then."""
    key = f"{name}_t1/ev/act_q"
    if key not in fx:
        return None
    g = lambda k: fx[f"{name}/ev/{k}"]
    t = lambda k: fx[f"{name}_t1/ev/{k}"]
    n = min(len(g("act_q")), len(t("act_q")))
    both = g("act_greedy")[:n] & t("act_greedy")[:n]
    dq = np.where(both, np.nan_to_num(np.abs(g("act_q")[:n] - t("act_q")[:n]).max(axis=1)), 0.0)
    m = min(len(g("upd_stats")), len(t("upd_stats")))
    ds = (np.abs(g("upd_stats")[:m] - t("upd_stats")[:m]) / np.maximum(1.0, np.abs(g("upd_stats")[:m]))).max(axis=1)
    return np.maximum.accumulate(dq), np.maximum.accumulate(ds)


def compare_loop(fx, name, tr, prepop, *, q_tol=1e-4, stat_rtol=2e-4, tie_gap=1e-3, min_actions=None, drift_factor=4.0):
    """The reference's trace vs ours.  Everything up to the first greedy action that differs must agree: events in the same order,
    actions / draws / rewards / dones exactly, Q of the acting row within an ABSOLUTE q_tol (north_star's 1e-4; the runs are
    free-running, so this also bounds the drift of the parameters over all updates so far), statistics within stat_rtol.
    Where the fixture holds the reference's one-thread twin, the bounds widen to `drift_factor` x the reference's own distance to
    that twin so far (a chaotic trajectory: the reference does not reproduce ITSELF more closely than that).  A differing greedy
    action is accepted only where the REFERENCE's own top-two Q gap is below `tie_gap` or below that drift bound, and only after
    `min_actions` matching action events.  Returns a summary dict."""
    env = _envelope(fx, name)
    env_q = (lambda i: drift_factor * float(env[0][min(i, len(env[0]) - 1)])) if env is not None else (lambda i: 0.0)
    env_s = (lambda u: drift_factor * float(env[1][min(u, len(env[1]) - 1)])) if env is not None else (lambda u: 0.0)
    g = lambda k: fx[f"{name}/{k}"]
    # -- after prepopulate (run.py:380-405): the buffer and the exploration stream are bit-equal
    assert np.array_equal(prepop["pos"], g("prepop/pos"))
    assert np.array_equal(prepop["eplens"], g("prepop/eplens"))
    assert np.array_equal(prepop["obss"], g("prepop/obss")), "replay observations after prepopulate differ"
    assert np.array_equal(prepop["actions"], g("prepop/actions"))
    assert np.array_equal(prepop["rewards"], g("prepop/rewards"))
    assert np.array_equal(prepop["dones"], g("prepop/dones"))
    assert prepop["rng_probe"] == int(g("prepop/rng_probe")), "RNG.rng consumed differently during get_agent / prepopulate"
    ref = {k: g(f"ev/{k}") for k in tr.ev}
    n_act_ref = len(ref["act_action"])
    # -- first divergence among the action events
    n_act = min(n_act_ref, len(tr.ev["act_action"]))
    div = None
    for i in range(n_act):
        same = (tr.ev["act_mode"][i] == ref["act_mode"][i] and tr.ev["act_eps"][i] == ref["act_eps"][i]
                and tr.ev["act_greedy"][i] == bool(ref["act_greedy"][i]) and tr.ev["act_action"][i] == ref["act_action"][i])
        if not same:
            div = i
            break
    if div is None and len(tr.ev["act_action"]) != n_act_ref:
        div = n_act
    summary = {"actions_ref": n_act_ref, "first_divergence": div}
    upto_act = n_act_ref if div is None else div
    if div is not None:
        # must be a greedy step on both sides, with identical epsilon / mode, at a near-tie of the reference's own Q
        assert div < n_act, (div, n_act)
        assert tr.ev["act_mode"][div] == ref["act_mode"][div] and tr.ev["act_eps"][div] == ref["act_eps"][div], div
        assert tr.ev["act_greedy"][div] and ref["act_greedy"][div], f"event {div}: epsilon branch differs (RNG stream desynchronised)"
        qs = np.sort(ref["act_q"][div])[::-1]
        gap = float(qs[0] - qs[1])
        summary["divergence_gap"] = gap
        assert gap <= max(tie_gap, 2.0 * env_q(div)), f"greedy action differs at event {div} where the reference's Q gap is {gap} (drift bound {env_q(div)})"
        assert min_actions is not None and div >= min_actions, f"diverged after {div} action events (< {min_actions})"
    # -- Q of the acting row, observe events, flushes, updates: everything that happened before the divergence
    n_upd_ref = len(ref["upd_after_act"])
    upd_ok = [u for u in range(min(n_upd_ref, len(tr.ev["upd_after_act"]))) if ref["upd_after_act"][u] <= upto_act]
    qerr = qrel = 0.0
    for i in range(upto_act):
        if ref["act_greedy"][i]:
            e = float(np.abs(tr.ev["act_q"][i] - ref["act_q"][i]).max())
            tol = max(q_tol, env_q(i))
            qerr, qrel = max(qerr, e), max(qrel, e / tol)
            assert e <= tol, (i, e, tol)
    summary.update({"q_abs_err_max": qerr, "q_err_over_bound": qrel})
    # observe event i follows action event i (one env step each); compare those fully inside the matched prefix
    n_obs = min(upto_act, len(ref["obs_action"]), len(tr.ev["obs_action"]))
    for k in ("obs_mode", "obs_action", "obs_reward", "obs_done"):
        assert np.array_equal(np.array(tr.ev[k][:n_obs]), ref[k][:n_obs]), k
    assert np.array_equal(np.array(tr.ev["obs_obs"][:n_obs]), ref["obs_obs"][:n_obs]), "environment observations differ"
    fl_ref = [f for f in ref["flush_after_obs"] if f <= n_obs]
    assert tr.ev["flush_after_obs"][:len(fl_ref)] == fl_ref
    serr = srel = 0.0
    for u in upd_ok:
        assert tr.ev["upd_after_act"][u] == ref["upd_after_act"][u], ("update placed differently in the loop", u)
        assert np.array_equal(tr.ev["upd_ep"][u], ref["upd_ep"][u]), ("sampler episode draws", u)
        assert np.array_equal(tr.ev["upd_start"][u], ref["upd_start"][u]), ("sampler start draws", u)
        for j, (a, b) in enumerate(zip(tr.ev["upd_stats"][u], ref["upd_stats"][u])):
            e = abs(a - b) / max(1.0, abs(b))
            tol = max(stat_rtol, env_s(u))
            if tol > stat_rtol and j in (2, 4, 5, 7):
                continue          # max / min statistics jump when a single element (an argmax choice) flips: only pinned while the bound is the strict one
            serr, srel = max(serr, e), max(srel, e / tol)
            assert e <= tol, (u, j, a, b, tol)
    summary.update({"updates_compared": len(upd_ok), "stat_rel_err_max": serr, "stat_err_over_bound": srel, "obs_events_compared": n_obs})
    if div is None:
        ref_rows = json.loads(str(g("log_rows")))
        assert tr.log_steps == list(g("log_steps"))
        for mine, theirs in zip(tr.rows, ref_rows):
            for k, v in theirs.items():
                if k.startswith("losses/"):
                    assert abs(mine[k] - v) <= 5e-3 * max(1.0, abs(v)), (k, mine[k], v)
                else:
                    assert mine[k] == v, (k, mine[k], v)
        assert np.array_equal(tr.final["pos"], g("final/pos")) and tr.final["eps"] == float(g("final/eps"))
        assert tr.final["num_train_steps"] == int(g("final/num_train_steps"))
        assert tr.final["rng_probe"] == int(g("final/rng_probe"))
    return summary
