#!/usr/bin/env python3
"""Experiment driver with the reference's command line (run.py:16-184 flags, :408-529 flow),
running the DTQN hot path on the MI355X engine (dtqn_amd).

    python run.py --envs DiscreteCarFlag-v0 --in-embed 64 --disable-wandb --verbose

Differences from the reference, all at the edges of the hot path:
  * `--envs` defaults to the LIST ["DiscreteCarFlag-v0"] (the reference's string default iterates
    over characters);
  * `--model` accepts DTQN only; `--render` and image domains are out of scope;
  * new flags: `--sampler {reference,device}` (replay index draw on the host with Python's `random`
    stream like the reference, or on the GPU), `--ref-quirks` (reproduce the reference's
    int-truncated actor context), `--prepopulate N` (the reference hard-codes 50 000);
  * under `python -m torch.distributed.run` every rank trains its own env / replay shard and the
    flat gradient is all-reduced over RCCL (rank 0 logs and evaluates).
"""
import argparse
import os
from time import time
from typing import Optional, Sequence

import torch

from dtqn_amd import dist as ddp
from dtqn_amd.utils import env_processing, epsilon_anneal
from dtqn_amd.utils.agent_utils import MODEL_MAP, get_agent
from dtqn_amd.utils.logging_utils import RunningAverage, get_logger, timestamp
from dtqn_amd.utils.random import RNG, set_global_seed


def get_args(argv: Optional[Sequence[str]] = None):
    p = argparse.ArgumentParser(description="DTQN on MI355X")
    p.add_argument("--project-name", type=str, default="DTQN-test", help="wandb project / local results directory")
    p.add_argument("--disable-wandb", action="store_true", help="log to CSV files instead of wandb")
    p.add_argument("--time-limit", type=float, default=None, help="hours before checkpointing and exiting")
    p.add_argument("--model", type=str, default="DTQN", choices=list(MODEL_MAP.keys()))
    p.add_argument("--envs", type=str, nargs="+", default=["DiscreteCarFlag-v0"],
                   help="one or more domains with identical observation / action spaces")
    p.add_argument("--num-steps", type=int, default=2_000_000, help="environment steps to train for")
    p.add_argument("--tuf", type=int, default=10_000, help="hard target-update period (in updates)")
    p.add_argument("--lr", type=float, default=3e-4)
    p.add_argument("--batch", type=int, default=32)
    p.add_argument("--buf-size", type=int, default=500_000, help="replay capacity in transitions")
    p.add_argument("--eval-frequency", type=int, default=5_000)
    p.add_argument("--eval-episodes", type=int, default=10)
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--context", type=int, default=50, help="context length L")
    p.add_argument("--obs-embed", type=int, default=8, help="per-dimension embedding width (discrete observations)")
    p.add_argument("--a-embed", type=int, default=0, help="action embedding width (0 = none)")
    p.add_argument("--in-embed", type=int, default=128, help="d_model")
    p.add_argument("--max-episode-steps", type=int, default=-1)
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--save-policy", action="store_true")
    p.add_argument("--verbose", action="store_true")
    p.add_argument("--render", action="store_true")
    p.add_argument("--history", type=int, default=50, help="number of trailing Q-values trained per window")
    p.add_argument("--heads", type=int, default=8)
    p.add_argument("--layers", type=int, default=2)
    p.add_argument("--dropout", type=float, default=0.0)
    p.add_argument("--discount", type=float, default=0.99)
    p.add_argument("--gate", type=str, default="res", choices=["res", "gru"])
    p.add_argument("--identity", action="store_true")
    p.add_argument("--pos", default="learned", choices=["learned", "sin", "none"])
    p.add_argument("--bag-size", type=int, default=0)
    p.add_argument("--slurm-job-id", default=0, type=str)
    # --- additions ---
    p.add_argument("--sampler", default="reference", choices=["reference", "device"])
    p.add_argument("--ref-quirks", action="store_true", help="reproduce the reference's int-truncated actor context")
    p.add_argument("--prepopulate", type=int, default=50_000, help="random steps before training (reference: 50 000)")
    p.add_argument("--num-envs", type=int, default=1,
                   help="host environments per learner (vectorised rollout): N > 1 steps N environments per vector step with ONE "
                        "batched actor launch and keeps the reference's 1 env step : 1 update ratio by running N updates per "
                        "vector step (all N actions of a vector step come from the same parameters)")
    p.add_argument("--overlap", action="store_true",
                   help="pipeline the actor forward of step t+1 with TD update t+1 on two HIP streams (same policy-vs-action "
                        "semantics; an episode that ends at step t becomes sampleable one update later)")
    return p.parse_args(argv)


TIME_CHECK_PERIOD = 256      # data-parallel runs: steps between collective --time-limit votes


def evaluate(agent, eval_env, eval_episodes: int, render: Optional[bool] = None):
    """Greedy evaluation: (success rate, mean return, mean episode length)  (reference run.py:187-243)."""
    if render:
        raise NotImplementedError("--render is outside dtqn_amd's scope")
    agent.eval_on()
    returns = successes = steps = 0
    for _ in range(eval_episodes):
        agent.context_reset(eval_env.reset())
        done, ep_return, info = False, 0, {}
        while not done:
            action = agent.get_action(epsilon=0.0)
            obs, reward, done, info = eval_env.step(action)
            agent.observe(obs, action, reward, done)
            ep_return += reward
        returns += ep_return
        steps += agent.context.timestep
        successes += int(info.get("is_success", False) or ep_return > 0)
    agent.eval_off()
    n = max(eval_episodes, 1)
    return successes / n, returns / n, steps / n


def step(agent, env, eps) -> bool:
    """One epsilon-greedy env step; TimeLimit truncation is not stored as a terminal (run.py:356-377)."""
    action = agent.get_action(epsilon=eps.val)
    obs, reward, done, info = env.step(action)
    agent.observe(obs, action, reward, False if info.get("TimeLimit.truncated", False) else done)
    return done


def step_overlapped(agent, env, eps) -> bool:
    """step() + train() with the actor forward and the TD update running concurrently on the GPU."""
    pending = agent.begin_action(epsilon=eps.val)      # actor stream, behind the previous update
    agent.train()                                      # learner stream; its optimizer kernel waits for the actor
    action = agent.finish_action(pending)
    obs, reward, done, info = env.step(action)
    agent.observe(obs, action, reward, False if info.get("TimeLimit.truncated", False) else done)
    return done


def prepopulate(agent, prepop_steps: int, envs) -> None:
    """Fill the replay buffer with uniformly random behaviour (run.py:380-405)."""
    t = 0
    while t < prepop_steps:
        env = RNG.rng.choice(envs)
        agent.context_reset(env.reset())
        done = False
        while not done:
            action = RNG.rng.integers(env.action_space.n)
            obs, reward, done, info = env.step(action)
            agent.observe(obs, action, reward, False if info.get("TimeLimit.truncated", False) else done)
            t += 1
        agent.replay_buffer.flush()


def train(agent, envs, eval_envs, env_strs, total_steps, eps, eval_frequency, eval_episodes, policy_path, save_policy,
          logger, mean_success_rate, mean_episode_length, mean_reward, time_remaining, verbose=False, is_main=True,
          overlap=False, vector=None):
    """Main loop: one env step, one TD update (run.py:246-353).  vector: a VectorActor over N environments; then one iteration in
    N steps all N environments at once with that vector step's N updates queued behind the actor forward (the other N - 1
    iterations only account for them), which keeps one update per env step.  The vector steps are driven by the updates still
    owed, not by `timestep % N`: a resumed run steps on its first iteration, and the time limit is only honoured at a
    vector-step boundary, so a checkpoint's `num_train_steps` always equals the loop's timestep."""
    start = time()
    agent.eval_off()
    if vector is None:
        env = RNG.rng.choice(envs)
        agent.context_reset(env.reset())
    else:
        vector.reset_all()
    owed = 0                 # updates of the current vector step the loop has not accounted for yet
    last_vote = -1           # data parallel: period index of the last time-limit vote
    for timestep in range(agent.num_train_steps, total_steps):
        if vector is not None:
            if owed == 0:                      # N env steps and the N updates that go with them (queued behind the actor forward)
                owed = min(vector.n, total_steps - timestep)
                vector.step_all(eps.val, updates=owed)
            owed -= 1
        elif overlap:
            if step_overlapped(agent, env, eps):       # includes this step's train()
                agent.replay_buffer.flush()
                env = RNG.rng.choice(envs)
                agent.context_reset(env.reset())
        else:
            if step(agent, env, eps):
                agent.replay_buffer.flush()
                env = RNG.rng.choice(envs)
                agent.context_reset(env.reset())
            agent.train()
        eps.anneal()
        if timestep % eval_frequency == 0:
            # data parallel: EVERY rank evaluates (on its own evaluation environments) so that no replica sits blocked in the next
            # gradient all-reduce while rank 0 plays its episodes; only the main rank logs
            hours = (time() - start) / 3600
            log = {"losses/TD_Error": agent.td_errors.mean(), "losses/Grad_Norm": agent.grad_norms.mean(),
                   "losses/Max_Q_Value": agent.qvalue_max.mean(), "losses/Mean_Q_Value": agent.qvalue_mean.mean(),
                   "losses/Min_Q_Value": agent.qvalue_min.mean(), "losses/Max_Target_Value": agent.target_max.mean(),
                   "losses/Mean_Target_Value": agent.target_mean.mean(), "losses/Min_Target_Value": agent.target_min.mean(),
                   "losses/hours": hours}
            for env_str, eval_env in zip(env_strs, eval_envs):
                sr, ret, length = evaluate(agent, eval_env, eval_episodes)
                log.update({f"{env_str}/SuccessRate": sr, f"{env_str}/Return": ret, f"{env_str}/EpisodeLength": length})
                if verbose and is_main:
                    print(f"[ {timestamp()} ] Training Steps: {timestep}, Env: {env_str}, Success Rate: {sr:.2f}, "
                          f"Return: {ret:.2f}, Episode Length: {length:.2f}, Hours: {hours:.2f}", flush=True)
            if is_main:
                logger.log(log, step=timestep)
        if save_policy and timestep % 50_000 == 0 and is_main:
            torch.save(agent.policy_network.state_dict(), policy_path)
        if time_remaining and owed == 0:
            # one process: the wall clock decides.  Data parallel: every rank must leave on the SAME iteration (the others
            # would block in the gradient all-reduce), so the ranks vote once per TIME_CHECK_PERIOD steps -- at the first
            # vector-step boundary (owed == 0) inside each period: with N environments those boundaries sit at timestep
            # = N - 1 (mod N) and need never coincide with a multiple of the period.  Every rank runs the same loop, so every rank
            # evaluates the same condition on the same iteration.
            if not ddp.is_distributed():
                stop = time() - start >= time_remaining
            elif timestep // TIME_CHECK_PERIOD != last_vote:
                last_vote = timestep // TIME_CHECK_PERIOD
                stop = ddp.agree_any(time() - start >= time_remaining, agent.device)
            else:
                stop = False
            if stop:
                if is_main:
                    print(f"Reached time limit. Saving checkpoint with {agent.num_train_steps} steps completed.")
                # every rank writes: rank 0 the full checkpoint, replicas their replay shard and RNG streams
                agent.save_checkpoint(policy_path, None, mean_success_rate, mean_reward, mean_episode_length, eps)
                return


def run_experiment(args):
    start = time()
    if os.environ.get("DTQN_DEBUG_HANG_DUMP"):       # debugging aid: dump every thread's Python stack after N seconds (a rank stuck in a collective)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["DTQN_DEBUG_HANG_DUMP"]), repeat=False)
    rank, world, local = ddp.init_from_env("cuda" if args.device.startswith("cuda") else "cpu")
    is_main = rank == 0
    device = torch.device(args.device if world == 1 or not args.device.startswith("cuda") else f"cuda:{local}")
    envs = [env_processing.make_env(e) for e in args.envs]
    eval_envs = [env_processing.make_env(e) for e in args.envs]
    set_global_seed(args.seed + rank, *(envs + eval_envs))         # per-rank env / replay / exploration streams
    eps = epsilon_anneal.LinearAnneal(1.0, 0.1, args.num_steps // 10)
    agent = get_agent(args.model, envs, args.obs_embed, args.a_embed, args.in_embed, args.buf_size, device, args.lr,
                      args.batch, args.context, args.max_episode_steps, args.history, args.tuf, args.discount,
                      args.heads, args.layers, args.dropout, args.identity, args.gate, args.pos, args.bag_size,
                      sampler=args.sampler, ref_quirks=args.ref_quirks, sample_seed=args.seed + rank)
    if is_main:
        print(f"[ {timestamp()} ] Creating {args.model} with "
              f"{sum(p.numel() for p in agent.policy_network.parameters())} parameters", flush=True)
    save_dir = os.path.join(os.getcwd(), "policies", args.project_name, *args.envs)
    os.makedirs(save_dir, exist_ok=True)
    policy_path = os.path.join(
        save_dir,
        f"model={args.model}_envs={','.join(args.envs)}_obs_embed={args.obs_embed}_a_embed={args.a_embed}_"
        f"in_embed={args.in_embed}_context={args.context}_heads={args.heads}_layers={args.layers}_batch={args.batch}_"
        f"gate={args.gate}_identity={args.identity}_history={args.history}_pos={args.pos}_bag={args.bag_size}_seed={args.seed}")
    if args.render:
        raise NotImplementedError("--render is outside dtqn_amd's scope")
    if os.path.exists(policy_path + "_mini_checkpoint.pt"):
        done_steps = agent.load_mini_checkpoint(policy_path)["step"]
        print(f"Found a mini checkpoint that completed {done_steps} training steps.")
        if done_steps >= args.num_steps:
            print("Removing checkpoint and exiting...")
            if os.path.exists(policy_path + "_checkpoint.pt"):
                os.remove(policy_path + "_checkpoint.pt")
            raise SystemExit(0)
        wandb_id, mean_success_rate, mean_reward, mean_episode_length, eps.val = agent.load_checkpoint(policy_path)
        wandb_kwargs = {"resume": "must", "id": wandb_id}
    else:
        wandb_kwargs = {"resume": None}
        prepopulate(agent, args.prepopulate, envs)
        mean_success_rate, mean_reward, mean_episode_length = RunningAverage(10), RunningAverage(10), RunningAverage(10)
    logger = get_logger(policy_path, args, wandb_kwargs) if is_main else None
    time_remaining = args.time_limit * 3600 - (time() - start) if args.time_limit else None
    vector = None
    if args.num_envs > 1:
        from dtqn_amd.agents.vector import VectorActor
        # N copies of the (first) training domain, each with its own seed
        venvs = [env_processing.make_env(args.envs[k % len(args.envs)]) for k in range(args.num_envs)]
        for k, e in enumerate(venvs):
            e.seed(args.seed + 1000 * (rank + 1) + k)
        vector = VectorActor(agent, venvs, ref_quirks=args.ref_quirks)
    train(agent, envs, eval_envs, args.envs, args.num_steps, eps, args.eval_frequency, args.eval_episodes, policy_path,
          args.save_policy, logger, mean_success_rate, mean_reward, mean_episode_length, time_remaining, args.verbose, is_main,
          overlap=args.overlap, vector=vector)
    if is_main:
        agent.save_mini_checkpoint(checkpoint_dir=policy_path, wandb_id=None)
    return agent


if __name__ == "__main__":
    run_experiment(get_args())
