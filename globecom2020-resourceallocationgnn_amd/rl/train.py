"""Training driver: the counterpart of the reference's RL_Train_main.py (main :21-75, start_env :78-95,
run_train :98-118) on the engine, for any number of V2V links and -- launched with torch.distributed.run --
data-parallel over the GPUs of one node (BASELINE.json configs[0] and configs[2]).

    python -m v2xgnn.rl.train --links 4 --feedback 16 --batch 512 --episodes 2 --train-steps 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m v2xgnn.rl.train --links 20 --feedback 64 --batch 4096 --episodes 1 --train-steps 20

Data-parallel scheme (--rollouts replicated, the default): every rank runs the SAME seeded simulator and replay sampling (a few hundred microseconds of
numpy per step), so all ranks hold the identical minibatch without any broadcast; each fit step takes the rank's
contiguous shard of it and all-reduces the gradient (v2xgnn.dp).  Weights therefore stay bit-identical on all ranks
and the epsilon-greedy rollouts stay in lock-step.  --rollouts sharded gives every rank its own simulator and replay
memory and 50/G of the transitions of a train step (Agent docstring): the simulator work per rank drops by G.
"""
import argparse
import json
import os
import random
import time

import numpy as np

from .agent import Agent
from .environment import Environ
from .sim_config import RL_Config


def start_env(n_links=None):
    """Lane grid of RL_Train_main.py:82-88; n_links: number of V2V links (default: the simulator's 4)."""
    up_lanes = [3.5 / 2, 3.5 / 2 + 3.5, 250 + 3.5 / 2, 250 + 3.5 + 3.5 / 2, 500 + 3.5 / 2, 500 + 3.5 + 3.5 / 2]
    down_lanes = [250 - 3.5 - 3.5 / 2, 250 - 3.5 / 2, 500 - 3.5 - 3.5 / 2, 500 - 3.5 / 2, 750 - 3.5 - 3.5 / 2, 750 - 3.5 / 2]
    left_lanes = [3.5 / 2, 3.5 / 2 + 3.5, 433 + 3.5 / 2, 433 + 3.5 + 3.5 / 2, 866 + 3.5 / 2, 866 + 3.5 + 3.5 / 2]
    right_lanes = [433 - 3.5 - 3.5 / 2, 433 - 3.5 / 2, 866 - 3.5 - 3.5 / 2, 866 - 3.5 / 2, 1299 - 3.5 - 3.5 / 2, 1299 - 3.5 / 2]
    env = Environ(down_lanes, up_lanes, left_lanes, right_lanes, 750, 1299)
    env.new_random_game(env.n_Veh)
    if n_links is not None and n_links != env.n_Veh:
        env.new_random_game(n_links)
    return env


def start_env_batched(n_links, n_envs, seed, lookahead=None):
    """E environments on the same lane grid, environment e seeded with seed + 104729 e (rl/batched_env.py).
    lookahead: compute every next simulator step on the library's worker thread while the agent scores and replays (same
    trajectories, see BatchedEnviron); default on, V2X_SIM_LOOKAHEAD=0 switches it off."""
    import os
    if lookahead is None:
        lookahead = os.environ.get("V2X_SIM_LOOKAHEAD", "1") != "0"
    from .batched_env import BatchedEnviron
    up_lanes = [3.5 / 2, 3.5 / 2 + 3.5, 250 + 3.5 / 2, 250 + 3.5 + 3.5 / 2, 500 + 3.5 / 2, 500 + 3.5 + 3.5 / 2]
    down_lanes = [250 - 3.5 - 3.5 / 2, 250 - 3.5 / 2, 500 - 3.5 - 3.5 / 2, 500 - 3.5 / 2, 750 - 3.5 - 3.5 / 2, 750 - 3.5 / 2]
    left_lanes = [3.5 / 2, 3.5 / 2 + 3.5, 433 + 3.5 / 2, 433 + 3.5 + 3.5 / 2, 866 + 3.5 / 2, 866 + 3.5 + 3.5 / 2]
    right_lanes = [433 - 3.5 - 3.5 / 2, 433 - 3.5 / 2, 866 - 3.5 - 3.5 / 2, 866 - 3.5 / 2, 1299 - 3.5 - 3.5 / 2, 1299 - 3.5 / 2]
    env = BatchedEnviron(down_lanes, up_lanes, left_lanes, right_lanes, 750, 1299, n_envs=n_envs,
                         seeds=[seed + 104729 * e for e in range(n_envs)])
    env.lookahead = bool(lookahead) and env.native
    env.new_random_game(n_links)
    return env


def run_train(env, cfg, brain=None, save_dir=None, verbose=False, **brain_kwargs):
    """RL_Train_main.py:98-118"""
    agent = Agent(env.n_Veh, env.n_RB, env.n_Neighbor, cfg.Num_Feedback, env, cfg, brain=brain, **brain_kwargs)
    out = agent.train(cfg.Num_Episodes, cfg.Num_Train_Steps, save_dir=save_dir, verbose=verbose)
    return agent, out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--links", type=int, default=4)
    ap.add_argument("--feedback", type=int, default=16)
    ap.add_argument("--gamma", type=float, default=0.5)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--episodes", type=int, default=1)
    ap.add_argument("--train-steps", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1001)           # RL_Train_main.py:44
    ap.add_argument("--save-dir", default=None)
    ap.add_argument("--use-graph", action="store_true")
    ap.add_argument("--envs", type=int, default=0,
                    help="step this many independent simulators as arrays (rl/batched_env.py) instead of one")
    ap.add_argument("--rollouts", choices=["replicated", "sharded"], default="replicated",
                    help="data-parallel runs: every rank steps the same simulator (bit-identical to one process) or every "
                         "rank its own, contributing 50/G transitions per train step (Agent docstring)")
    args = ap.parse_args(argv)
    if args.links < 4 or args.links % 4:
        # the simulator drops vehicles in groups of four, one per direction (Environment.py:217-231), and the
        # observation divides by links - 2 (BS_brain.py:405)
        ap.error("--links must be a multiple of 4 and at least 4 (got %d)" % args.links)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    force_dp = os.environ.get("V2X_FORCE_DP") == "1"            # run the RCCL path with a single rank (tests)
    if world > 1 or force_dp:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29544")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    sharded = args.rollouts == "sharded" and world > 1
    random.seed(args.seed + (7919 * rank if sharded else 0))      # sharded rollouts: one simulator / exploration stream per rank
    np.random.seed(args.seed + (7919 * rank if sharded else 0))
    cfg = RL_Config()
    cfg.set_train_value(args.feedback, args.gamma, args.batch, 1, 0.1)       # RL_Train_main.py:33-35,60
    cfg.Num_Episodes, cfg.Num_Train_Steps = args.episodes, args.train_steps
    env = (start_env_batched(args.links, args.envs, args.seed + (7919 * rank if sharded else 0)) if args.envs > 0
           else start_env(args.links))
    t0 = time.perf_counter()
    import contextlib
    ctx = contextlib.nullcontext()
    if args.use_graph:
        # the engine captures hipGraphs only on a non-default stream (capture is not allowed on the legacy stream)
        import torch
        torch.cuda.set_device(local)
        ctx = torch.cuda.stream(torch.cuda.Stream(device=local))
    with ctx:
        agent, (loss, reward_step, reward_ep, q_mean, q_max, _, _) = run_train(
            env, cfg, save_dir=args.save_dir if rank == 0 else None, verbose=rank == 0,
            device=local, seed=args.seed, use_graph=args.use_graph, data_parallel=world > 1 or force_dp,
            rollouts=args.rollouts)
    dt = time.perf_counter() - t0
    if rank == 0:
        n_fit = args.episodes * args.train_steps
        print(json.dumps({"links": args.links, "feat_dim": args.feedback, "batch": args.batch, "n_gpus": world,
                          "episodes": args.episodes, "train_steps": args.train_steps, "env_steps": int(agent.num_step),
                          "wall_s": round(dt, 3), "fit_steps_per_s": round(n_fit / dt, 2),
                          "mean_loss_last_episode": [round(float(v), 6) for v in loss[:, -1].mean(axis=1)],
                          "reward_last_episode": round(float(reward_ep[-1]), 4)}))
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
