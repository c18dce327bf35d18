"""Evaluation-of-training driver: the counterpart of the reference's RL_Evaluated_main_Epsilon_DiffTrails.py (main :17-,
run_test -> Agent.evaluate_training_diff_trials, BS_brain.py:1164-1451): every checkpoint a training run saved (one per 5
episodes) is evaluated under a fixed epsilon-greedy policy over several re-seeded trials, next to the random-action
baseline and the brute-force optimum.

    python -m v2xgnn.rl.train    --links 4 --episodes 10 --train-steps 20 --batch 512 --save-dir runs/a
    python -m v2xgnn.rl.evaluate --links 4 --episodes 10 --batch 512 --save-dir runs/a --test-steps 100 --trials 10
"""
import argparse
import json
import random

import numpy as np

from .agent import Agent
from .sim_config import RL_Config
from .train import start_env


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--links", type=int, default=4)
    ap.add_argument("--feedback", type=int, default=16)
    ap.add_argument("--gamma", type=float, default=0.05)          # RL_Evaluated_main_Epsilon_DiffTrails.py:25
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--episodes", type=int, default=10, help="training episodes of the run (a checkpoint every 5)")
    ap.add_argument("--train-steps", type=int, default=20)
    ap.add_argument("--save-dir", required=True)
    ap.add_argument("--test-steps", type=int, default=100)        # :42
    ap.add_argument("--trials", type=int, default=10)             # :39
    ap.add_argument("--epsilon", type=float, default=0.0)         # :37
    ap.add_argument("--opt", action="store_true")
    ap.add_argument("--seed", type=int, default=1)                # :22
    args = ap.parse_args(argv)
    if args.links < 4 or args.links % 4:
        ap.error("--links must be a multiple of 4 and at least 4 (got %d)" % args.links)
    random.seed(args.seed)
    np.random.seed(args.seed)
    cfg = RL_Config()
    cfg.set_train_value(args.feedback, args.gamma, args.batch, 1, 0.1)
    env = start_env(args.links)
    agent = Agent(env.n_Veh, env.n_RB, env.n_Neighbor, cfg.Num_Feedback, env, cfg, seed=args.seed, device_replay=False)
    out = agent.evaluate_training_diff_trials(args.episodes, args.test_steps, args.opt, args.epsilon, args.trials,
                                              model_dir=args.save_dir, num_train_steps=args.train_steps)
    ret, ra = (out[0], out[2]) if args.opt else (out[1], out[3])
    summary = {"links": args.links, "checkpoints": int(ret.shape[1]), "trials": args.trials,
               "mean_return_per_checkpoint": [round(float(v), 4) for v in ret.mean(axis=0)],
               "mean_return_random": round(float(ra.mean()), 4)}
    if args.opt:
        summary["mean_return_optimal"] = round(float(out[4].mean()), 4)
    else:
        summary["optimal_return_per_trial"] = [round(float(v), 4) for v in out[0]]
    print(json.dumps(summary))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
