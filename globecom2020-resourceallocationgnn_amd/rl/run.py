"""Evaluation driver: the counterpart of the reference's RL_Run_main.py (load_trained_model :105-148, run_test :151-)
-- load the weights a training run saved, then compare the greedy policy with the random-action baseline and
(optionally) the brute-force optimum.

    python -m v2xgnn.rl.train --links 4 --episodes 5 --train-steps 20 --batch 512 --save-dir runs/a
    python -m v2xgnn.rl.run   --links 4 --episodes 5 --train-steps 20 --batch 512 --save-dir runs/a \\
                              --test-episodes 10 --test-steps 50 --opt
"""
import argparse
import json
import os
import random

import numpy as np

from .agent import Agent
from .sim_config import RL_Config
from .train import start_env


def weight_file_names(num_episodes, num_train_steps, batch_size):
    """The reference's naming scheme (BS_brain.py:859-868, RL_Run_main.py:135-141)."""
    tag = '-Episode-%d-Step-%d-Batch-%d.h5' % (num_episodes, num_train_steps, batch_size)
    return 'Q-Network_model_weights' + tag, 'Target-Network_model_weights' + tag


def load_trained_model(env, cfg, model_dir, brain=None, **brain_kwargs):
    """RL_Run_main.py:105-148"""
    agent = Agent(env.n_Veh, env.n_RB, env.n_Neighbor, cfg.Num_Feedback, env, cfg, brain=brain, **brain_kwargs)
    online, target = weight_file_names(cfg.Num_Episodes, cfg.Num_Train_Steps, cfg.Batch_Size)
    agent.brain.model.load_weights(os.path.join(model_dir, online))
    agent.brain.target_model.load_weights(os.path.join(model_dir, target))
    return agent


def run_test(cfg, agent):
    """RL_Run_main.py:151-: -> dict of the test_run outputs plus the mean rewards per scheme."""
    out = agent.test_run(cfg.Num_Run_Episodes, cfg.Num_Test_Steps, cfg.Opt_Flag)
    names = ['Expect_Return', 'Reward', 'Per_V2V_Rate', 'Per_V2I_Rate', 'Per_V2B_Interference']
    res = {}
    for prefix, chunk in zip(('', 'RA_', 'Opt_'), (out[0:5], out[5:10], out[10:15])):
        for name, arr in zip(names, chunk):
            res[prefix + name] = arr
    return res


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--links", type=int, default=4)
    ap.add_argument("--feedback", type=int, default=16)
    ap.add_argument("--gamma", type=float, default=0.5)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--episodes", type=int, default=1, help="episode count of the training run whose weights are loaded")
    ap.add_argument("--train-steps", type=int, default=20)
    ap.add_argument("--save-dir", required=True)
    ap.add_argument("--test-episodes", type=int, default=10)
    ap.add_argument("--test-steps", type=int, default=50)
    ap.add_argument("--opt", action="store_true", help="also run the brute-force optimum (C^N joint actions)")
    ap.add_argument("--seed", type=int, default=11)
    args = ap.parse_args(argv)
    if args.links < 4 or args.links % 4:
        # the simulator drops vehicles in groups of four, one per direction (Environment.py:217-231), and the
        # observation divides by links - 2 (BS_brain.py:405)
        ap.error("--links must be a multiple of 4 and at least 4 (got %d)" % args.links)
    random.seed(args.seed)
    np.random.seed(args.seed)
    cfg = RL_Config()
    cfg.set_train_value(args.feedback, args.gamma, args.batch, 1, 0.1)
    cfg.Num_Episodes, cfg.Num_Train_Steps = args.episodes, args.train_steps
    cfg.set_test_values(args.test_episodes, args.test_steps, args.opt, 1, 0.1)
    env = start_env(args.links)
    agent = load_trained_model(env, cfg, args.save_dir, seed=args.seed)
    res = run_test(cfg, agent)
    summary = {"links": args.links, "test_episodes": args.test_episodes, "test_steps": args.test_steps,
               "mean_reward_gnn": float(res['Reward'].mean()), "mean_reward_random": float(res['RA_Reward'].mean())}
    if args.opt:
        summary["mean_reward_optimal"] = float(res['Opt_Reward'].mean())
    print(json.dumps(summary))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
