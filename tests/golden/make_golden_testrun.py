#!/usr/bin/env python3
"""Capture the reference's evaluation loop `Agent.test_run` (BS_brain.py:986-1162: greedy policy vs the random-action
and brute-force baselines) on the reference simulator with the recording fake brain of make_golden.py.

Runs ONLY in the build container (imports /root/reference); writes tests/golden/golden_testrun_n4.npz.
    python tests/golden/make_golden_testrun.py
"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                               # noqa: E402  (stubs + FakeBS + make_env)

NAMES = ['Expect_Return', 'Reward', 'Per_V2V_Rate', 'Per_V2I_Rate', 'Per_V2B_Interference',
         'RA_Expect_Return', 'RA_Reward', 'RA_Per_V2V_Rate', 'RA_Per_V2I_Rate', 'RA_Per_V2B_Interference',
         'Opt_Expect_Return', 'Opt_Reward', 'Opt_Per_V2V_Rate', 'Opt_Per_V2I_Rate', 'Opt_Per_V2B_Interference']


def main():
    mg.install_stubs()
    sys.path.insert(0, mg.REF)
    import Environment
    import Sim_Config
    import BS_brain
    seed = 4242
    random.seed(seed)
    np.random.seed(seed)
    cfg = Sim_Config.RL_Config()
    cfg.set_train_value(16, 0.5, 32, 1, 0.1)
    env = mg.make_env(Environment)
    BS_brain.BS = mg.FakeBS
    BS_brain.Memory.samples = []
    agent = BS_brain.Agent(env.n_Veh, env.n_RB, env.n_Neighbor, cfg.Num_Feedback, env, cfg)
    out = agent.test_run(3, 7, True)
    fx = {'seed': seed, 'episodes': 3, 'steps': 7}
    for name, v in zip(NAMES, out):
        fx[name] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, 'golden_testrun_n4.npz'), **fx)
    print('wrote golden_testrun_n4.npz', {k: np.asarray(v).shape for k, v in fx.items()})

    # ---- observation at 20 links: Agent.get_state is written for any num_D2D (BS_brain.py:389-407); only the
    #      network builder and the payload loops are hard-coded to 4 links
    random.seed(2020)
    np.random.seed(2020)
    env = mg.make_env(Environment)
    env.new_random_game(20)
    BS_brain.Memory.samples = []
    agent = BS_brain.Agent(20, env.n_RB, env.n_Neighbor, cfg.Num_Feedback, env, cfg)
    obs = {'v2v': [], 'v2i': [], 'edge': [], 'dest': []}
    for step in range(3):
        v2v, v2i, edge = zip(*[agent.get_state([k, 0]) for k in range(20)])
        obs['v2v'].append(np.stack(v2v)); obs['v2i'].append(np.stack(v2i)); obs['edge'].append(np.stack(edge))
        obs['dest'].append(np.array([v.destinations[0] for v in env.vehicles]))
        a = np.random.randint(0, env.n_RB, size=(20, 1))
        env.compute_reward_with_channel_selection(a.copy())
        env.renew_positions()
        env.renew_channels_fastfading()
        env.Compute_Interference(a.copy())
    np.savez_compressed(os.path.join(HERE, 'golden_observe_n20.npz'), **{k: np.stack(v) for k, v in obs.items()})
    print('wrote golden_observe_n20.npz')

    # ---- simulator at 100 links (BASELINE config 4's graph size): positions in full, channels as a strided sample
    random.seed(2100)
    np.random.seed(2100)
    env = mg.make_env(Environment)
    env.new_random_game(100)
    big = {'pos': [], 'dest': [], 'v2v_sample': [], 'v2v_sum': [], 'v2i': [], 'v2v_rate': [], 'v2i_rate': []}
    for step in range(3):
        a = np.random.randint(0, env.n_RB, size=(100, 1))
        r_v2v, r_v2i, _ = env.compute_reward_with_channel_selection(a.copy())
        env.renew_positions()
        env.renew_channels_fastfading()
        env.Compute_Interference(a.copy())
        big['pos'].append(np.array([v.position for v in env.vehicles], float))
        big['dest'].append(np.array([v.destinations[0] for v in env.vehicles]))
        big['v2v_sample'].append(env.V2V_channels_with_fastfading[::7, ::11, :].copy())
        big['v2v_sum'].append(env.V2V_channels_with_fastfading.sum(axis=(0, 1)))
        big['v2i'].append(env.V2I_channels_with_fastfading.copy())
        big['v2v_rate'].append(r_v2v.copy()); big['v2i_rate'].append(r_v2i.copy())
    np.savez_compressed(os.path.join(HERE, 'golden_env_n100.npz'), **{k: np.stack(v) for k, v in big.items()})
    print('wrote golden_env_n100.npz')


if __name__ == '__main__':
    main()
