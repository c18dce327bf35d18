#!/usr/bin/env python3
"""Capture the reference's `Agent.evaluate_training_diff_trials` (BS_brain.py:1164-1451: fixed-epsilon evaluation of
every saved checkpoint over several re-seeded trials, next to the random-action and brute-force baselines) on the
reference simulator with the recording fake brain of make_golden.py (its `load_weights` is a no-op: the loop, the RNG
consumption and the bookkeeping are what is pinned, not a network).

Runs ONLY in the build container (imports /root/reference); writes tests/golden/golden_evaltrials_n4.npz.
    python tests/golden/make_golden_evaltrials.py
"""
import os
import random
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                               # noqa: E402  (stubs + FakeBS + make_env)

NAMES_OPT = ['Return', 'Reward', 'RA_Return', 'RA_Reward', 'Opt_Return', 'Opt_Reward', 'Opt_V2V', 'Opt_V2I', 'Opt_Interference']
NAMES_RA = ['Evaluated_Opt_Return', 'Return', 'Reward', 'RA_Return', 'RA_Reward']


class FakeBSWithFiles(mg.FakeBS):
    def __init__(self, *a):
        super().__init__(*a)
        self.loads = []
        self.model = types.SimpleNamespace(load_weights=lambda path: self.loads.append(os.path.basename(path)))
        self.target_model = types.SimpleNamespace(load_weights=lambda path: self.loads.append(os.path.basename(path)))


def main():
    mg.install_stubs()
    sys.path.insert(0, mg.REF)
    import Environment
    import Sim_Config
    import BS_brain
    fx = {'episodes': 10, 'steps': 5, 'epsilon': 0.3, 'trials': 2}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)                       # the reference creates its 'cwd\\Train-Result-...\\' folder
        try:
            for opt_flag, names, tag in ((True, NAMES_OPT, 'opt/'), (False, NAMES_RA, 'ra/')):
                seed = 777
                random.seed(seed)
                np.random.seed(seed)
                cfg = Sim_Config.RL_Config()
                cfg.set_train_value(16, 0.5, 32, 1, 0.1)
                env = mg.make_env(Environment)
                BS_brain.BS = FakeBSWithFiles
                BS_brain.Memory.samples = []
                agent = BS_brain.Agent(env.n_Veh, env.n_RB, env.n_Neighbor, cfg.Num_Feedback, env, cfg)
                out = agent.evaluate_training_diff_trials(fx['episodes'], fx['steps'], opt_flag, fx['epsilon'], fx['trials'])
                assert len(out) == len(names)
                for name, v in zip(names, out):
                    fx[tag + name] = np.asarray(v)
                fx[tag + 'loads'] = np.array(agent.brain.loads)
        finally:
            os.chdir(cwd)
    fx['seed'] = 777
    np.savez_compressed(os.path.join(HERE, 'golden_evaltrials_n4.npz'), **fx)
    print('wrote golden_evaltrials_n4.npz', {k: np.asarray(v).shape for k, v in fx.items()})


if __name__ == '__main__':
    main()
