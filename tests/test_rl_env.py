"""CPU: the simulator counterpart (v2xgnn.rl.Environ) against seeded trajectories captured from the reference
Environment.py in the build container (tests/golden/make_golden.py, sections 3 and 4)."""
import os
import random

import numpy as np
import pytest

from v2xgnn.rl import Environ
from util import GOLDEN


def make_env():
    # lane constants of the reference drivers (RL_Train_main.py:82-88)
    up = [3.5 / 2, 3.5 / 2 + 3.5, 250 + 3.5 / 2, 250 + 3.5 + 3.5 / 2, 500 + 3.5 / 2, 500 + 3.5 + 3.5 / 2]
    down = [250 - 3.5 - 3.5 / 2, 250 - 3.5 / 2, 500 - 3.5 - 3.5 / 2, 500 - 3.5 / 2, 750 - 3.5 - 3.5 / 2, 750 - 3.5 / 2]
    left = [3.5 / 2, 3.5 / 2 + 3.5, 433 + 3.5 / 2, 433 + 3.5 + 3.5 / 2, 866 + 3.5 / 2, 866 + 3.5 + 3.5 / 2]
    right = [433 - 3.5 - 3.5 / 2, 433 - 3.5 / 2, 866 - 3.5 - 3.5 / 2, 866 - 3.5 / 2, 1299 - 3.5 - 3.5 / 2, 1299 - 3.5 / 2]
    env = Environ(down, up, left, right, 750, 1299)
    env.new_random_game(env.n_Veh)
    return env


@pytest.mark.parametrize("n_veh", [4, 20])
def test_seeded_trajectory_matches_reference(n_veh):
    g = np.load(os.path.join(GOLDEN, 'golden_env_n%d.npz' % n_veh))
    random.seed(2020 + n_veh)
    np.random.seed(2020 + n_veh)
    env = make_env()
    if n_veh != env.n_Veh:
        env.new_random_game(n_veh)
    assert np.array_equal(np.array([v.position for v in env.vehicles], float), g['init_pos'])
    assert [v.direction for v in env.vehicles] == list(g['init_dir'])
    assert np.array_equal([v.velocity for v in env.vehicles], g['velocity'])
    assert np.array_equal([v.destinations[0] for v in env.vehicles], g['init_dest'])
    assert np.allclose(env.V2V_channels_with_fastfading, g['init_v2v'], rtol=1e-11, atol=1e-9)
    assert np.allclose(env.V2I_channels_with_fastfading, g['init_v2i'], rtol=1e-11, atol=1e-9)
    for t in range(int(g['steps'])):
        a = np.random.randint(0, env.n_RB, size=(n_veh, 1))
        assert np.array_equal(a, g['actions'][t])
        v2v_rate, v2i_rate, interference = env.compute_reward_with_channel_selection(a.copy())
        assert np.allclose(v2v_rate, g['v2v_rate'][t], rtol=1e-9, atol=1e-12), t
        assert np.allclose(v2i_rate, g['v2i_rate'][t], rtol=1e-9, atol=1e-12), t
        assert np.allclose(interference, g['interference'][t], rtol=1e-9, atol=0), t
        env.renew_positions()
        env.renew_channels_fastfading()
        env.Compute_Interference(a.copy())
        assert np.array_equal(np.array([v.position for v in env.vehicles], float), g['pos'][t]), t
        assert [v.direction for v in env.vehicles] == list(g['dirs'][t])
        assert np.allclose(env.V2V_channels_with_fastfading, g['v2v'][t], rtol=1e-11, atol=1e-9), t
        assert np.allclose(env.V2I_channels_with_fastfading, g['v2i'][t], rtol=1e-11, atol=1e-9), t
        assert np.allclose(env.V2V_Interference_all, g['v2v_interference_all'][t], rtol=1e-10, atol=1e-9), t


def test_turns_and_exits_match_reference():
    """Mobility at crossings and map borders incl. the RNG consumption of the 0.4-probability turns."""
    g = np.load(os.path.join(GOLDEN, 'golden_env_cross.npz'))
    random.seed(77)
    env = make_env()
    assert np.array_equal([v.velocity for v in env.vehicles], g['velocity'])
    per = int(g['per'])
    k = 0
    changed = 0
    for rd in range(g['place_pos'].shape[0]):
        for i, v in enumerate(env.vehicles):
            v.position, v.direction = list(g['place_pos'][rd, i]), str(g['place_dir'][rd, i])
        for _ in range(per):
            env.renew_positions()
            assert np.array_equal(np.array([v.position for v in env.vehicles], float), g['out_pos'][k]), (rd, k)
            assert [v.direction for v in env.vehicles] == list(g['out_dir'][k]), (rd, k)
            k += 1
        changed += sum(v.direction != str(g['place_dir'][rd, i]) for i, v in enumerate(env.vehicles))
    assert changed > 100          # the fixture really exercises turns and exits


def test_hundred_link_simulator_matches_reference():
    """BASELINE config 4's graph size: positions / destinations bit-exact, channels (strided sample + column sums) and
    rates to 1e-9 against the reference simulator (tests/golden/make_golden_testrun.py)."""
    g = np.load(os.path.join(GOLDEN, 'golden_env_n100.npz'))
    random.seed(2100)
    np.random.seed(2100)
    env = make_env()
    env.new_random_game(100)
    for step in range(3):
        a = np.random.randint(0, env.n_RB, size=(100, 1))
        r_v2v, r_v2i, _ = env.compute_reward_with_channel_selection(a.copy())
        env.renew_positions()
        env.renew_channels_fastfading()
        env.Compute_Interference(a.copy())
        assert np.array_equal(np.array([v.position for v in env.vehicles], float), g['pos'][step])
        assert np.array_equal([v.destinations[0] for v in env.vehicles], g['dest'][step])
        assert np.allclose(env.V2V_channels_with_fastfading[::7, ::11, :], g['v2v_sample'][step], rtol=1e-11, atol=1e-9)
        assert np.allclose(env.V2V_channels_with_fastfading.sum(axis=(0, 1)), g['v2v_sum'][step], rtol=1e-11, atol=1e-6)
        assert np.allclose(env.V2I_channels_with_fastfading, g['v2i'][step], rtol=1e-11, atol=1e-9)
        assert np.allclose(r_v2v, g['v2v_rate'][step], rtol=1e-9, atol=1e-12)
        assert np.allclose(r_v2i, g['v2i_rate'][step], rtol=1e-9, atol=1e-12)
