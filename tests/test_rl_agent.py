"""CPU: the DQN agent glue (v2xgnn.rl.Agent / Memory) against the replay memory and the exact fit payload the
reference Agent produced on the reference simulator with a recording fake brain (tests/golden/make_golden.py,
section 1).  The fake brain here draws its "Q values" from the same private RandomState(77) stream."""
import os
import random
import types

import numpy as np
import pytest

from v2xgnn.rl import Agent, Memory, RL_Config
from test_rl_env import make_env
from util import GOLDEN


class RecordingBrain(object):
    def __init__(self, num_d2d, input_node_info, input_edge_info, num_d2d_feedback, num_d2d_neighbor, num_ch):
        self.num_D2D, self.num_Neighbor, self.num_CH = num_d2d, num_d2d_neighbor, num_ch
        self.num_Feedback = num_d2d_feedback
        self.num_One_Node_Input = ((input_node_info - 1) * num_ch + 1) * num_d2d_neighbor
        self.num_One_Edge_Input = input_edge_info * num_ch
        self.num_One_D2D_Input = self.num_One_Node_Input + self.num_One_Edge_Input
        self.num_D2D_Input = num_d2d * self.num_One_D2D_Input + num_d2d ** 2
        self.prng = np.random.RandomState(77)
        self.predicts, self.fits = [], []

    def predict(self, data, target=False):
        B = data['D1_Node_Input'].shape[0]
        out = [self.prng.normal(2.0, 1.0, size=(B, self.num_CH)).astype(np.float32) for _ in range(self.num_D2D)]
        self.predicts.append(({k: v.copy() for k, v in data.items()}, bool(target), [o.copy() for o in out]))
        return out

    def train_dnn(self, x, y, batch_size):
        self.fits.append((x, y, batch_size))
        return types.SimpleNamespace(history={('D%d_Decide_Output_loss' % (k + 1)): [0.0] for k in range(self.num_D2D)})


def _same(a, b):
    return np.allclose(a, b, rtol=1e-10, atol=1e-10)


def test_agent_reproduces_reference_memory_and_fit_payload():
    g = np.load(os.path.join(GOLDEN, 'golden_agent_n4.npz'))
    seed = int(g['seed'])
    random.seed(seed)
    np.random.seed(seed)
    cfg = RL_Config()
    cfg.set_train_value(16, float(g['gamma']), int(g['batch_size']), 1, 0.1)
    env = make_env()
    brain = RecordingBrain(env.n_Veh, 3, 1, cfg.Num_Feedback, env.n_Neighbor, env.n_RB)
    agent = Agent(env.n_Veh, env.n_RB, env.n_Neighbor, cfg.Num_Feedback, env, cfg, brain=brain)
    agent.num_Episodes, agent.num_Train_Step, agent.num_transition = 10, 20, 50
    assert np.array_equal([v.destinations[0] for v in env.vehicles], g['destinations'])
    assert agent.num_States == 4 * 13 + 16 and agent.num_Actions == 4

    init = agent.generate_d2d_initial_states()
    for k in init:
        assert init[k].shape == g['init/' + k].shape and _same(init[k], g['init/' + k]), k

    agent.num_step = 10 ** 9
    rewards = agent.generate_d2d_transition(24)
    assert _same(rewards, g['rollout_rewards'])
    mem = agent.memory.samples
    assert len(mem) == 24
    assert _same(np.stack([s[0][0] for s in mem]), g['mem_states'])
    assert np.array_equal(np.stack([s[1][0] for s in mem]), g['mem_actions'])
    assert _same(np.array([s[2] for s in mem]), g['mem_rewards'])
    assert _same(np.stack([s[3][0] for s in mem]), g['mem_states_next'])

    n_roll = len(brain.predicts)
    _, q_mean, q_max_mean, _, _ = agent.replay()
    (s, t0, p), (s_, t1, p_) = brain.predicts[n_roll:n_roll + 2]
    assert (t0, t1) == (False, True)
    for k in s:
        assert _same(s[k], g['replay_s/' + k]), k
        assert _same(s_[k], g['replay_s_next/' + k]), k
    for i in range(4):
        assert np.array_equal(p[i], g['replay_p/%d' % i]) and np.array_equal(p_[i], g['replay_p_next/%d' % i])
    fit_x, fit_y, bs = brain.fits[0]
    assert bs == int(g['batch_size'])
    for k in fit_x:
        assert fit_x[k].shape == g['fit_x/' + k].shape and _same(fit_x[k], g['fit_x/' + k]), k
    for k in fit_y:
        assert fit_y[k].dtype == g['fit_y/' + k].dtype
        # The fixture was captured from the reference code under numpy 2.2.6 (NEP 50: GAMMA * float32 stays float32); the
        # agent evaluates the target like the reference's pinned numpy-1.x stack (float64 product): <= 1 fp32 ulp apart.
        got, ref = np.asarray(fit_y[k], np.float64), np.asarray(g['fit_y/' + k], np.float64)
        assert np.all(np.abs(got - ref) <= np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)), k
    assert np.allclose(q_mean, g['q_mean'], rtol=2e-7, atol=0) and np.allclose(q_max_mean, g['q_max_mean'], rtol=2e-7, atol=0)


def test_memory_capacity_and_sampling_branches():
    m = Memory(5)
    for i in range(8):
        m.add([i])
    assert [s[0] for s in m.samples] == [3, 4, 5, 6, 7]
    np.random.seed(0)
    assert sorted(s[0] for s in m.sample(5)) == [3, 4, 5, 6, 7]          # without replacement
    assert len(m.sample(9)) == 9                                         # with replacement
    assert Memory(5).samples == []                                       # storage is per instance
    # the vectorised with-replacement draw is the reference's scalar loop (BS_brain.py:266-269), same RNG consumption
    np.random.seed(9)
    ref = [np.random.randint(0, len(m.samples)) for _ in range(40)]
    after = np.random.random()
    np.random.seed(9)
    assert list(m.sample_indices(40)) == ref and np.random.random() == after


def test_epsilon_schedule_and_random_actions():
    random.seed(3)
    np.random.seed(3)
    cfg = RL_Config()
    env = make_env()
    brain = RecordingBrain(env.n_Veh, 3, 1, 16, env.n_Neighbor, env.n_RB)
    agent = Agent(env.n_Veh, env.n_RB, env.n_Neighbor, 16, env, cfg, brain=brain)
    agent.num_Episodes, agent.num_Train_Step, agent.num_transition = 2, 5, 50
    state = agent.observe()
    a = agent.select_action_while_training(state)
    assert agent.epsilon == 1 and a.shape == (4, 1) and a.dtype.kind == 'i' and len(brain.predicts) == 0
    agent.num_step = 200                                                  # half way through the 0.8 * 500 decay steps
    agent.select_action_while_training(state)
    assert abs(agent.epsilon - (1 - 0.99 * 0.5)) < 1e-12
    agent.num_step = 400
    agent.select_action_while_training(state)
    assert agent.epsilon == 0.01


def test_test_run_matches_reference_evaluation_loop():
    """Greedy / random-action / brute-force-optimal evaluation (BS_brain.py:986-1162) against the reference's own run
    (tests/golden/make_golden_testrun.py: same simulator seed, same fake-brain stream)."""
    g = np.load(os.path.join(GOLDEN, 'golden_testrun_n4.npz'))
    seed = int(g['seed'])
    random.seed(seed)
    np.random.seed(seed)
    cfg = RL_Config()
    cfg.set_train_value(16, 0.5, 32, 1, 0.1)
    env = make_env()
    brain = RecordingBrain(env.n_Veh, 3, 1, cfg.Num_Feedback, env.n_Neighbor, env.n_RB)
    agent = Agent(env.n_Veh, env.n_RB, env.n_Neighbor, cfg.Num_Feedback, env, cfg, brain=brain)
    out = agent.test_run(int(g['episodes']), int(g['steps']), True)
    names = ['Expect_Return', 'Reward', 'Per_V2V_Rate', 'Per_V2I_Rate', 'Per_V2B_Interference']
    assert len(out) == 15
    for i, name in enumerate([p + n for p in ('', 'RA_', 'Opt_') for n in names]):
        assert out[i].shape == g[name].shape, name
        assert np.allclose(out[i], g[name], rtol=1e-9, atol=1e-12), name
    assert len(agent.test_run(1, 2, False)) == 10


def test_observation_at_twenty_links_matches_reference_get_state():
    """Agent.observe() (all links at once) against the reference's per-link get_state at 20 links
    (tests/golden/make_golden_testrun.py), incl. the adjacency rule of BS_brain.py:441-445."""
    g = np.load(os.path.join(GOLDEN, 'golden_observe_n20.npz'))
    random.seed(2020)
    np.random.seed(2020)
    cfg = RL_Config()
    cfg.set_train_value(16, 0.5, 32, 1, 0.1)
    env = make_env()
    env.new_random_game(20)
    brain = RecordingBrain(20, 3, 1, 16, env.n_Neighbor, env.n_RB)
    agent = Agent(20, env.n_RB, env.n_Neighbor, 16, env, cfg, brain=brain)
    for step in range(3):
        state, adj = agent.observe()
        assert np.array_equal([v.destinations[0] for v in env.vehicles], g['dest'][step])
        assert np.allclose(state[:, 0:4], g['v2v'][step], rtol=1e-10, atol=1e-12)
        assert np.allclose(state[:, 4:8], g['v2i'][step], rtol=1e-10, atol=1e-12)
        assert np.all(state[:, 8] == env.V2V_power_dB_List[env.fixed_v2v_power_index])
        assert np.allclose(state[:, 9:13], g['edge'][step], rtol=1e-10, atol=1e-12)
        for k in range(20):                                  # same numbers as the per-link accessor
            v2v, v2i, edge = agent.get_state([k, 0])
            assert np.allclose(state[k, 0:4], v2v) and np.allclose(state[k, 9:13], edge)
        want = np.ones((20, 20)) - np.eye(20)
        want[g['dest'][step], np.arange(20)] = 0
        assert np.array_equal(adj, want)
        a = np.random.randint(0, env.n_RB, size=(20, 1))
        env.compute_reward_with_channel_selection(a.copy())
        env.renew_positions()
        env.renew_channels_fastfading()
        env.Compute_Interference(a.copy())


def test_evaluate_training_diff_trials_matches_reference_loop(tmp_path):
    """Evaluation of the training process (BS_brain.py:1164-1451) against the reference's own run
    (tests/golden/make_golden_evaltrials.py: reference Agent + simulator, recording fake brain whose load_weights
    records file names): same checkpoints loaded in the same order, same re-seeding per trial, same returns."""
    g = np.load(os.path.join(GOLDEN, 'golden_evaltrials_n4.npz'))
    names_opt = ['Return', 'Reward', 'RA_Return', 'RA_Reward', 'Opt_Return', 'Opt_Reward', 'Opt_V2V', 'Opt_V2I', 'Opt_Interference']
    names_ra = ['Evaluated_Opt_Return', 'Return', 'Reward', 'RA_Return', 'RA_Reward']
    for opt_flag, names, tag in ((True, names_opt, 'opt/'), (False, names_ra, 'ra/')):
        seed = int(g['seed'])
        random.seed(seed)
        np.random.seed(seed)
        cfg = RL_Config()
        cfg.set_train_value(16, 0.5, 32, 1, 0.1)
        env = make_env()
        brain = RecordingBrain(env.n_Veh, 3, 1, cfg.Num_Feedback, env.n_Neighbor, env.n_RB)
        loads = []
        brain.model = types.SimpleNamespace(load_weights=lambda p: loads.append(os.path.basename(p)))
        brain.target_model = types.SimpleNamespace(load_weights=lambda p: loads.append(os.path.basename(p)))
        agent = Agent(env.n_Veh, env.n_RB, env.n_Neighbor, cfg.Num_Feedback, env, cfg, brain=brain)
        out = agent.evaluate_training_diff_trials(int(g['episodes']), int(g['steps']), opt_flag, float(g['epsilon']),
                                                  int(g['trials']), model_dir=str(tmp_path))
        assert len(out) == len(names)
        for o, name in zip(out, names):
            assert o.shape == g[tag + name].shape, name
            assert np.allclose(o, g[tag + name], rtol=1e-9, atol=1e-12), name
        # the reference builds 'cwd\\folder\\name' (Windows separators): compare the file names
        assert loads == [s.split('\\')[-1] for s in g[tag + 'loads']]
    assert os.path.basename(agent.checkpoint_dir('/x')) == 'Train-Result-RealFB-16-Batch-32-Gamma-0.5-V2Iweight-0.1'


def test_random_channels_consume_the_numpy_stream_like_the_reference_loop():
    """agent._random_channels: one randint call for all links == the reference's per-link np.random.choice(range(C), nn)
    loop, value for value, and it leaves the process-wide stream in the same state."""
    from v2xgnn.rl.agent import _random_channels
    for C in (3, 4, 5, 7):
        for nn in (1, 2):
            for n in (4, 20):
                np.random.seed(5 + C)
                np.random.random()
                ref = np.zeros((n, nn))
                for k in range(n):
                    ref[k, :] = np.random.choice(range(0, C), nn)
                after_ref = np.random.random()
                np.random.seed(5 + C)
                np.random.random()
                got = _random_channels(n, nn, C)
                assert np.array_equal(ref.astype(int), got) and got.dtype.kind == 'i'
                assert np.random.random() == after_ref


def test_more_than_31_links_keep_the_host_replay_memory_under_auto():
    """rl/replay.DeviceReplay stores a transition's adjacency as one int32 source mask per link; device_replay='auto' (the
    default on a GPU brain) must therefore fall back to the reference's host-side Memory at 32 links instead of failing at
    the first stored transition, and an explicit device_replay=True must say so at construction."""
    import pytest
    random.seed(3)
    np.random.seed(3)
    cfg = RL_Config()
    cfg.set_train_value(16, 0.5, 8, 1, 0.1)
    env = make_env()
    env.new_random_game(32)
    brain = RecordingBrain(32, 3, 1, 16, 1, 4)
    brain.model = types.SimpleNamespace(engine=types.SimpleNamespace(_h=1, device=0))      # looks like the gfx950 engine
    agent = Agent(32, env.n_RB, env.n_Neighbor, 16, env, cfg, brain=brain)
    assert agent.device_replay is None
    agent.generate_d2d_transition(3)                                  # stores through the host Memory
    assert len(agent.memory.samples) == 3
    with pytest.raises(ValueError, match="at most 31 links"):
        Agent(32, env.n_RB, env.n_Neighbor, 16, env, cfg, brain=brain, device_replay=True)


def test_native_sampler_is_numpys_choice_without_replacement():
    """Memory.sample_indices past 16,384 stored transitions: the library's restatement of np.random.choice(n, k, replace=False)
    (= permutation(n)[:k]: Fisher-Yates from the top with random_interval's masked rejection) on the process-wide generator --
    the same indices, the same dtype, and the generator in the same state afterwards, for sizes around powers of two and at
    the reference's capacity."""
    from v2xgnn.rl import native_sim
    from v2xgnn.rl.agent import Memory
    if not native_sim.available():
        pytest.skip("libv2xsim.so not built")
    for n, k in ((1, 1), (2, 1), (3, 2), (100, 7), (1025, 1025), (16384, 4096), (65536, 3), (65537, 4096), (200000, 4096), (1000000, 512)):
        np.random.seed(n + k)
        np.random.random(3); np.random.normal()                 # (a cached Gaussian in the state must survive the round trip)
        a, ra, ga = np.random.choice(n, k, replace=False), np.random.random(2), np.random.normal(size=3)
        np.random.seed(n + k)
        np.random.random(3); np.random.normal()
        b, rb, gb = native_sim.np_choice_noreplace(n, k), np.random.random(2), np.random.normal(size=3)
        assert a.dtype == b.dtype and np.array_equal(a, b) and np.array_equal(ra, rb) and np.array_equal(ga, gb), (n, k)
    mem = Memory(10 ** 6)
    mem.samples = [None] * 50000
    np.random.seed(5)
    want, want2 = np.random.choice(50000, 4096, replace=False), np.random.choice(70000, 4096, replace=False)
    np.random.seed(5)
    assert np.array_equal(mem.sample_indices(4096), want) and np.array_equal(mem.sample_indices(4096, 70000), want2)


def test_choice_ahead_notices_a_draw_in_its_window(monkeypatch):
    """ChoiceAhead borrows np.random's state between its start and result().  A draw from the process-wide
    generator in between must not be rolled back: result() redoes the draw from the CURRENT state (what the reference's
    np.random.choice at that point would draw), or raises under V2X_RL_STRICT_RNG=1.  Untouched state: the helper's result."""
    from v2xgnn.rl import native_sim
    if not native_sim.available():
        pytest.skip("libv2xsim.so not built")
    n, k = 30000, 512
    np.random.seed(77)
    want = np.random.choice(n, k, replace=False)
    after = np.random.random(3)
    np.random.seed(77)
    ca = native_sim.ChoiceAhead(n, k)
    assert np.array_equal(ca.result(), want) and np.array_equal(np.random.random(3), after)
    # a draw inside the window
    np.random.seed(77)
    stray_want = np.random.random(2)
    want2 = np.random.choice(n, k, replace=False)
    after2 = np.random.random(3)
    np.random.seed(77)
    n_fb = native_sim.ChoiceAhead.fallbacks
    ca = native_sim.ChoiceAhead(n, k)
    stray = np.random.random(2)
    got = ca.result()
    assert np.array_equal(stray, stray_want) and np.array_equal(got, want2) and np.array_equal(np.random.random(3), after2)
    assert native_sim.ChoiceAhead.fallbacks == n_fb + 1
    monkeypatch.setenv("V2X_RL_STRICT_RNG", "1")
    np.random.seed(77)
    ca = native_sim.ChoiceAhead(n, k)
    np.random.normal()                                            # (leaves a cached gauss value: position may be unchanged on a refill boundary)
    with pytest.raises(RuntimeError, match="np.random was used"):
        ca.result()


def test_native_policy_draws_are_the_python_loops_draws():
    """v2xsim_np_policy_draws: the epsilon draws and the exploring simulators' random actions of one batched iteration on numpy's
    process-wide generator -- the same draws in the same order, the same actions and greedy set, the same generator state afterwards
    as the loop of np.random.random() / np.random.randint(0, C, (n, 1)) (BS_brain.py:308-352 per simulator), incl. action counts
    that are not a power of two (randint's masked rejection) and a cached Gaussian in the state."""
    from v2xgnn.rl import native_sim
    if not native_sim.available():
        pytest.skip("libv2xsim.so not built")
    for E, n, C, step0 in ((50, 20, 4, 0), (7, 4, 4, 123), (33, 31, 5, 40), (1, 3, 3, 9), (50, 20, 1, 5)):
        eps_steps = 0.8 * 5 * 20 * 50
        per_step = (1.0 - 0.01) / eps_steps
        np.random.seed(E * 1000 + n)
        np.random.normal()
        want_a, want_g = np.zeros((E, n, 1), np.int64), []
        for e in range(E):
            step_no = step0 + e * (1 if E < 50 else 90)
            eps = 1.0 - per_step * step_no if step_no < eps_steps else 0.01
            if np.random.random() < eps:
                want_a[e] = np.random.randint(0, C, size=(n, 1))
            else:
                want_g.append(e)
        after = (np.random.random(2), np.random.normal(size=2))
        if E == 50:
            continue                                             # (the library takes consecutive step numbers: compared for the others)
        np.random.seed(E * 1000 + n)
        np.random.normal()
        a, g, eps_last = native_sim.np_policy_draws(E, n, C, 1.0, 0.01, per_step, eps_steps, step0)
        assert np.array_equal(a, want_a) and list(g) == want_g and eps_last == eps
        assert np.array_equal(np.random.random(2), after[0]) and np.array_equal(np.random.normal(size=2), after[1])
    # a long mixed run: consecutive steps across the end of the schedule
    np.random.seed(4)
    eps_steps, per_step = 300.0, 0.99 / 300.0
    want = []
    for e in range(400):
        eps = 1.0 - per_step * e if e < eps_steps else 0.01
        want.append(np.random.randint(0, 4, size=(6, 1)) if np.random.random() < eps else None)
    tail = np.random.random(3)
    np.random.seed(4)
    a, g, _ = native_sim.np_policy_draws(400, 6, 4, 1.0, 0.01, per_step, eps_steps, 0)
    assert list(g) == [e for e, w in enumerate(want) if w is None] and 0 < len(g) < 400
    assert all(np.array_equal(a[e], w) for e, w in enumerate(want) if w is not None)
    assert np.array_equal(np.random.random(3), tail)
