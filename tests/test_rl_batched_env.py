"""CPU: the batched simulator (v2xgnn.rl.BatchedEnviron).  One environment on the process-wide stdlib
generator reproduces the trajectories captured from the reference Environment.py (tests/golden/golden_env_*.npz) like the
single simulator does; environment e of a seeded batch IS the single simulator run after random.seed(seeds[e])."""
import os
import random

import numpy as np
import pytest

from v2xgnn.rl import Environ, BatchedEnviron
from v2xgnn.rl.mtstream import MTStream
from test_rl_env import make_env
from util import GOLDEN

UP = [3.5 / 2, 3.5 / 2 + 3.5, 250 + 3.5 / 2, 250 + 3.5 + 3.5 / 2, 500 + 3.5 / 2, 500 + 3.5 + 3.5 / 2]
DOWN = [250 - 3.5 - 3.5 / 2, 250 - 3.5 / 2, 500 - 3.5 - 3.5 / 2, 500 - 3.5 / 2, 750 - 3.5 - 3.5 / 2, 750 - 3.5 / 2]
LEFT = [3.5 / 2, 3.5 / 2 + 3.5, 433 + 3.5 / 2, 433 + 3.5 + 3.5 / 2, 866 + 3.5 / 2, 866 + 3.5 + 3.5 / 2]
RIGHT = [433 - 3.5 - 3.5 / 2, 433 - 3.5 / 2, 866 - 3.5 - 3.5 / 2, 866 - 3.5 / 2, 1299 - 3.5 - 3.5 / 2, 1299 - 3.5 / 2]


def make_batched(n_envs=1, seeds=None):
    env = BatchedEnviron(DOWN, UP, LEFT, RIGHT, 750, 1299, n_envs=n_envs, seeds=seeds)
    env.new_random_game(env.n_Veh)
    return env


def test_mtstream_is_the_stdlib_generator():
    for seed in (0, 1001, 2 ** 40 + 5):
        r, s = random.Random(seed), MTStream(seed)
        for _ in range(3):
            assert r.random() == s.random() and r.uniform(0, 1) == s.uniform(0, 1)
            assert [r.randint(10, 15) for _ in range(20)] == [s.randint(10, 15) for _ in range(20)]
            assert [r.randrange(0, 6) for _ in range(20)] == [s.randrange(0, 6) for _ in range(20)]
            assert [r.randint(0, 1299) for _ in range(20)] == [s.randint(0, 1299) for _ in range(20)]
            assert np.array_equal([r.gauss(0, 3) for _ in range(7)], s.gauss_array((7,), 3))          # odd: caches a value
            assert np.array_equal([r.gauss(0, 1) for _ in range(8)], s.gauss_array((2, 4), 1).ravel())
            assert r.sample(list(range(17)), 1) == s.sample(list(range(17)), 1)                         # pool method
            assert r.sample(list(range(97)), 1) == s.sample(list(range(97)), 1)                         # rejection set
            assert r.sample(list(range(40)), 7) == s.sample(list(range(40)), 7)
        r2 = random.Random()
        s.return_stdlib(r2)
        assert r2.random() == r.random() and r2.gauss(0, 1) == r.gauss(0, 1)


@pytest.mark.parametrize("n_veh", [4, 20])
def test_one_batched_environment_matches_reference_trajectory(n_veh):
    g = np.load(os.path.join(GOLDEN, 'golden_env_n%d.npz' % n_veh))
    random.seed(2020 + n_veh)
    np.random.seed(2020 + n_veh)
    env = make_batched()
    if n_veh != env.n_Veh:
        env.new_random_game(n_veh)
    assert np.array_equal(env.pos[0], g['init_pos']) and env.directions(0) == list(g['init_dir'])
    assert np.array_equal(env.vel[0], g['velocity']) and np.array_equal(env.dest[0], g['init_dest'])
    assert np.allclose(env.V2V_channels_with_fastfading[0], g['init_v2v'], rtol=1e-11, atol=1e-9)
    assert np.allclose(env.V2I_channels_with_fastfading[0], g['init_v2i'], rtol=1e-11, atol=1e-9)
    for t in range(int(g['steps'])):
        a = np.random.randint(0, env.n_RB, size=(n_veh, 1))
        v2v_rate, v2i_rate, interference = env.act(a[None])
        assert np.allclose(v2v_rate[0], g['v2v_rate'][t], rtol=1e-9, atol=1e-12), t
        assert np.allclose(v2i_rate[0], g['v2i_rate'][t], rtol=1e-9, atol=1e-12), t
        assert np.allclose(interference[0], g['interference'][t], rtol=1e-9, atol=0), t
        assert np.array_equal(env.pos[0], g['pos'][t]), t
        assert env.directions(0) == list(g['dirs'][t])
        assert np.allclose(env.V2V_channels_with_fastfading[0], g['v2v'][t], rtol=1e-11, atol=1e-9), t
        assert np.allclose(env.V2I_channels_with_fastfading[0], g['v2i'][t], rtol=1e-11, atol=1e-9), t
        assert np.allclose(env.V2V_Interference_all[0], g['v2v_interference_all'][t], rtol=1e-10, atol=1e-9), t


def _agent_observe(env):
    """Agent.observe on a single simulator (v2xgnn.rl.Agent, same arithmetic as BS_brain.py:389-467)."""
    from v2xgnn.rl import Agent, RL_Config
    import types
    brain = types.SimpleNamespace(num_D2D_Input=0, num_One_D2D_Input=13, num_One_Node_Input=9, num_Feedback=16)
    agent = Agent(env.n_Veh, env.n_RB, env.n_Neighbor, 16, env, RL_Config(), brain=brain, device_replay=False)
    return agent.observe()


@pytest.mark.parametrize("n_veh,steps", [(4, 300), (20, 60)])
def test_every_environment_of_a_batch_is_the_seeded_single_simulator(n_veh, steps):
    seeds = [11, 12, 1001]
    batch = make_batched(n_envs=3, seeds=seeds)
    batch.new_random_game(n_veh)
    singles = []
    for s in seeds:
        random.seed(s)
        env = make_env()
        env.new_random_game(n_veh)
        singles.append((env, random.getstate()))
    rng = np.random.default_rng(5)
    turned = 0
    for t in range(steps):
        a = rng.integers(0, 4, size=(3, n_veh, 1))
        sb, ab = batch.observe()
        rb = batch.act(a)
        for e, (env, st) in enumerate(singles):
            random.setstate(st)                               # each single simulator continues ITS stdlib stream
            s1, a1 = _agent_observe(env)
            assert np.allclose(sb[e], s1, rtol=1e-12, atol=1e-12) and np.array_equal(ab[e], a1)
            v2v, v2i, intf = env.compute_reward_with_channel_selection(a[e].copy())
            before = [v.direction for v in env.vehicles]
            env.renew_positions()
            env.renew_channels_fastfading()
            env.Compute_Interference(a[e].copy())
            turned += sum(b != v.direction for b, v in zip(before, env.vehicles))
            singles[e] = (env, random.getstate())
            assert np.array_equal(batch.pos[e], np.array([v.position for v in env.vehicles], float)), (t, e)
            assert batch.directions(e) == [v.direction for v in env.vehicles]
            assert np.array_equal(batch.dest[e], [v.destinations[0] for v in env.vehicles])
            assert np.allclose(batch.V2V_channels_with_fastfading[e], env.V2V_channels_with_fastfading, rtol=1e-13, atol=1e-11)
            assert np.allclose(rb[0][e], v2v, rtol=1e-10, atol=1e-13) and np.allclose(rb[1][e], v2i, rtol=1e-10, atol=1e-13)
            assert np.allclose(rb[2][e], intf, rtol=1e-10, atol=0)
            assert np.allclose(batch.V2V_Interference_all[e], env.V2V_Interference_all, rtol=1e-11, atol=1e-10)
    if n_veh == 4:
        assert turned > 0                                     # the run really crossed lanes (RNG draws in renew_positions)
    # the streams end where the single simulators' stdlib generators end
    for e, (env, st) in enumerate(singles):
        r = random.Random()
        r.setstate(st)
        assert batch.streams[e].random() == r.random()


def _rollout_and_replay(env, seed):
    """40 transitions + one replay with the recording fake brain: the memory and the exact fit payload"""
    from v2xgnn.rl import Agent, RL_Config
    from test_rl_agent import RecordingBrain
    np.random.seed(seed)
    cfg = RL_Config()
    cfg.set_train_value(16, 0.5, 32, 1, 0.1)
    brain = RecordingBrain(4, 3, 1, 16, 1, 4)
    agent = Agent(4, 4, 1, 16, env, cfg, brain=brain, device_replay=False)
    agent.num_Episodes, agent.num_Train_Step, agent.num_transition = 1, 2, 40
    agent.num_step = 20                                        # mid-schedule: both random and greedy actions occur
    rewards = agent.generate_d2d_transition(40)
    agent.replay()
    mem = agent.memory.samples
    return rewards, mem, brain.fits[0], len(brain.predicts), agent.num_step


def test_agent_on_one_batched_environment_equals_agent_on_the_single_simulator():
    random.seed(31)
    r1, m1, f1, p1, n1 = _rollout_and_replay(make_env(), 8)
    random.seed(31)
    r2, m2, f2, p2, n2 = _rollout_and_replay(make_batched(), 8)
    assert np.allclose(r1, r2, rtol=1e-10, atol=0) and p1 == p2 and n1 == n2 == 60 and len(m1) == len(m2) == 40
    for a, b in zip(m1, m2):
        assert np.allclose(a[0], b[0], rtol=1e-11, atol=1e-12) and np.array_equal(a[1], b[1])
        assert np.isclose(a[2], b[2], rtol=1e-10) and np.allclose(a[3], b[3], rtol=1e-11, atol=1e-12)
    (x1, y1, _), (x2, y2, _) = f1, f2
    for k in x1:
        assert np.allclose(x1[k], x2[k], rtol=1e-11, atol=1e-12), k
    for k in y1:
        assert np.allclose(y1[k], y2[k], rtol=1e-6, atol=1e-7), k


def test_agent_training_loop_on_a_batch_of_environments():
    """Agent.train on 4 simulators stepped as arrays (engine injected: the oracle): 52 = ceil(50 / 4) x 4 transitions per
    train step, greedy environments share one forward pass, losses finite, the loop trains."""
    from v2xgnn import BS
    from v2xgnn.rl import Agent, RL_Config
    from oracle_engine import OracleEngine
    np.random.seed(3)
    cfg = RL_Config()
    cfg.set_train_value(16, 0.5, 32, 1, 0.1)
    env = make_batched(4, seeds=[1, 2, 3, 4])
    brain = BS(4, 3, 1, 16, 1, 4, seed=3, engine_factory=lambda spec: OracleEngine(spec))
    w0 = np.concatenate([a.ravel() for a in brain.model.get_weights()])
    agent = Agent(4, 4, 1, 16, env, cfg, brain=brain)
    loss, reward_step, _, q_mean, q_max, _, _ = agent.train(1, 3)
    assert agent.num_transition == 52 and agent.num_step == 156 and len(agent.memory.samples) == 156
    assert reward_step.shape == (1, 3, 52) and np.all(np.isfinite(reward_step)) and np.all(reward_step > 0)
    assert np.all(np.isfinite(loss)) and np.all(q_max >= q_mean - 1e-12)
    assert not np.allclose(w0, np.concatenate([a.ravel() for a in brain.model.get_weights()]))


def test_channel_update_threads_do_not_change_the_result():
    """workers > 1 splits the environments of a step over threads; environments are independent (own stream, own state),
    so 1, 3 and 8 threads give bit-identical simulators."""
    from v2xgnn.rl.batched_env import BatchedEnviron
    up = [3.5 / 2, 3.5 / 2 + 3.5, 250 + 3.5 / 2, 250 + 3.5 + 3.5 / 2, 500 + 3.5 / 2, 500 + 3.5 + 3.5 / 2]
    dn = [250 - 3.5 - 3.5 / 2, 250 - 3.5 / 2, 500 - 3.5 - 3.5 / 2, 500 - 3.5 / 2, 750 - 3.5 - 3.5 / 2, 750 - 3.5 / 2]
    le = [3.5 / 2, 3.5 / 2 + 3.5, 433 + 3.5 / 2, 433 + 3.5 + 3.5 / 2, 866 + 3.5 / 2, 866 + 3.5 + 3.5 / 2]
    ri = [433 - 3.5 - 3.5 / 2, 433 - 3.5 / 2, 866 - 3.5 - 3.5 / 2, 866 - 3.5 / 2, 1299 - 3.5 - 3.5 / 2, 1299 - 3.5 / 2]
    outs = []
    for w in (1, 3, 8):
        env = BatchedEnviron(dn, up, le, ri, 750, 1299, n_envs=11, seeds=[5 + 104729 * e for e in range(11)], workers=w)
        env.new_random_game(8)
        rng = np.random.default_rng(3)
        acc = []
        for _ in range(6):
            s, adj = env.observe(4)
            v2v, v2i, _ = env.act(rng.integers(0, 4, size=(11, 8, 1)))
            acc += [s.copy(), adj.copy(), v2v.copy(), v2i.copy(), env.pos.copy()]
        outs.append(acc)
        assert env.workers == min(w, 11)
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert np.array_equal(a, b)


def _lanes():
    up = [3.5 / 2, 3.5 / 2 + 3.5, 250 + 3.5 / 2, 250 + 3.5 + 3.5 / 2, 500 + 3.5 / 2, 500 + 3.5 + 3.5 / 2]
    dn = [250 - 3.5 - 3.5 / 2, 250 - 3.5 / 2, 500 - 3.5 - 3.5 / 2, 500 - 3.5 / 2, 750 - 3.5 - 3.5 / 2, 750 - 3.5 / 2]
    le = [3.5 / 2, 3.5 / 2 + 3.5, 433 + 3.5 / 2, 433 + 3.5 + 3.5 / 2, 866 + 3.5 / 2, 866 + 3.5 + 3.5 / 2]
    ri = [433 - 3.5 - 3.5 / 2, 433 - 3.5 / 2, 866 - 3.5 - 3.5 / 2, 866 - 3.5 / 2, 1299 - 3.5 - 3.5 / 2, 1299 - 3.5 / 2]
    return dn, up, le, ri


@pytest.mark.parametrize("n_links", [4, 20])
def test_native_step_equals_the_numpy_step(n_links):
    """csrc/v2xsim.c (C + OpenMP: MT19937 bulk draws, Box-Muller, shadowing, path loss, fast fading, rates, interference,
    observation) against the numpy expressions it restates: integer state (positions' lane logic, directions, receivers)
    and the random streams stay IDENTICAL, real-valued outputs agree to libm rounding (the shadowing recursion carries
    it along: 1e-10 after 150 steps is generous)."""
    from v2xgnn.rl import native_sim
    if not native_sim.available():
        pytest.skip("libv2xsim.so not built")
    E = 5
    envs = []
    for native in (False, True):
        env = BatchedEnviron(*_lanes(), 750, 1299, n_envs=E, seeds=[11 + 104729 * e for e in range(E)], workers=1, native=native)
        env.new_random_game(n_links)
        envs.append(env)
    a, b = envs
    assert not a.native and b.native
    rng = np.random.default_rng(1)
    for it in range(150):
        (sa, aa), (sb, ab) = a.observe(4), b.observe(4)
        act = rng.integers(0, 4, size=(E, n_links, 1))
        ra, rb = a.act(act), b.act(act)
        assert np.array_equal(a.pos, b.pos) and np.array_equal(a.dirs, b.dirs) and np.array_equal(aa, ab)
        for x, y in list(zip(ra, rb)) + [(sa, sb), (a.V2V_Interference_all, b.V2V_Interference_all),
                                         (a.V2I_Interference, b.V2I_Interference), (a.V2V_Interference, b.V2V_Interference)]:
            assert x.shape == y.shape
            assert np.allclose(x, y, rtol=1e-10, atol=0), (it, np.abs(x - y).max())
    for s0, s1 in zip(a.streams, b.streams):
        assert s0._export() == s1._export()                    # same MT19937 position after 150 x 3,360 (or 160) draws


def test_native_walk_with_turns_and_reentries_equals_the_numpy_walk():
    """renew_positions in C (v2xsim_positions / v2xsim_advance) against the numpy + Python walk: vehicles made fast enough to
    reach a crossing lane almost every step and to leave the map within the test -- positions, directions and the streams
    (one uniform per reached lane, in the reference's checking order) stay identical, bit for bit."""
    from v2xgnn.rl import native_sim
    if not native_sim.available():
        pytest.skip("libv2xsim.so not built")
    E, n = 6, 8
    envs = []
    for native in (False, True):
        env = BatchedEnviron(*_lanes(), 750, 1299, n_envs=E, seeds=[5 + 7919 * e for e in range(E)], workers=1, native=native)
        env.new_random_game(n)
        env.vel[:] = env.vel * 150.0                           # 15-22 m per step: a lane every few steps, the map edge in < 100
        envs.append(env)
    a, b = envs
    turns = exits = 0
    for it in range(400):
        d0, p0 = a.dirs.copy(), a.pos.copy()
        a.renew_positions()
        b.renew_positions()
        assert np.array_equal(a.pos, b.pos) and np.array_equal(a.dirs, b.dirs), it
        turns += int((a.dirs != d0).sum())
        exits += int((np.abs(a.pos - p0).max(axis=2) > 30).sum())
    assert turns > 200 and exits > 20, (turns, exits)
    for s0, s1 in zip(a.streams, b.streams):
        assert s0._export() == s1._export()
    # ... and as part of the one-call step (act): same walk, then the channel draws
    act = np.zeros((E, n, 1), int)
    for it in range(50):
        ra, rb = a.act(act), b.act(act)
        assert np.array_equal(a.pos, b.pos) and np.array_equal(a.dirs, b.dirs), it
        assert np.allclose(ra[0], rb[0], rtol=1e-9, atol=0)
    for s0, s1 in zip(a.streams, b.streams):
        assert s0._export() == s1._export()


def test_lookahead_is_invisible():
    """lookahead=True computes the next step on the library's worker thread while the caller is busy: every array the
    simulator exposes, every rate and every stream is bit for bit what the simulator without it produces -- through plain
    steps, episode resets, steps made by hand (renew_positions / renew_channels_fastfading) and direct draws from a stream, all
    of which find a started look-ahead and must drop it."""
    from v2xgnn.rl import native_sim
    if not native_sim.available():
        pytest.skip("libv2xsim.so not built")
    E, n = 7, 20
    envs = []
    for ahead in (False, True):
        env = BatchedEnviron(*_lanes(), 750, 1299, n_envs=E, seeds=[3 + 15485863 * e for e in range(E)], lookahead=ahead)
        env.new_random_game(n)
        envs.append(env)
    a, b = envs
    assert b.lookahead and not a.lookahead and a._one_call_step()
    rng = np.random.default_rng(9)
    names = ("pos", "dirs", "vel", "dest", "_v2i_shadow", "_v2v_shadow", "V2V_channels_abs", "V2I_channels_abs",
             "V2V_channels_with_fastfading", "V2I_channels_with_fastfading", "V2V_Interference_all", "_mt_keys", "_mt_pos")

    def same(tag):
        for k in names:
            assert np.array_equal(getattr(a, k), getattr(b, k)), (tag, k)
        for x, y in zip(a.observe(4) + a.observe_packed(4), b.observe(4) + b.observe_packed(4)):
            assert x.dtype == y.dtype and np.array_equal(x, y), tag
    for it in range(60):
        act = rng.integers(0, 4, size=(E, n, 1))
        for x, y in zip(a.act(act), b.act(act)):
            assert np.array_equal(x, y)
        assert a._ahead is None and b._ahead is not None      # the next step is already on its way
        same(it)
        if it % 13 == 5:
            a.new_random_game(n); b.new_random_game(n)
            assert b._ahead is None
            same("reset")
        if it % 17 == 3:
            a.renew_positions(); b.renew_positions()
            a.renew_channels_fastfading(); b.renew_channels_fastfading()
            same("by hand")
        if it % 19 == 7:
            assert a.streams[2].random() == b.streams[2].random() and b._ahead is None        # a bare draw drops it too
            for x, y in zip(a.act(act), b.act(act)):
                assert np.array_equal(x, y)
            with a._rng() as ra, b._rng() as rb:
                assert ra[1].random() == rb[1].random()
            assert b._ahead is None
    # the packed observation is the dense one: float32 rows, source masks, CSR sources by destination
    st, adj = b.observe(4)
    xe, mask, col, regular = b.observe_packed(4)
    assert np.array_equal(xe[:, :, :13], st.astype(np.float32)) and not xe[:, :, 13:].any()
    assert np.array_equal(mask, ((adj != 0).astype(np.int64) << np.arange(n)[None, :, None]).sum(axis=1))
    for e in range(E):
        assert regular[e] == bool(np.all((adj[e] != 0).sum(axis=0) == n - 2))
        if regular[e]:
            assert np.array_equal(col[e], np.nonzero(adj[e].T)[1])


def test_native_mt19937_is_the_stdlib_stream():
    from v2xgnn.rl import native_sim
    if not native_sim.available():
        pytest.skip("libv2xsim.so not built")
    keys, pos = np.empty((3, 624), np.uint32), np.zeros(3, np.int32)
    streams = [MTStream(s) for s in (0, 12345, 2 ** 40 + 7)]
    for e, s in enumerate(streams):
        s.random_array(100 * e + 1)                            # different positions inside the 624-word block
        s.attach(keys[e], pos, e)
    want = [random.Random(s) for s in (0, 12345, 2 ** 40 + 7)]
    for e, r in enumerate(want):
        for _ in range(100 * e + 1):
            r.random()
    got = native_sim.mt_uniforms(keys, pos, 1500)              # crosses two reloads of the state
    for e, r in enumerate(want):
        assert got[e].tolist() == [r.random() for _ in range(1500)]
        assert streams[e].random() == r.random()               # the numpy-side view continues where the C draws stopped
        assert streams[e].randint(0, 10 ** 6) == r.randint(0, 10 ** 6)
    assert native_sim.mt_uniforms(keys, pos, 2)[1].tolist() == [want[1].random(), want[1].random()]


def test_native_reset_draws_are_the_stdlib_draws():
    """v2xsim_reset_vehicles / v2xsim_sample_dest: add_new_vehicles_by_number's randrange / randint draws and renew_neighbor's
    random.sample(candidates, 1) of CPython, draw for draw, on the environments' own MT19937 states (several reloads of the
    624-word state inside one call; populations of 1, 5, 17 and 21 candidates)."""
    from v2xgnn.rl import native_sim
    if not native_sim.available():
        pytest.skip("libv2xsim.so not built")
    seeds = (0, 12345, 2 ** 40 + 7)
    lanes = ([1.0, 2.0, 3.0, 4.0, 5.0, 6.0], [11.0, 12.0, 13.0, 14.0, 15.0, 16.0], [21.0, 22.0, 23.0, 24.0, 25.0, 26.0],
             [31.0, 32.0, 33.0, 34.0, 35.0, 36.0])
    for n in (4, 20, 400):
        keys, pos = np.empty((3, 624), np.uint32), np.zeros(3, np.int32)
        streams = [MTStream(s) for s in seeds]
        for e, s in enumerate(streams):
            s.attach(keys[e], pos, e)
        xy, dirs, vel = native_sim.reset_vehicles(keys, pos, n, lanes, 750, 1299)
        for e, seed in enumerate(seeds):
            r = random.Random(seed)
            k = 0
            for _ in range(n // 4):
                ind = r.randrange(0, 6)
                for code, lx, ly in ((1, lanes[0][ind], None), (0, lanes[1][ind], None), (2, None, lanes[2][ind]), (3, None, lanes[3][ind])):
                    want = (lx, r.randint(0, 1299)) if ly is None else (r.randint(0, 750), ly)
                    assert tuple(xy[e, k]) == want and dirs[e, k] == code and vel[e, k] == r.randint(10, 15), (n, e, k)
                    k += 1
            assert streams[e].random() == r.random()           # both generators stand at the same draw
    for m in (1, 5, 17, 21):
        keys, pos = np.empty((3, 624), np.uint32), np.zeros(3, np.int32)
        streams = [MTStream(s) for s in seeds]
        for e, s in enumerate(streams):
            s.attach(keys[e], pos, e)
        cand = np.random.default_rng(m).integers(0, 1000, size=(3, 40, m))
        dest = native_sim.sample_dest(keys, pos, cand)
        for e, seed in enumerate(seeds):
            r = random.Random(seed)
            assert dest[e].tolist() == [r.sample(cand[e, i].tolist(), 1)[0] for i in range(40)], (m, e)
            assert streams[e].random() == r.random()
    with pytest.raises(ValueError):
        native_sim.sample_dest(keys, pos, np.zeros((3, 4, 22), np.int64))


def test_device_replay_add_many_stages_what_add_stages():
    """The batched rollout stores E transitions per step with one vectorised call; the staged records (packed features,
    CSR columns, source masks, regularity flag) must be those of E single adds -- incl. an irregular graph (a link that
    is its own receiver: in-degree n - 1)."""
    from v2xgnn.rl.replay import DeviceReplay
    rng = np.random.default_rng(5)
    n, K = 6, 7
    adj = np.ones((K, n, n)) - np.eye(n)[None]
    for k in range(K):
        dest = rng.integers(0, n - 1, size=n)
        dest = dest + (dest >= np.arange(n))
        adj[k, dest, np.arange(n)] = 0
    adj[3, :, 2] = 1 - np.eye(n)[2]                            # link 2 of transition 3: receiver == itself -> only the self edge is missing
    x, e = rng.normal(size=(K, n, 9)), rng.normal(size=(K, n, 4))
    x2, e2 = rng.normal(size=(K, n, 9)), rng.normal(size=(K, n, 4))
    act, rew = rng.integers(0, 4, size=(K, n)), rng.normal(size=K)
    one, many = DeviceReplay(64, n), DeviceReplay(64, n)
    for k in range(K):
        one.add(x[k], e[k], adj[k], act[k], rew[k], x2[k], e2[k])
    many.add_many(x, e, adj, act, rew, x2, e2)
    # (staged as blocks of transitions: K blocks of one against one block of K)
    assert len(one) == len(many) == K and len(one._stage) == K and len(many._stage) == 1
    for i in range(7):
        u, v = np.concatenate([s[i] for s in one._stage]), many._stage[0][i]
        assert u.dtype == np.asarray(v).dtype and np.array_equal(u, np.asarray(v)), i
    assert list(many._stage[0][6]) == [True, True, True, False, True, True, True]


def test_deferred_step_is_the_step():
    """act_deferred() hands out the rates and leaves the simulator step pending: the next packed observation is readable from
    the look-ahead result, the public arrays still show the old state, and whatever touches the simulator next (an observe, a
    reset, a draw from a stream) finds the step applied first -- trajectories are those of act()."""
    from v2xgnn.rl import native_sim
    if not native_sim.available():
        pytest.skip("libv2xsim.so not built")
    E, n = 5, 20
    envs = []
    for ahead in (False, True):
        env = BatchedEnviron(*_lanes(), 750, 1299, n_envs=E, seeds=[29 + 7 * e for e in range(E)], lookahead=ahead)
        env.new_random_game(n)
        envs.append(env)
    a, b = envs
    rng = np.random.default_rng(4)
    for it in range(40):
        act = rng.integers(0, 4, size=(E, n, 1))
        pos_before, had_job = b.pos.copy(), b._ahead is not None
        ra, rb = a.act(act), b.act_deferred(act)
        for x, y in zip(ra, rb):
            assert np.array_equal(x, y)
        assert b._step_pending and np.array_equal(b.pos, pos_before)
        assert np.array_equal(b.next_packed_observation(4), a.observe_packed(4)[0])
        assert b._step_pending == had_job and (had_job or it == 0 or it % 7 in (4, 6))   # (no look-ahead result to read from after a reset / a draw)
        if it % 7 == 3:
            a.new_random_game(n); b.new_random_game(n)         # applies the pending step, then resets; no look-ahead wasted
            assert not b._step_pending and b._ahead is None
        elif it % 7 == 5:
            assert a.streams[1].random() == b.streams[1].random() and not b._step_pending
        for x, y in zip(a.observe(4) + a.observe_packed(4), b.observe(4) + b.observe_packed(4)):
            assert np.array_equal(x, y)
        assert not b._step_pending
        for k in ("pos", "dirs", "_v2v_shadow", "V2V_channels_with_fastfading", "V2V_Interference_all", "_mt_keys", "_mt_pos"):
            assert np.array_equal(getattr(a, k), getattr(b, k)), (it, k)


def test_pool_survives_a_fork():
    """The library's pool threads do not exist in a forked child (pthread_atfork handler in csrc/v2xsim.c): a child that
    inherits a simulator steps it on a pool of its own and gets the parent's trajectory."""
    from v2xgnn.rl import native_sim
    if not native_sim.available() or not hasattr(os, "fork"):
        pytest.skip("libv2xsim.so not built / no fork")
    E, n = 6, 8
    env = BatchedEnviron(*_lanes(), 750, 1299, n_envs=E, seeds=[17 + 31 * e for e in range(E)], lookahead=True)
    env.new_random_game(n)
    act = np.zeros((E, n, 1), int)
    env.act(act)                                               # pool threads exist, a look-ahead step is in flight
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:                                               # child: drop the inherited job, three steps, report a digest
        try:
            env._ahead = None
            env.lookahead = False
            for _ in range(3):
                env.act(act)
            os.write(w, env.pos.tobytes() + env._mt_pos.tobytes())
        finally:
            os._exit(0)
    os.close(w)
    for _ in range(3):
        env.act(act)
    want = env.pos.tobytes() + env._mt_pos.tobytes()
    got = b""
    while len(got) < len(want):
        chunk = os.read(r, len(want) - len(got))
        if not chunk:
            break
        got += chunk
    os.close(r)
    _, status = os.waitpid(pid, 0)
    assert status == 0 and got == want


def test_fork_with_a_lookahead_in_flight_is_recomputed_in_the_child():
    """a child forked while a look-ahead job is in flight inherits the job dict but not the threads that were
    writing its arrays; it must not commit them.  The child here does NOTHING to protect itself (keeps _ahead, keeps
    look-ahead on): the pid recorded in the job makes _advance drop it and recompute -- the parent's trajectory."""
    from v2xgnn.rl import native_sim
    if not native_sim.available() or not hasattr(os, "fork"):
        pytest.skip("libv2xsim.so not built / no fork")
    E, n = 6, 8
    env = BatchedEnviron(*_lanes(), 750, 1299, n_envs=E, seeds=[5 + 13 * e for e in range(E)], lookahead=True)
    env.new_random_game(n)
    act = np.zeros((E, n, 1), int)
    env.act(act)
    assert env._ahead is not None and env._ahead["pid"] == os.getpid()
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        try:
            for _ in range(3):
                env.act(act)
            os.write(w, env.pos.tobytes() + env._mt_pos.tobytes() + env.V2V_channels_with_fastfading.tobytes())
        finally:
            os._exit(0)
    os.close(w)
    for _ in range(3):
        env.act(act)
    want = env.pos.tobytes() + env._mt_pos.tobytes() + env.V2V_channels_with_fastfading.tobytes()
    got = b""
    while len(got) < len(want):
        chunk = os.read(r, len(want) - len(got))
        if not chunk:
            break
        got += chunk
    os.close(r)
    _, status = os.waitpid(pid, 0)
    assert status == 0 and got == want


def test_two_host_threads_stepping_two_simulators():
    """ctypes releases the GIL, so two Python threads can be inside the library's loops at once; its pool has ONE job
    slot.  The second caller must run its loop alone instead of posting over the first one's job (which stranded
    environments of the first job: silently wrong arrays).  Two simulators stepped concurrently == stepped one after the other."""
    import threading
    from v2xgnn.rl import native_sim
    if not native_sim.available():
        pytest.skip("libv2xsim.so not built")
    E, n, steps = 24, 12, 12
    native_sim.set_threads(4)

    def run(seed0, out, concurrent_with=None):
        env = BatchedEnviron(*_lanes(), 750, 1299, n_envs=E, seeds=[seed0 + 7 * e for e in range(E)], lookahead=False)
        env.new_random_game(n)
        act = np.zeros((E, n, 1), int)
        if concurrent_with is not None:
            concurrent_with.wait()
        for _ in range(steps):
            env.act(act)
        out.append((env.pos.copy(), env._mt_pos.copy(), env.V2V_channels_with_fastfading.copy(), env.V2V_Interference_all.copy()))
    seq = {s: [] for s in (3, 1000)}
    for s in seq:
        run(s, seq[s])
    for rep in range(3):
        par = {s: [] for s in seq}
        bar = threading.Barrier(2)
        ths = [threading.Thread(target=run, args=(s, par[s], bar)) for s in seq]
        [t.start() for t in ths]
        [t.join() for t in ths]
        for s in seq:
            for a, b in zip(seq[s][0], par[s][0]):
                assert np.array_equal(a, b), (rep, s)


@pytest.mark.parametrize("threads,n,batch", [(1, 20, False), (6, 20, False), (4, 8, False), (12, 12, False), (1, 20, True), (6, 20, True),
                                             (12, 12, True)])
def test_native_rollout_is_T_single_steps(threads, n, batch):
    """v2xsim_rollout (the reference's loop shape -- ONE simulator, T sequential transitions with a B = 1 predict
    each, BS_brain.py:409-553 -- as one library call with the simulator step cut over a team of threads that works ahead of the
    caller): transitions, rates, the simulator's every array, its MT19937 stream AND numpy's process-wide stream (epsilon draws,
    random actions) bit-identical to T iterations of observe -> epsilon-greedy -> act in Python.  The predict is a ctypes
    callback here (a deterministic function of the packed observation); on the GPU it is v2x_forward_call."""
    import ctypes
    from v2xgnn.rl import native_sim
    if not native_sim.available():
        pytest.skip("libv2xsim.so not built")
    T, C = 37, 4
    ne = n * (n - 2)

    def fake_q(xe, col):
        s = xe[:, :13].astype(np.float64).sum(axis=1)[:, None] * (np.arange(C)[None, :] + 1.0) + col[:C].sum()
        return np.sin(s).astype(np.float32)

    def make():
        env = BatchedEnviron(*_lanes(), 750, 1299, n_envs=1, seeds=[1234 + n], lookahead=False)
        env.new_random_game(n)
        env.act(np.zeros((1, n, 1), int))
        return env
    pol = dict(eps_max=0.9, eps_min=0.01, eps_per_step=0.8 / 30, eps_steps=30.0, step_no0=3)
    # ---- T single steps in Python
    native_sim.set_threads(1)
    ref = make()
    np.random.seed(99)
    rec = {k: [] for k in ("xe", "xe_next", "col", "mask", "action", "v2v", "v2i")}
    n_greedy = 0
    for t in range(T):
        xe, mask, col, regular = ref.observe_packed(C)
        assert regular.all()
        step_no = pol["step_no0"] + t
        eps = pol["eps_max"] - pol["eps_per_step"] * step_no if step_no < pol["eps_steps"] else pol["eps_min"]
        if np.random.random() < eps:
            a = np.random.randint(0, C, size=(n, 1))
        else:
            a = np.argmax(fake_q(xe[0], col[0]), axis=1)[:, None]
            n_greedy += 1
        v2v, v2i, _ = ref.act(a[None])
        for k, v in zip(("xe", "col", "mask", "action", "v2v", "v2i"), (xe[0], col[0], mask[0], a[:, 0], v2v[0], v2i[0])):
            rec[k].append(np.array(v, copy=True))
        rec["xe_next"].append(ref.observe_packed(C)[0][0].copy())
    ref_np = np.random.get_state()
    assert 0 < n_greedy < T
    # ---- the same as one library call
    native_sim.set_threads(threads)
    env = make()
    np.random.seed(99)
    G = T if batch else 1                                    # batch: ONE predict call for all T observations
    xe_pin, col_pin, q_pin = np.zeros((G * n, 16), np.float32), np.zeros(G * ne, np.int32), np.zeros((G * n, C), np.float32)
    n_calls = [0]

    def predict(_ctx):
        n_calls[0] += 1
        for g_ in range(G):
            q_pin[g_ * n:(g_ + 1) * n] = fake_q(xe_pin[g_ * n:(g_ + 1) * n], col_pin[g_ * ne:(g_ + 1) * ne])
        return 0
    cb = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p)(predict)
    out = env.native_rollout(T, C, dict(pol, predict=ctypes.cast(cb, ctypes.c_void_p).value, predict_ctx=None,
                                        xe_pin=xe_pin, col_pin=col_pin, q_pin=q_pin, batch_predict=batch))
    native_sim.set_threads(1)
    assert out["done"] == T and out["n_greedy"] == n_greedy and n_calls[0] == (1 if batch else n_greedy)
    for k, o in (("xe", "xe"), ("xe_next", "xe_next"), ("col", "col"), ("mask", "mask"), ("action", "action"), ("v2v", "v2v_rate"), ("v2i", "v2i_rate")):
        assert np.array_equal(np.stack(rec[k]), out[o]), k
    got_np = np.random.get_state()
    assert got_np[2] == ref_np[2] and np.array_equal(got_np[1], ref_np[1])
    for k in ("pos", "dirs", "_v2i_shadow", "_v2v_shadow", "V2V_channels_abs", "V2I_channels_abs", "V2V_channels_with_fastfading",
              "V2I_channels_with_fastfading", "V2V_Interference_all", "_mt_keys", "_mt_pos", "V2I_Interference", "V2V_Interference"):
        assert np.array_equal(np.asarray(getattr(env, k)), np.asarray(getattr(ref, k))), k
    for a, b in zip(env.observe_packed(C), ref.observe_packed(C)):
        assert np.array_equal(a, b)
    # ... and both go on identically
    act = np.ones((1, n, 1), int)
    for _ in range(3):
        ra, rb_ = env.act(act), ref.act(act)
        assert all(np.array_equal(x, y) for x, y in zip(ra, rb_))
    # a failing predict: the simulator stands where the completed transitions left it
    env2, ref2 = make(), make()
    np.random.seed(7)
    calls = [0]

    def flaky(_ctx):
        calls[0] += 1
        if calls[0] == (1 if batch else 3):
            return 1
        q_pin[:n] = fake_q(xe_pin[:n], col_pin[:ne])
        return 0
    cb2 = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p)(flaky)
    native_sim.set_threads(threads)
    out2 = env2.native_rollout(T, C, dict(pol, eps_max=0.3, predict=ctypes.cast(cb2, ctypes.c_void_p).value, predict_ctx=None,
                                          xe_pin=xe_pin, col_pin=col_pin, q_pin=q_pin, batch_predict=batch))
    np_after = np.random.get_state()
    native_sim.set_threads(1)
    assert out2["rc"] <= -1000 and 0 <= out2["done"] < T and out2["done"] == -1000 - out2["rc"]
    for t in range(out2["done"]):
        ref2.act(out2["action"][t][None, :, None])
    for k in ("pos", "V2V_channels_with_fastfading", "_mt_keys", "_mt_pos"):
        assert np.array_equal(np.asarray(getattr(env2, k)), np.asarray(getattr(ref2, k))), k
    # numpy's stream: after the failed transition's epsilon draw (the draws of the completed transitions + that one)
    np.random.seed(7)
    for t in range(out2["done"] + 1):
        step_no = pol["step_no0"] + t
        eps = 0.3 - pol["eps_per_step"] * step_no if step_no < pol["eps_steps"] else pol["eps_min"]
        if np.random.random() < eps:
            assert t < out2["done"]
            np.random.randint(0, C, size=(n, 1))
    assert np.random.get_state()[2] == np_after[2] and np.array_equal(np.random.get_state()[1], np_after[1])
