"""GPU: the reference's training loop (config 1: 4 links, F=16) end to end on the engine -- simulator -> agent ->
BS -> C-ABI -> kernels -- and the compact agent path against the dict path and the oracle."""
import os
import random

import numpy as np
import pytest

from util import assert_fwd_close

pytestmark = pytest.mark.gpu


def _setup(seed, batch=64, n_veh=4, feat=16, device_replay='auto'):
    from v2xgnn.rl import Agent, RL_Config
    from test_rl_env import make_env
    random.seed(seed)
    np.random.seed(seed)
    cfg = RL_Config()
    cfg.set_train_value(feat, 0.5, batch, 1, 0.1)
    env = make_env()
    if n_veh != env.n_Veh:
        env.new_random_game(n_veh)
    agent = Agent(n_veh, env.n_RB, env.n_Neighbor, feat, env, cfg, seed=seed, device_replay=device_replay)
    return agent, env, cfg


def test_training_loop_runs_and_learns_signal(tmp_path):
    agent, env, cfg = _setup(11)
    w0 = [w.copy() for w in agent.brain.model.get_weights()]
    loss, reward_step, reward_ep, q_mean, q_max, _, _ = agent.train(2, 5, save_dir=str(tmp_path), save_interval=2)
    assert loss.shape == (4, 2, 5) and np.all(np.isfinite(loss)) and np.all(loss >= 0)
    assert reward_step.shape == (2, 5, 50) and np.all(np.isfinite(reward_step))
    assert agent.num_step == 2 * 5 * 50 and len(agent.memory.samples) == 500
    assert np.all(q_max >= q_mean)
    w1 = agent.brain.model.get_weights()
    assert any(not np.array_equal(a, b) for a, b in zip(w0, w1))
    # num_step hit 500 on the last replay -> target network synced (BS_brain.py:847-848)
    for a, b in zip(w1, agent.brain.target_model.get_weights()):
        assert np.array_equal(a, b)
    tag = '-Episode-2-Step-5-Batch-64.h5'
    assert os.path.exists(os.path.join(str(tmp_path), 'Q-Network_model_weights' + tag))
    assert os.path.exists(os.path.join(str(tmp_path), 'Target-Network_model_weights' + tag))
    # evaluation driver on the weights just saved (counterpart of RL_Run_main.py)
    from v2xgnn.rl.run import load_trained_model, run_test
    cfg.Num_Episodes, cfg.Num_Train_Steps = 2, 5
    cfg.set_test_values(1, 4, True, 1, 0.1)
    agent2 = load_trained_model(env, cfg, str(tmp_path), seed=99)
    for a, b in zip(w1, agent2.brain.model.get_weights()):
        assert np.array_equal(a, b)
    res = run_test(cfg, agent2)
    assert res['Reward'].shape == res['RA_Reward'].shape == res['Opt_Reward'].shape == (1, 4)
    assert np.all(res['Opt_Reward'] >= res['Reward'] - 1e-9) and np.all(res['Opt_Reward'] >= res['RA_Reward'] - 1e-9)


def test_compact_path_equals_dict_path_and_oracle():
    from oracle import compact
    from oracle.spec import GnnSpec as OSpec
    from v2xgnn.packing import PackedBatch
    agent, env, cfg = _setup(12, device_replay=False)
    agent.generate_d2d_transition(40)
    batch = agent.memory.sample(32)
    s = np.stack([b[0][0] for b in batch])
    states, adj = s[:, :52].reshape(32, 4, 13), s[:, 52:].reshape(32, 4, 4)
    q_compact = agent._predict(states, adj)                                           # [N, B, C]
    q_dict = np.stack(agent.brain.predict(agent._feed(states, adj)))
    assert np.array_equal(q_compact, q_dict)
    spec = OSpec(n_nodes=4, n_channels=4, feat_dim=16, n_mp_layers=2)
    params = compact.params_from_list(spec, agent.brain.model.get_weights())
    pb = PackedBatch.from_dense(states[:, :, :9], states[:, :, 9:], adj)
    graph = ((np.arange(33) * 4).astype(np.int32), pb.row_ptr, pb.col_idx)
    om = compact.OracleModel(spec, params, dtype=np.float64)
    q_ref = om.predict(states[:, :, :9].reshape(128, 9), states[:, :, 9:].reshape(128, 4), graph)
    assert_fwd_close(np.transpose(q_compact, (1, 0, 2)).reshape(-1, 4), q_ref)

def test_twenty_link_agent_step():
    """The same loop at the headline topology size (20 links, F=64): one replay step."""
    agent, env, cfg = _setup(13, batch=128, n_veh=20, feat=64)
    agent.num_Episodes, agent.num_Train_Step = 1, 1
    r = agent.generate_d2d_transition(10)
    assert np.all(np.isfinite(r))
    result, q_mean, q_max, _, _ = agent.replay()
    assert len(q_mean) == 20
    assert all(np.isfinite(result.history['D%d_Decide_Output_loss' % (k + 1)][0]) for k in range(20))


def test_dqn_driver_twenty_links_rccl_single_rank():
    """BASELINE config 3 plumbing on one GPU: the training driver under torch.distributed.run with backend nccl (RCCL),
    20 links x 64 features, data-parallel fit path forced on with one rank."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, V2X_FORCE_DP="1", MASTER_ADDR="127.0.0.1", PYTHONPATH=root)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29547", "-m", "v2xgnn.rl.train", "--links", "20", "--feedback", "64", "--batch", "512",
           "--episodes", "1", "--train-steps", "2", "--use-graph"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["links"] == 20 and res["env_steps"] == 100 and len(res["mean_loss_last_episode"]) == 20
    assert all(np.isfinite(v) and v >= 0 for v in res["mean_loss_last_episode"])


@pytest.mark.parametrize("n_veh,feat,batch,n_trans", [(4, 16, 64, 40), (20, 64, 96, 130)])
def test_device_replay_equals_host_replay(n_veh, feat, batch, n_trans):
    """The HBM-resident replay step (gather -> online/target forward -> target rule -> fit, rl/replay.py) against the
    host path that reproduces the reference's payloads: same sampled transitions, same targets, same update."""
    out = []
    for dev in (False, True):
        agent, env, cfg = _setup(21, batch=batch, n_veh=n_veh, feat=feat, device_replay=dev)
        assert (agent.device_replay is not None) == dev
        agent.num_Episodes, agent.num_Train_Step = 1, 4
        steps = []
        for _ in range(3):                                  # first replay samples with replacement, later ones without
            agent.generate_d2d_transition(n_trans)
            result, q_mean, q_max, _, _ = agent.replay()
            steps.append((np.array([result.history['D%d_Decide_Output_loss' % (k + 1)][0] for k in range(n_veh)]),
                          q_mean, q_max))
        out.append((steps, np.concatenate([w.ravel() for w in agent.brain.model.get_weights()])))
    for (l0, m0, x0), (l1, m1, x1) in zip(out[0][0], out[1][0]):
        assert np.allclose(l0, l1, rtol=2e-4, atol=1e-6)
        assert np.allclose(m0, m1, rtol=1e-5, atol=1e-6) and np.allclose(x0, x1, rtol=1e-5, atol=1e-6)
    assert np.allclose(out[0][1], out[1][1], rtol=1e-3, atol=2e-5)


def test_device_replay_ring_and_gather():
    """Ring overwrite + logical (FIFO) indexing of the HBM replay memory and the two glue kernels against numpy."""
    import torch
    from v2xgnn.rl.replay import DeviceReplay
    rng = np.random.default_rng(3)
    n, cap = 5, 37
    rep = DeviceReplay(cap, n)
    adj = np.ones((n, n)) - np.eye(n)
    for q in range(n):
        adj[(q + 1) % n, q] = 0
    log = []
    for i in range(90):                                     # wraps the 37-slot ring twice, flushed in uneven groups
        x, e = rng.normal(size=(n, 9)), rng.normal(size=(n, 4))
        x2, e2 = rng.normal(size=(n, 9)), rng.normal(size=(n, 4))
        a, r = rng.integers(0, 4, size=n), float(rng.normal())
        rep.add(x, e, adj, a, r, x2, e2)
        log.append((x, e, a, r, x2, e2))
        if i % 7 == 3:
            rep.flush()
    assert len(rep) == cap
    kept = log[-cap:]                                       # FIFO: logical index 0 = oldest surviving transition
    idx = rng.integers(0, cap, size=50)
    sb, sb_next, action, reward = rep.sample(idx)
    xe = sb.xe.cpu().numpy().reshape(50, n, 16)
    xe2 = sb_next.xe.cpu().numpy().reshape(50, n, 16)
    for k, i in enumerate(idx):
        x, e, a, r, x2, e2 = kept[i]
        assert np.array_equal(xe[k, :, :9], x.astype(np.float32)) and np.array_equal(xe[k, :, 9:13], e.astype(np.float32))
        assert np.array_equal(xe2[k, :, :9], x2.astype(np.float32)) and np.all(xe[k, :, 13:] == 0)
        assert np.array_equal(action[k].cpu().numpy(), a) and float(reward[k]) == r
    assert sb.n_edges == 50 * n * (n - 2) and np.array_equal(sb.row_ptr.cpu().numpy(), np.arange(50 * n + 1) * (n - 2))
    q = torch.from_numpy(rng.normal(size=(50 * n, 4)).astype(np.float32)).cuda()
    qn = torch.from_numpy(rng.normal(size=(50 * n, 4)).astype(np.float32)).cuda()
    y = rep.dqn_targets(q, qn, action, reward, 0.37).cpu().numpy()
    ref = q.cpu().numpy().copy()
    # r + GAMMA * max q' in float64, rounded once (the reference's numpy-1.x scalar semantics, BS_brain.py:690)
    tgt = (reward.cpu().numpy()[:, None] + 0.37 * qn.cpu().numpy().reshape(50, n, 4).max(axis=2).astype(np.float64)).astype(np.float32)
    ref.reshape(50, n, 4)[np.arange(50)[:, None], np.arange(n)[None, :], action.cpu().numpy()] = tgt
    assert np.array_equal(y, ref)


def test_device_replay_index_sharding_sums_to_the_full_batch_gradient():
    """Data-parallel device replay: every rank draws the SAME indices and processes its contiguous share.  Emulated on
    one GPU: the two ranks' gradients (global Huber denominator) must add up to the gradient of the whole minibatch,
    and their Q statistics to the whole-batch statistics."""
    agent, env, cfg = _setup(31, batch=64, n_veh=20, feat=64, device_replay=True)
    agent.num_Episodes, agent.num_Train_Step = 1, 4
    agent.generate_d2d_transition(150)
    model = agent.brain.model
    eng = model.engine

    class FakeDist(object):
        def __init__(self):
            self.reduced = []

        def all_reduce(self, t, group=None, op=None):
            self.reduced.append(t.clone())

    class FakeTrainer(object):
        def __init__(self, rank, world):
            self.rank, self.world, self.group, self.dist = rank, world, None, FakeDist()
            self.grad = None

        def train_step(self, batch, y, n_graphs_global, want_loss=True):
            loss = eng.forward_backward(batch, y, n_global=n_graphs_global)      # no optimizer step: weights stay put
            self.grad = eng.get_grad_flat().astype(np.float64)
            return loss

    state = np.random.get_state()
    full = FakeTrainer(0, 1)
    model.trainer = full
    _, qm_full, qx_full, _, _ = agent.replay()
    parts, stats = [], []
    for r in range(2):
        np.random.set_state(state)                                               # same draw on every rank
        tr = FakeTrainer(r, 2)
        model.trainer = tr
        agent.replay()
        parts.append(tr.grad)
        stats.append(tr.dist.reduced[-1].cpu().numpy())                          # the rank's partial Q-statistic sums
    model.trainer = None
    from util import assert_grad_close
    assert_grad_close(parts[0] + parts[1], full.grad, "sum of the two ranks' gradients")
    tot = (stats[0] + stats[1]) / 64
    assert np.allclose(tot[0], qm_full, rtol=1e-5) and np.allclose(tot[1], qx_full, rtol=1e-5)


def test_device_replay_handles_a_link_that_is_its_own_receiver():
    """Two vehicles on one spot can make a link its own receiver (Environment.py:365-375): in-degree n-1 instead of
    n-2.  Minibatches containing such a transition are expanded from the per-link source masks; the CSR must equal the
    host packer's and the forward pass must agree with the host path."""
    import torch
    from v2xgnn import GnnSpec, GnnEngine, PackedBatch
    from v2xgnn.packing import adj_to_csr
    from v2xgnn.rl.replay import DeviceReplay
    from oracle import compact as oc
    from util import f32_params
    rng = np.random.default_rng(4)
    n = 8
    rep = DeviceReplay(64, n)
    log = []
    for i in range(20):
        adj = np.ones((n, n)) - np.eye(n)
        for q in range(n):
            adj[(q + 1 + i) % n if (q + 1 + i) % n != q else (q + 2) % n, q] = 0
        if i in (5, 11):
            adj[:, 3] = 1
            adj[3, 3] = 0                                     # link 3 is its own receiver: in-degree n-1
        x, e = rng.normal(size=(n, 9)), rng.normal(size=(n, 4))
        rep.add(x, e, adj, rng.integers(0, 4, size=n), 0.5, x + 1, e)
        log.append((x, e, adj))
    for idx in (np.array([0, 1, 2, 7]), np.array([4, 5, 6, 11, 11, 19])):
        sb, sb_next, action, reward = rep.sample(idx)
        xs, es, adjs = (np.stack([log[i][j] for i in idx]) for j in range(3))
        ref = PackedBatch.from_dense(xs, es, adjs)
        assert np.array_equal(sb.row_ptr.cpu().numpy(), ref.row_ptr) and np.array_equal(sb.col_idx.cpu().numpy()[:ref.n_edges], ref.col_idx)
        assert sb.n_edges == ref.n_edges and sb.max_edges >= ref.max_edges
        spec = GnnSpec(n_nodes=n, feat_dim=16)
        eng = GnnEngine(spec)
        eng.set_weights(oc.params_to_list(f32_params(spec, np.random.default_rng(1))))
        assert np.array_equal(eng.forward(sb).cpu().numpy(), eng.forward(ref))
        eng.close()


@pytest.mark.parametrize("n_veh,feat,batch,n_envs", [(4, 16, 64, 6), (20, 64, 256, 10)])
def test_batched_rollouts_on_the_engine(n_veh, feat, batch, n_envs):
    """The DQN loop with E simulators stepped as arrays (native C + OpenMP step when libv2xsim.so is built), one forward
    pass per step for all greedy environments (k_predict_small for up to 256 node rows), transitions stored in the HBM
    replay memory with one vectorised add per step: same episode as with the HOST replay memory (same transitions, same
    minibatches, losses and weights to fp32 rounding), and the numpy simulator step gives the same trajectory."""
    from v2xgnn.rl import Agent, RL_Config, native_sim
    from v2xgnn.rl.train import start_env_batched

    def episode(device_replay, native):
        random.seed(21)
        np.random.seed(21)
        old = os.environ.get("V2X_SIM_NATIVE")
        os.environ["V2X_SIM_NATIVE"] = "1" if native else "0"
        native_sim._tried, native_sim._lib = False, None
        try:
            env = start_env_batched(n_veh, n_envs, 21)
        finally:
            if old is None:
                del os.environ["V2X_SIM_NATIVE"]
            else:
                os.environ["V2X_SIM_NATIVE"] = old
            native_sim._tried, native_sim._lib = False, None
        cfg = RL_Config()
        cfg.set_train_value(feat, 0.5, batch, 1, 0.1)
        agent = Agent(n_veh, env.n_RB, env.n_Neighbor, feat, env, cfg, seed=21, device_replay=device_replay)
        loss, reward_step, _, q_mean, q_max, _, _ = agent.train(1, 4)
        w = np.concatenate([a.ravel() for a in agent.brain.model.get_weights()])
        return env, agent, loss, reward_step, q_mean, w

    env_d, ag_d, loss_d, rew_d, qm_d, w_d = episode(True, native_sim.available())
    env_h, ag_h, loss_h, rew_h, qm_h, w_h = episode(False, native_sim.available())
    assert ag_d.device_replay is not None and ag_h.device_replay is None
    assert env_d.native == native_sim.available()
    assert np.all(np.isfinite(loss_d)) and ag_d.num_step == ag_h.num_step == 4 * (-(-50 // n_envs) * n_envs)
    assert np.array_equal(rew_d, rew_h)                       # same rollouts (exploration stream, greedy actions)
    assert np.allclose(loss_d, loss_h, rtol=2e-4, atol=1e-6) and np.allclose(qm_d, qm_h, rtol=1e-4, atol=1e-5)
    assert np.allclose(w_d, w_h, rtol=1e-3, atol=2e-5)
    if native_sim.available():                                # the numpy step: same trajectory to libm rounding
        env_n, ag_n, loss_n, rew_n, qm_n, w_n = episode(True, False)
        assert not env_n.native
        assert np.array_equal(env_n.pos, env_d.pos) and np.array_equal(env_n.dirs, env_d.dirs)
        assert np.allclose(rew_n, rew_d, rtol=1e-9, atol=1e-12)
        assert np.allclose(loss_n, loss_d, rtol=2e-4, atol=1e-6)


@pytest.mark.parametrize("n_veh,feat,batch,n_envs,capacity,sampler_min", [(4, 16, 64, 6, None, None), (20, 64, 256, 50, None, None),
                                                                            (4, 16, 64, 6, 130, None), (20, 64, 256, 50, None, 300)])
def test_packed_rollout_with_lookahead_is_the_array_rollout_bitwise(n_veh, feat, batch, n_envs, capacity, sampler_min, monkeypatch):
    """the rollout of a batched step as observe_packed -> pinned copy -> one predict launch -> argmax ->
    add_many_packed, with the next simulator step computed on the library's worker thread meanwhile -- against the array path
    (float64 observations, dense adjacency, PackedBatch.from_dense, add_many; V2X_RL_PACKED=0) on a simulator without
    look-ahead: same epsilon draws, same greedy actions, same rewards, the same bytes in the replay memory and bit-identical
    losses and weights after two episodes (a reset drops a started look-ahead step).  Inside Agent.train the packed rollout
    also copies the observation half of its transitions to their replay slots and draws / uploads the replay's minibatch before
    it scores (the GPU is still fitting); capacity 130: a replay memory that wraps every third step, where those shortcuts
    must step aside; sampler_min: the minibatch draw restated in libv2xsim.so, on a helper thread in the packed run and in line
    in the array run (the last case: 300 stored transitions instead of 16,384 switch it on)."""
    from v2xgnn.rl import Agent, RL_Config, native_sim
    from v2xgnn.rl import agent as agent_mod
    from v2xgnn.rl.train import start_env_batched
    if not native_sim.available():
        pytest.skip("libv2xsim.so not built")
    if capacity is not None:
        monkeypatch.setattr(agent_mod, "MEMORY_CAPACITY", capacity)
    if sampler_min is not None:        # the library's sampler from 300 stored transitions on: in the packed run it works on a helper thread
        monkeypatch.setattr(agent_mod, "NATIVE_SAMPLER_MIN", sampler_min)

    def run(packed, lookahead):
        random.seed(33)
        np.random.seed(33)
        old = os.environ.get("V2X_RL_PACKED")
        os.environ["V2X_RL_PACKED"] = "1" if packed else "0"
        try:
            env = start_env_batched(n_veh, n_envs, 33, lookahead=lookahead)
            cfg = RL_Config()
            cfg.set_train_value(feat, 0.5, batch, 1, 0.1)
            agent = Agent(n_veh, env.n_RB, env.n_Neighbor, feat, env, cfg, seed=33, device_replay=True)
            loss, reward_step, _, q_mean, q_max, _, _ = agent.train(2, 6)
        finally:
            if old is None:
                del os.environ["V2X_RL_PACKED"]
            else:
                os.environ["V2X_RL_PACKED"] = old
        rep = agent.device_replay
        rep.flush()
        k = rep.size
        mem = [t[:k].cpu().numpy() for t in (rep.xe, rep.xe_next, rep.col, rep.mask, rep.action, rep.reward)]
        return env, agent, loss, reward_step, q_mean, np.concatenate([a.ravel() for a in agent.brain.model.get_weights()]), mem

    env_p, ag_p, loss_p, rew_p, qm_p, w_p, mem_p = run(True, True)
    env_a, ag_a, loss_a, rew_a, qm_a, w_a, mem_a = run(False, False)
    assert env_p.lookahead and not env_a.lookahead and getattr(ag_p, "_rollout_io", None) is not None
    assert getattr(ag_a, "_rollout_io", None) is None
    assert ag_p.num_step == ag_a.num_step and len(ag_p.memory.samples) == len(ag_a.memory.samples)
    assert np.array_equal(rew_p, rew_a)
    for a, b in zip(mem_p, mem_a):
        assert a.shape == b.shape and np.array_equal(a, b)
    assert np.array_equal(loss_p, loss_a) and np.array_equal(qm_p, qm_a) and np.array_equal(w_p, w_a)
    assert np.array_equal(env_p.pos, env_a.pos) and np.array_equal(env_p._mt_keys, env_a._mt_keys)


@pytest.mark.parametrize("B", [3, 50, 200])
def test_forward_to_host_from_pinned_memory_equals_forward(B):
    """GnnEngine.forward_to_host on a batch whose arrays live in PINNED HOST memory (the rollout predict: the kernels read the
    observations over the bus, the library copies Q back and synchronises) against the ordinary device-batch forward:
    the same bits, on the one-launch predict (3 graphs), the split-tile kernels (50) and the whole-tile kernels (200)."""
    import torch
    from v2xgnn import GnnSpec, GnnEngine, PackedBatch
    from v2xgnn.engine import DeviceBatch
    from util import f32_params, random_inputs
    from oracle import compact as oc
    N, F = 20, 64
    spec = GnnSpec(n_nodes=N, feat_dim=F)
    rng = np.random.default_rng(77 + B)
    eng = GnnEngine(spec)
    eng.set_weights(oc.params_to_list(f32_params(spec, rng)))
    x, e, adj = random_inputs(rng, B, N, ref_topology=True)
    pb = PackedBatch.from_dense(x, e, adj)
    db = eng.to_device(pb)
    q_dev = eng.forward(db).cpu().numpy()
    pin = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory()
    hb = DeviceBatch.from_tensors(B, N, pin(pb.xe), db.row_ptr, pin(pb.col_idx), pb.max_edges)
    q = np.full((B * N, spec.n_channels), np.nan, np.float32)
    for _ in range(3):
        assert eng.forward_to_host(hb, q) is q and np.array_equal(q, q_dev)
    with pytest.raises(ValueError):
        eng.forward_to_host(hb, q.astype(np.float64))
    with pytest.raises(ValueError):
        eng.forward_to_host(pb, q)
    eng.close()


def test_native_sampler_is_this_box_numpys_choice():
    """libv2xsim's restatement of np.random.choice(n, k, replace=False) follows a numpy-version-specific
    legacy algorithm; the draw-for-draw comparison is a CPU test that would otherwise only ever see the build container's numpy.
    The same comparison against the numpy of the GPU box, plus the look-ahead form and its guard."""
    import test_rl_agent as t
    t.test_native_sampler_is_numpys_choice_without_replacement()
    from v2xgnn.rl import native_sim
    np.random.seed(123)
    want, after = np.random.choice(70000, 4096, replace=False), np.random.random(2)
    np.random.seed(123)
    ca = native_sim.ChoiceAhead(70000, 4096)
    assert np.array_equal(ca.result(), want) and np.array_equal(np.random.random(2), after)


def test_forward_to_host_refuses_pageable_host_memory():
    """forward_to_host hands host pointers to kernels (on_device = 1); a batch of pageable CPU tensors must be a
    ValueError, not a GPU page fault.  Pinned tensors pass (v2x_device_addressable checks the mapping address)."""
    import torch
    import v2xgnn
    from v2xgnn import GnnSpec, GnnEngine
    from v2xgnn.engine import DeviceBatch
    spec = GnnSpec(n_nodes=4, feat_dim=16, n_mp_layers=2)
    eng = GnnEngine(spec)
    rng = np.random.default_rng(3)
    eng.set_weights([rng.normal(0, 0.2, size=s).astype(np.float32) for s in v2xgnn.keras_list_shapes(spec)])
    B, n = 3, 4
    xe = torch.from_numpy(rng.normal(0.8, 0.3, size=(B * n, 16)).astype(np.float32))
    xe[:, 13:] = 0
    rp = torch.arange(B * n + 1, dtype=torch.int32) * 2
    col = torch.tensor([[(q + 1) % n, (q + 2) % n] for q in range(n)] * B, dtype=torch.int32).sort(dim=1).values.reshape(-1).contiguous()
    q = np.empty((B * n, 4), np.float32)
    lib = eng._lib
    assert lib.v2x_device_addressable(xe.data_ptr()) == 0
    with pytest.raises(ValueError, match="cannot address"):
        eng.forward_to_host(DeviceBatch.from_tensors(B, n, xe, rp.cuda(), col.cuda(), 2 * n), q)
    pinned = [t.pin_memory() for t in (xe, rp, col)]
    assert all(lib.v2x_device_addressable(t.data_ptr()) == 1 for t in pinned)
    got = eng.forward_to_host(DeviceBatch.from_tensors(B, n, pinned[0], pinned[1], pinned[2], 2 * n), q).copy()
    want = eng.forward(DeviceBatch.from_tensors(B, n, xe.cuda(), rp.cuda(), col.cuda(), 2 * n)).cpu().numpy()
    assert np.array_equal(got, want)
    eng.close()


@pytest.mark.parametrize("n_veh,feat,batch,batch_predict", [(4, 16, 64, True), (20, 64, 96, True), (20, 64, 96, False)])
def test_native_rollout_of_one_simulator_is_the_per_transition_rollout_bitwise(n_veh, feat, batch, batch_predict, monkeypatch):
    """the reference's loop shape (ONE simulator, 50 sequential transitions with a B = 1 predict each,
    BS_brain.py:409-553, :818-832) as one library call per rollout (v2xsim_rollout; the predict is v2x_forward_call on pinned
    buffers; the simulator step is computed by a team of threads while the predict is in flight -- or, batch_predict, ALL 50
    observations first and one predict of 50 graphs: they do not depend on the actions) against the same agent with
    V2X_RL_NATIVE_ROLLOUT=0 (one _packed_iteration per transition): rewards, the bytes of the replay memory, losses, Q
    statistics, weights, both random streams -- bit for bit over two episodes (a reset in between)."""
    from v2xgnn.rl import Agent, RL_Config, native_sim
    from v2xgnn.rl.train import start_env_batched
    if not native_sim.available():
        pytest.skip("libv2xsim.so not built")

    def run(native):
        random.seed(41)
        np.random.seed(41)
        monkeypatch.setenv("V2X_RL_NATIVE_ROLLOUT", "1" if native else "0")
        monkeypatch.setenv("V2X_RL_ROLLOUT_BATCH_PREDICT", "1" if batch_predict else "0")
        native_sim.set_threads(6 if native else 1)
        env = start_env_batched(n_veh, 1, 41, lookahead=not native)
        cfg = RL_Config()
        cfg.set_train_value(feat, 0.5, batch, 1, 0.1)
        agent = Agent(n_veh, env.n_RB, env.n_Neighbor, feat, env, cfg, seed=41, device_replay=True)
        loss, reward_step, _, q_mean, q_max, _, _ = agent.train(2, 8)
        rep = agent.device_replay
        rep.flush()
        k = rep.size
        mem = [t[:k].cpu().numpy() for t in (rep.xe, rep.xe_next, rep.col, rep.mask, rep.action, rep.reward)]
        return env, agent, loss, reward_step, q_mean, np.concatenate([a.ravel() for a in agent.brain.model.get_weights()]), mem, np.random.get_state()

    env_n, ag_n, loss_n, rew_n, qm_n, w_n, mem_n, np_n = run(True)
    env_p, ag_p, loss_p, rew_p, qm_p, w_p, mem_p, np_p = run(False)
    native_sim.set_threads(1)
    assert getattr(ag_n, "_native_io", None) and all("closure" in io for io in ag_n._native_io.values())
    assert (50 in ag_n._native_io) == batch_predict and not getattr(ag_p, "_native_io", None)
    assert ag_n.num_step == ag_p.num_step == 2 * 8 * 50 and ag_n.epsilon == ag_p.epsilon
    assert np.array_equal(rew_n, rew_p)
    for a, b in zip(mem_n, mem_p):
        assert a.shape == b.shape and np.array_equal(a, b)
    assert np.array_equal(loss_n, loss_p) and np.array_equal(qm_n, qm_p) and np.array_equal(w_n, w_p)
    assert np.array_equal(env_n.pos, env_p.pos) and np.array_equal(env_n._mt_keys, env_p._mt_keys)
    assert np_n[2] == np_p[2] and np.array_equal(np_n[1], np_p[1])


def test_multi_gather_and_q_statistics_kernels():
    """v2x_gather_rows_multi (the five gathers of a replay minibatch as one launch) against torch indexing, and v2x_q_stats (the Q
    statistics of BS_brain.py:743-746 as float64 sums per link, two-level with a fixed order) against torch's float64 sums; twice
    in a row (the kernel re-arms its arrival counters) and bit-identical between the two calls."""
    import ctypes as C
    import torch
    import v2xgnn
    lib = v2xgnn.load_library()
    g = torch.Generator(device="cpu").manual_seed(5)
    n_store, k = 1000, 333
    srcs = [torch.randn((n_store, 20, 16), generator=g), torch.randint(0, 9, (n_store, 20), generator=g, dtype=torch.int32),
            torch.randn(n_store, generator=g, dtype=torch.float64), torch.randint(0, 99, (n_store, 7), generator=g, dtype=torch.int32)]
    srcs = [t.cuda() for t in srcs]
    idx = torch.randint(0, n_store, (k,), generator=g, dtype=torch.int32).cuda()
    dsts = [torch.empty((k,) + tuple(t.shape[1:]), dtype=t.dtype, device="cuda") for t in srcs]
    n = len(srcs)
    sp, dp, rb = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_int64 * n)()
    for j, (s, d) in enumerate(zip(srcs, dsts)):
        sp[j], dp[j] = s.data_ptr(), d.data_ptr()
        rb[j] = s[0].numel() * s.element_size() if s.dim() > 1 else s.element_size()
    assert lib.v2x_gather_rows_multi(n, sp, dp, rb, idx.data_ptr(), k, None) == 0
    torch.cuda.synchronize()
    for s, d in zip(srcs, dsts):
        assert torch.equal(d, s[idx.long()])
    assert lib.v2x_gather_rows_multi(9, sp, dp, rb, idx.data_ptr(), k, None) == -1
    for B, N, Cc in ((4096, 20, 4), (7, 4, 4), (300, 31, 4)):
        y = torch.randn((B * N, Cc), generator=g).cuda() * 3 + 1
        out1 = torch.zeros((2, N), dtype=torch.float64, device="cuda")
        out2 = torch.zeros_like(out1)
        assert lib.v2x_q_stats(y.data_ptr(), B, N, Cc, out1.data_ptr(), None) == 0
        assert lib.v2x_q_stats(y.data_ptr(), B, N, Cc, out2.data_ptr(), None) == 0
        torch.cuda.synchronize()
        y3 = y.view(B, N, Cc)
        want = torch.stack([y3.sum(dim=(0, 2), dtype=torch.float64), y3.amax(dim=2).sum(dim=0, dtype=torch.float64)])
        assert torch.equal(out1, out2)
        assert torch.allclose(out1, want, rtol=1e-12, atol=1e-9), (B, N)
