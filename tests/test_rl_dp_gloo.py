"""CPU, world_size 2, gloo: the full DQN loop (simulator -> agent -> BS -> data-parallel fit) of BASELINE config 3's
scheme at reference size.  Every rank runs the same seeded simulator and replay sampling; each fit step shards the
minibatch and all-reduces the gradient.  Result must equal the single-process loop (the compute engine injected here
is the float64 oracle; on the GPU box it is the gfx950 engine over RCCL)."""
import os
import random
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from v2xgnn import BS
from v2xgnn.rl import Agent, RL_Config
from v2xgnn.rl.train import start_env
from oracle_engine import OracleEngine


def _run(data_parallel):
    random.seed(5)
    np.random.seed(5)
    cfg = RL_Config()
    cfg.set_train_value(16, 0.5, 32, 1, 0.1)
    env = start_env(4)
    brain = BS(4, 3, 1, 16, 1, 4, seed=3, data_parallel=data_parallel, engine_factory=lambda spec: OracleEngine(spec))
    agent = Agent(4, env.n_RB, env.n_Neighbor, 16, env, cfg, brain=brain)
    loss, reward_step, _, q_mean, _, _, _ = agent.train(1, 3)
    w = np.concatenate([a.ravel() for a in brain.model.get_weights()])
    mem = np.stack([s[0][0] for s in agent.memory.samples])
    return w, loss, reward_step, mem


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = _run(True)
    finally:
        dist.destroy_process_group()


def test_two_rank_dqn_loop_equals_single_process_loop():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    w1, loss1, rew1, mem1 = _run(False)
    for r in (0, 1):
        w, loss, rew, mem = ret[r]
        assert np.array_equal(mem, mem1)                      # identical rollouts on every rank
        assert np.array_equal(rew, rew1)
        assert np.allclose(loss, loss1, rtol=1e-9, atol=1e-12)
        assert np.allclose(w, w1, rtol=1e-6, atol=1e-7)       # get_weights() rounds to fp32
    assert np.array_equal(ret[0][0], ret[1][0])               # replicas stay bit-identical
    assert not np.allclose(w1, _initial_weights())            # and the loop really trained


def _initial_weights():
    brain = BS(4, 3, 1, 16, 1, 4, seed=3, engine_factory=lambda spec: OracleEngine(spec))
    return np.concatenate([a.ravel() for a in brain.model.get_weights()])


# ---------------------------------------------------------------------------------------------- sharded rollouts
def _run_sharded(rank, train_steps):
    random.seed(5 + 7919 * rank)                              # every rank its own simulator and exploration stream
    np.random.seed(5 + 7919 * rank)
    cfg = RL_Config()
    cfg.set_train_value(16, 0.5, 32, 1, 0.1)
    env = start_env(4)
    brain = BS(4, 3, 1, 16, 1, 4, seed=3, data_parallel=True, engine_factory=lambda spec: OracleEngine(spec))
    agent = Agent(4, env.n_RB, env.n_Neighbor, 16, env, cfg, brain=brain, rollouts='sharded')
    loss, reward_step, _, q_mean, _, _, _ = agent.train(1, train_steps)
    w = np.concatenate([a.ravel() for a in brain.model.get_weights()])
    wt = np.concatenate([a.ravel() for a in brain.target_model.get_weights()])
    mem = np.stack([s[0][0] for s in agent.memory.samples])
    return w, wt, loss, q_mean, mem, agent.num_step, agent.num_transition


def _worker_sharded(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = _run_sharded(rank, 10)
    finally:
        dist.destroy_process_group()


def test_two_rank_dqn_loop_with_sharded_rollouts():
    """rollouts='sharded': each rank steps its own simulator (25 of the 50 transitions per train step), samples its 16 of
    the 32 graphs of the minibatch from its own memory; gradients are all-reduced with the global Huber denominator.
    Different experience per rank, identical replicas, the same per-step statistics on both ranks, and the target
    network still syncs after 500 collected transitions (= 10 train steps, BS_brain.py:846-847)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_worker_sharded, args=(2, port, ret), nprocs=2, join=True)
    (w0, wt0, l0, q0, m0, ns0, nt0), (w1, wt1, l1, q1, m1, ns1, nt1) = ret[0], ret[1]
    assert nt0 == nt1 == 25 and ns0 == ns1 == 250 and m0.shape == m1.shape == (250, 68)
    assert not np.array_equal(m0, m1)                         # the ranks really explored different trajectories
    assert np.array_equal(w0, w1) and np.array_equal(wt0, wt1)        # replicas bit-identical
    assert np.array_equal(w0, wt0)                            # target synced at the 10th step (2 x 250 = 500 transitions)
    assert np.allclose(l0, l1, rtol=1e-12, atol=0) and np.allclose(q0, q1, rtol=1e-12, atol=0)   # all-reduced statistics
    assert np.all(np.isfinite(l0)) and not np.allclose(w0, _initial_weights())
