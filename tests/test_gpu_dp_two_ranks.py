"""GPU: the data-parallel step with the REAL engine on two ranks.  The box has one GPU, so both ranks share cuda:0 and
the process group is gloo (it all-reduces CUDA tensors through the host; RCCL refuses two ranks on one device) -- the
collective differs from production, everything else (sharding, global Huber denominator, in-place all-reduce on the
aliased gradient buffer, Adam on every rank, hipGraph replay) is the production path."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, use_graph, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from v2xgnn import GnnSpec, PackedBatch, GnnEngine
        from v2xgnn.dp import DataParallelTrainer
        from oracle import compact as oc
        from util import f32_params, random_inputs
        spec = GnnSpec(n_nodes=20, feat_dim=64)
        rng = np.random.default_rng(4)
        P = f32_params(spec, rng)
        x, e, adj = random_inputs(rng, 256, 20)
        y = rng.normal(2.5, 1.0, size=(256 * 20, 4)).astype(np.float32)
        pb = PackedBatch.from_dense(x, e, adj)
        eng = GnnEngine(spec, use_graph=use_graph)
        eng.set_weights(oc.params_to_list(P))
        tr = DataParallelTrainer(eng, overlap=use_graph)        # the graph-replayed case also runs the two-phase (overlapped) step
        sb, sy = tr.shard(pb, y)
        db, yd = eng.to_device(sb), torch.from_numpy(np.ascontiguousarray(sy)).cuda()
        torch.cuda.synchronize()
        losses = []
        with torch.cuda.stream(torch.cuda.Stream()):
            for _ in range(3):
                losses.append(tr.train_step(db, yd, n_graphs_global=pb.n_graphs).cpu().numpy())
            torch.cuda.synchronize()
        ret[rank] = (eng.get_flat(), np.stack(losses))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("use_graph", [False, True])
def test_two_rank_engine_step_equals_single_process_step(use_graph):
    import torch.multiprocessing as mp
    from v2xgnn import GnnSpec, PackedBatch, GnnEngine
    from oracle import compact as oc
    from util import f32_params, random_inputs
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, port, use_graph, ret), nprocs=2, join=True)
    spec = GnnSpec(n_nodes=20, feat_dim=64)
    rng = np.random.default_rng(4)
    P = f32_params(spec, rng)
    x, e, adj = random_inputs(rng, 256, 20)
    y = rng.normal(2.5, 1.0, size=(256 * 20, 4)).astype(np.float32)
    pb = PackedBatch.from_dense(x, e, adj)
    eng = GnnEngine(spec)
    eng.set_weights(oc.params_to_list(P))
    losses = np.stack([eng.train_step(pb, y) for _ in range(3)])
    w = eng.get_flat()
    assert np.array_equal(ret[0][0], ret[1][0])                     # replicas stay bit-identical
    for r in (0, 1):
        assert np.allclose(ret[r][1], losses, rtol=2e-5, atol=1e-7)
        assert np.allclose(ret[r][0], w, rtol=2e-4, atol=2e-6)        # fp32 summation order of the two half-batch gradients


def _dqn_worker(rank, world, port, ret):
    import random
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from v2xgnn.rl import Agent, RL_Config
        from v2xgnn.rl.train import start_env
        random.seed(9)
        np.random.seed(9)
        cfg = RL_Config()
        cfg.set_train_value(64, 0.5, 64, 1, 0.1)
        env = start_env(20)
        agent = Agent(20, env.n_RB, env.n_Neighbor, 64, env, cfg, seed=2, data_parallel=world > 1)
        assert agent.device_replay is not None
        loss, reward_step, _, q_mean, _, _, _ = agent.train(1, 3)
        ret[rank] = (np.concatenate([w.ravel() for w in agent.brain.model.get_weights()]), loss, reward_step, q_mean)
    finally:
        if world > 1:
            dist.destroy_process_group()


def test_two_rank_dqn_loop_with_device_replay_on_the_engine():
    """BASELINE config 3's scheme on the real engine: two ranks (sharing the one GPU, gloo collective) run the same
    seeded simulator, keep their replay memories in HBM, draw the same indices and fit their halves of every minibatch.
    Replicas must stay bit-identical and agree with the single-process loop."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_dqn_worker, args=(2, port, ret), nprocs=2, join=True)
    single = mp.Manager().dict()
    mp.spawn(_dqn_worker, args=(1, port, single), nprocs=1, join=True)
    w1, loss1, rew1, qm1 = single[0]
    assert np.array_equal(ret[0][0], ret[1][0])                     # replicas bit-identical
    for r in (0, 1):
        w, loss, rew, qm = ret[r]
        assert np.array_equal(rew, rew1)                            # identical rollouts (same weights -> same actions)
        assert np.allclose(loss, loss1, rtol=2e-4, atol=1e-6)
        assert np.allclose(qm, qm1, rtol=1e-3, atol=1e-4)          # (3 fit steps of fp32 rounding differences: the shard sums differ)
        assert np.allclose(w, w1, rtol=1e-3, atol=2e-5)


BASELINE_METRIC = "graph-instances/sec (fwd+bwd), 20-V2V-link graphs, batch 4096"


@pytest.mark.parametrize("scaling,batch,global_batch,per_gpu", [("weak", 512, 1024, 512), ("strong", 512, 512, 256),
                                                                  (None, 4096, 4096, 2048), ("weak", 4096, 8192, 4096)])
def test_bench_two_ranks_on_one_gpu(scaling, batch, global_batch, per_gpu):
    """bench.py launched exactly as the driver launches it for N = 2 (torch.distributed.run, one rank per process:
    barrier, max-over-ranks timing, whole-job value, rank-0 JSON line); the box has one GPU, so both ranks use cuda:0
    and the collective is gloo (V2X_BENCH_ONE_DEVICE / V2X_BENCH_BACKEND exist for this test only).  weak: --batch graphs
    per GPU; strong (the DEFAULT, which is what the driver's `bench.py --gpus N` gets): the global batch cut
    into two shards.  BASELINE.json's metric string appears only when the GLOBAL batch is 4096 -- not for 4096 per GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, V2X_BENCH_ONE_DEVICE="1", V2X_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29551", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--min-seconds", "0",
           "--batch", str(batch), "--no-cpu-baseline", "--no-roofline", "--no-weak-pass"] + (["--scaling", scaling] if scaling else [])
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                        # rank 0 only
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == global_batch and res["config"]["parallelism"] == "dp2"
    assert res["config"]["graphs_per_gpu"] == per_gpu
    assert res["value"] > 0 and res["scaling"] == (scaling or "strong") == res["config"]["scaling"] and res["steps"] == 5
    assert (res["metric"] == BASELINE_METRIC) == (global_batch == 4096)
    assert res["config"]["aggregation"] in ("complement", "edge-bitset-walk", "edge-gather")
    assert abs(res["value"] - global_batch * 5 / (res["ms_per_step"] * 5e-3)) / res["value"] < 1e-3
    assert res["config"]["launch"] in ("eager", "hipGraph replay") and set(res["config"]["launch_probe_ms"]) == {"hipGraph replay", "eager"}


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher around it: bench.py re-executes itself through
    torch.distributed.run with two ranks (both on the box's one GPU here, gloo collective), and the one line carries BOTH
    regimes -- the strong-scaling value of the metric (global batch fixed) and a second timed pass at --batch graphs per GPU
    (config.weak) -- plus the ranks the process group really had."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(V2X_BENCH_ONE_DEVICE="1", V2X_BENCH_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--min-seconds", "0",
           "--batch", "512", "--no-roofline"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["ranks_seen"] == 2 and res["config"]["collective_backend"] == "gloo"
    assert res["scaling"] == "strong" and res["config"]["global_batch"] == 512 and res["config"]["graphs_per_gpu"] == 256
    weak = res["config"]["weak"]
    assert weak["graphs_per_gpu"] == 512 and weak["global_batch"] == 1024 and weak["value"] > 0
    assert res["aggregation"] == "edge-bitset-walk" and res["fast_path"]["aggregation"] == "complement" and res["fast_path"]["value"] > 0


def _wide_worker(rank, world, port, mode, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from v2xgnn import GnnSpec, PackedBatch, GnnEngine
        from v2xgnn.dp import DataParallelTrainer
        from oracle import compact as oc
        from util import f32_params, random_inputs
        N, F, L, B = 12, 128, 2, 64
        spec = GnnSpec(n_nodes=N, feat_dim=F, n_mp_layers=L)
        rng = np.random.default_rng(5)
        P = f32_params(spec, rng)
        x, e, adj = random_inputs(rng, B, N)
        y = rng.normal(2.5, 1.0, size=(B * N, 4)).astype(np.float32)
        pb = PackedBatch.from_dense(x, e, adj)
        with torch.cuda.stream(torch.cuda.Stream()):
            eng = GnnEngine(spec, use_graph=(mode != "plain"))
            eng.set_weights(oc.params_to_list(P))
            tr = DataParallelTrainer(eng, overlap=(mode == "overlap"), shard_optimizer=(mode == "shard"))
            sb, sy = tr.shard(pb, y)
            db, yd = eng.to_device(sb), torch.from_numpy(np.ascontiguousarray(sy)).cuda()
            torch.cuda.synchronize()
            losses, grads = [], []
            for _ in range(3):
                losses.append(tr.train_step(db, yd, n_graphs_global=pb.n_graphs).cpu().numpy())
                torch.cuda.synchronize()
                grads.append(eng.get_grad_flat())
            m, v, it = tr.gather_optimizer_state()
        ret[rank] = (eng.get_flat(), np.stack(losses), np.stack(grads), m, v, it, eng.grad_buckets())
    finally:
        dist.destroy_process_group()


def test_two_rank_wide_model_per_layer_buckets_and_sharded_optimizer():
    """VERDICT r03 item 1c on the real engine (feat_dim 128, two ranks on the box's one GPU, gloo): the per-layer phases of
    v2x_forward_backward_phase with one asynchronous all-reduce per bucket, and the sharded optimizer step (reduce-scatter
    emulated by gloo's all-reduce, v2x_apply_gradients_range on the rank's half of every bucket, all-gather of the parameters
    into the aliased parameter buffer), against the one-all-reduce step.  Per-layer launches and the merged single-GPU launch
    sum rows in the same order, so the bucketed gradients are bit-identical to the single all-reduce's."""
    import torch.multiprocessing as mp
    runs = {}
    for mode in ("plain", "overlap", "shard"):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        ret = mp.Manager().dict()
        mp.spawn(_wide_worker, args=(2, port, mode, ret), nprocs=2, join=True)
        runs[mode] = dict(ret)
    buckets = runs["plain"][0][6]
    assert len(buckets) == 4 and buckets[3][0] == 0 and buckets[0][0] + buckets[0][1] == runs["plain"][0][0].size
    for r in (0, 1):
        w0, l0, g0 = runs["plain"][r][0], runs["plain"][r][1], runs["plain"][r][2]
        assert np.all(np.isfinite(w0))
        for mode in ("overlap", "shard"):
            w, l, g = runs[mode][r][0], runs[mode][r][1], runs[mode][r][2]
            assert np.allclose(l, l0, rtol=1e-6, atol=1e-8), mode
            if mode == "overlap":
                assert np.array_equal(g[0], g0[0]), "first step's bucketed gradients differ from the single all-reduce's"
            # (later steps: the sharded / bucketed Adam and the single launch may contract their multiply-adds differently)
            assert np.abs(w - w0).max() <= 2e-5, (mode, np.abs(w - w0).max())
        m0, v0, it0 = runs["plain"][r][3], runs["plain"][r][4], runs["plain"][r][5]
        ms, vs, its = runs["shard"][r][3], runs["shard"][r][4], runs["shard"][r][5]
        assert it0 == its == 3
        assert np.abs(ms - m0).max() <= 1e-4 * np.abs(m0).max() and np.abs(vs - v0).max() <= 1e-4 * np.abs(v0).max()
    for mode in ("plain", "overlap", "shard"):
        assert np.array_equal(runs[mode][0][0], runs[mode][1][0]), mode + ": replicas diverged"
