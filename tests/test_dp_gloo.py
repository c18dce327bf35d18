"""CPU, world_size 2, gloo: the data-parallel step (shard -> local forward/backward with the GLOBAL
Huber denominator -> ONE all-reduce of the flat gradient -> Adam on every rank) equals the
single-process step on the full batch.  The compute backend here is the oracle (tests only);
on the GPU box the backend is GnnEngine and the process group is RCCL."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

import v2xgnn
from v2xgnn import GnnSpec, PackedBatch
from v2xgnn.dp import DataParallelTrainer
from oracle import compact as oc
from oracle_engine import OracleEngine
from util import ospec, random_inputs


def OracleBackend(spec, params):
    return OracleEngine(spec, params)


def _data(spec, B):
    rng = np.random.default_rng(12)
    P = oc.init_params(ospec(spec), rng, random_bias=True)
    x, e, adj = random_inputs(rng, B, spec.n_nodes)
    y = rng.normal(2.5, 1.0, size=(B * spec.n_nodes, 4))
    return P, PackedBatch.from_dense(x, e, adj), y


def _ragged_data():
    """Variable-size graphs (BASELINE configs[4] in miniature): shared weights, ONE Huber mean over all node rows."""
    import bench
    rng = np.random.default_rng(21)
    sizes, offs, row_ptr, col_idx, x, e, y = bench.synth_ragged(rng, 11, 3, 12)
    spec = GnnSpec(n_nodes=1, feat_dim=16, share_weights=True, variable_graphs=True)
    P = oc.init_params(ospec(spec), rng, random_bias=True)
    pb = PackedBatch(len(sizes), 0, v2xgnn.pack_xe(x, e), row_ptr, col_idx, graph_off=offs)
    return spec, P, pb, y.astype(np.float64)


def _worker_ragged(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        spec, P, pb, y = _ragged_data()
        tr = DataParallelTrainer(OracleBackend(spec, P))
        sb, sy = tr.shard(pb, y)
        losses = []
        for _ in range(2):
            # the Huber mean is over the GLOBAL row count, so shard gradients (and losses) SUM to the full batch's
            losses.append(np.asarray(tr.train_step(sb, sy, n_graphs_global=pb.n_rows)))
        flat = np.concatenate([a.ravel() for a in oc.param_arrays(tr.backend.params)])
        ret[rank] = (flat, np.stack(losses), sb.n_graphs, sb.n_rows, sb.n_edges)
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        spec = GnnSpec(n_nodes=5, feat_dim=16)
        P, pb, y = _data(spec, 8)
        tr = DataParallelTrainer(OracleBackend(spec, P))
        assert (tr.rank, tr.world) == (rank, world)
        losses = []
        for _ in range(3):
            sb, sy = tr.shard(pb, y)
            assert sb.n_graphs == 4
            losses.append(np.asarray(tr.train_step(sb, sy, n_graphs_global=pb.n_graphs)))
        flat = np.concatenate([a.ravel() for a in oc.param_arrays(tr.backend.params)])
        ret[rank] = (flat, np.stack(losses))
    finally:
        dist.destroy_process_group()


def test_two_rank_step_equals_single_process_step():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    spec = GnnSpec(n_nodes=5, feat_dim=16)
    P, pb, y = _data(spec, 8)
    single = OracleBackend(spec, P)
    ref_losses = []
    for _ in range(3):
        ref_losses.append(single.forward_backward(pb, y))
        single.apply_gradients()
    ref = np.concatenate([a.ravel() for a in oc.param_arrays(single.params)])
    for r in range(2):
        flat, losses = ret[r]
        assert np.abs(flat - ref).max() < 1e-12
        assert np.abs(losses - np.stack(ref_losses)).max() < 1e-12
    assert np.array_equal(ret[0][0], ret[1][0])          # replicas stay bit-identical


def test_two_rank_step_on_variable_size_graphs_sharded_by_edges():
    """ragged batches are cut into contiguous shards of whole graphs balanced by edges + nodes; with the
    global row count in the Huber denominator the two-rank step equals the single-process step."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_ragged, args=(2, port, ret), nprocs=2, join=True)
    spec, P, pb, y = _ragged_data()
    single = OracleBackend(spec, P)
    ref_losses = []
    for _ in range(2):
        ref_losses.append(single.forward_backward(pb, y, n_global=pb.n_rows))
        single.apply_gradients()
    ref = np.concatenate([a.ravel() for a in oc.param_arrays(single.params)])
    assert ret[0][2] + ret[1][2] == pb.n_graphs and ret[0][3] + ret[1][3] == pb.n_rows and ret[0][4] + ret[1][4] == pb.n_edges
    for r in range(2):
        flat, losses = ret[r][0], ret[r][1]
        assert np.abs(flat - ref).max() < 1e-12
        assert np.abs(losses - np.stack(ref_losses)).max() < 1e-12
    assert np.array_equal(ret[0][0], ret[1][0])


# ------------------------------------------------------------------------------ per-layer buckets, sharded optimizer step
def _wide_data():
    spec = GnnSpec(n_nodes=3, feat_dim=128, n_mp_layers=2)
    rng = np.random.default_rng(31)
    P = oc.init_params(ospec(spec), rng, random_bias=True)
    x, e, adj = random_inputs(rng, 6, spec.n_nodes)
    y = rng.normal(2.5, 1.0, size=(6 * spec.n_nodes, 4))
    return spec, P, PackedBatch.from_dense(x, e, adj), y


def _worker_buckets(rank, world, port, ret, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        spec, P, pb, y = _wide_data()
        eng = OracleBackend(spec, P)
        tr = DataParallelTrainer(eng, overlap=(mode == "overlap"), shard_optimizer=mode.startswith("shard"))
        sb, sy = tr.shard(pb, y)
        losses, grads = [], []
        for i in range(3):
            # "shard_fallback": in the middle step the backend's phases are not available for the batch (what a host-resident
            # batch is to the GPU engine): the trainer must keep the optimizer step sharded, not run full-range Adam
            eng.phases_on_host = not (mode == "shard_fallback" and i == 1)
            losses.append(np.asarray(tr.train_step(sb, sy, n_graphs_global=pb.n_graphs)))
            grads.append(eng.grad_tensor().numpy().copy())
        flat = np.concatenate([a.ravel() for a in oc.param_arrays(eng.params)])
        m, v, it = tr.gather_optimizer_state()
        ret[rank] = (flat, np.stack(losses), np.stack(grads), m, v, it, [tuple(b) for b in eng.grad_buckets()])
    finally:
        dist.destroy_process_group()


def _run_buckets(mode):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_worker_buckets, args=(2, port, ret, mode), nprocs=2, join=True)
    return ret


def test_two_rank_bucketed_and_sharded_steps_equal_the_single_all_reduce():
    """VERDICT r03 item 1c at feat_dim 128 (one gradient bucket per layer: [Dense], [stage 2], [stage 1], [embed]): the
    phased step with one asynchronous all-reduce per bucket as soon as the bucket is final, and the sharded optimizer step
    (reduce-scatter, Adam on the rank's half of every bucket, all-gather of the parameters), against the one-all-reduce step
    and the single-process step.  The oracle backend poisons (NaN) every bucket it has not released yet, so a collective
    started before its phase would show.  Gradients must be BIT-identical to the single all-reduce's (same two-rank sum)."""
    plain, over, shard, shard_fb = _run_buckets("plain"), _run_buckets("overlap"), _run_buckets("shard"), _run_buckets("shard_fallback")
    spec, P, pb, y = _wide_data()
    single = OracleBackend(spec, P)
    ref_losses = []
    for _ in range(3):
        ref_losses.append(single.forward_backward(pb, y))
        single.apply_gradients()
    ref = np.concatenate([a.ravel() for a in oc.param_arrays(single.params)])
    m_ref, v_ref, it_ref = single.get_optimizer_state()
    buckets = plain[0][6]
    assert len(buckets) == spec.n_mp_layers + 2 and sum(n for _, n in buckets) == single.n_params
    assert buckets[0][0] > buckets[1][0] > buckets[2][0] > buckets[3][0] == 0
    for r in range(2):
        for run in (plain, over, shard, shard_fb):
            flat, losses = run[r][0], run[r][1]
            assert np.all(np.isfinite(flat))
            assert np.abs(flat - ref).max() < 1e-12
            assert np.abs(losses - np.stack(ref_losses)).max() < 1e-12
        assert np.array_equal(over[r][2], plain[r][2])                  # bucketed all-reduces: the same sums, bit for bit
        assert np.array_equal(over[r][0], plain[r][0])
        assert np.array_equal(shard[r][0], plain[r][0])                 # sharded Adam + all-gather: the same weights
        m, v, it = shard[r][3], shard[r][4], shard[r][5]
        assert it == it_ref == 3 and np.abs(m - m_ref).max() < 1e-12 and np.abs(v - v_ref).max() < 1e-14
        # a step without the phases in between two sharded ones: still the sharded optimizer step -- same weights,
        # same gathered moments; a full-range Adam there would have used stale moments outside the rank's slices
        assert np.array_equal(shard_fb[r][0], plain[r][0])
        assert shard_fb[r][5] == 3 and np.abs(shard_fb[r][3] - m_ref).max() < 1e-12 and np.abs(shard_fb[r][4] - v_ref).max() < 1e-14
    assert np.array_equal(shard[0][0], shard[1][0])                     # replicas bit-identical after the all-gather
