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
