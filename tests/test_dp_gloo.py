"""CPU, world_size 2, gloo: the data-parallel step (shard -> local forward/backward with the GLOBAL
Huber denominator -> ONE all-reduce of the flat gradient -> Adam on every rank) equals the
single-process step on the full batch.  The compute backend here is the oracle (tests only);
on the GPU box the backend is GnnEngine and the process group is RCCL."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from v2xgnn import GnnSpec, PackedBatch
from v2xgnn.dp import DataParallelTrainer
from oracle import compact as oc
from oracle.keras_semantics import KerasAdam
from util import ospec, random_inputs


class OracleBackend(object):
    """Implements the backend protocol of DataParallelTrainer with the CPU oracle."""

    def __init__(self, spec, params):
        self.spec, self.os = spec, ospec(spec)
        self.params = oc.cast_params(params, np.float64)
        self.opt = KerasAdam()
        n = sum(a.size for a in oc.param_arrays(self.params))
        self._g = torch.zeros(n, dtype=torch.float64)

    def forward_backward(self, batch, y, n_global=None, want_loss=True):
        graph = ((np.arange(batch.n_graphs + 1) * batch.n_nodes).astype(np.int32), batch.row_ptr, batch.col_idx)
        M = oc.csr_to_matrix(*graph, dtype=np.float64)
        x, e = batch.xe[:, :9].astype(np.float64), batch.xe[:, 9:13].astype(np.float64)
        q, cache = oc.forward(self.os, self.params, x, e, M)
        loss, dq = oc.huber_loss_and_grad(self.os, q, np.asarray(y, np.float64), n_global)
        g = oc.backward(self.os, self.params, cache, dq)
        self._g.copy_(torch.from_numpy(np.concatenate([a.ravel() for a in oc.param_arrays(g)])))
        return loss

    def grad_tensor(self):
        return self._g

    def apply_gradients(self):
        flat = self._g.numpy()
        grads, pos = [], 0
        for a in oc.param_arrays(self.params):
            grads.append(flat[pos:pos + a.size].reshape(a.shape))
            pos += a.size
        self.opt.step(oc.param_arrays(self.params), grads)


def _data(spec, B):
    rng = np.random.default_rng(12)
    P = oc.init_params(ospec(spec), rng, random_bias=True)
    x, e, adj = random_inputs(rng, B, spec.n_nodes)
    y = rng.normal(2.5, 1.0, size=(B * spec.n_nodes, 4))
    return P, PackedBatch.from_dense(x, e, adj), y


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
