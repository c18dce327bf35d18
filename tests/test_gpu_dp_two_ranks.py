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
        tr = DataParallelTrainer(eng)
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
