"""GPU, BASELINE.json's full size (20 links, feat_dim 64, 2 layers, batch 4096; the oracle would need minutes here):
size-independent properties of the hot path instead of an element-wise oracle comparison."""
import numpy as np
import pytest

import v2xgnn
from v2xgnn import GnnSpec, PackedBatch, GnnEngine
from util import assert_close, assert_grad_close

pytestmark = pytest.mark.gpu
N, F, B = 20, 64, 4096


@pytest.fixture(scope="module")
def setup():
    import bench
    rng = np.random.default_rng(2024)
    x, e, adj, y = bench.synth_batch(rng, B, N)
    spec = GnnSpec(n_nodes=N, feat_dim=F)
    shapes = v2xgnn.keras_list_shapes(spec)
    w = [rng.normal(0, 0.05, size=s).astype(np.float32) if len(s) == 1 else
         rng.uniform(-np.sqrt(6.0 / sum(s)), np.sqrt(6.0 / sum(s)), size=s).astype(np.float32) for s in shapes]
    return spec, w, x, e, adj, y


def _engine(spec, w, **kw):
    eng = GnnEngine(spec, **kw)
    eng.set_weights(w)
    return eng


def test_loss_is_the_huber_mean_of_the_forward_output(setup):
    spec, w, x, e, adj, y = setup
    eng = _engine(spec, w)
    pb = PackedBatch.from_dense(x, e, adj)
    q = eng.forward(pb).astype(np.float64)
    assert q.shape == (B * N, 4) and np.all(np.isfinite(q))
    ab = np.abs(q - y)
    quad = np.minimum(ab, 1.0)
    ref = (0.5 * quad * quad + (ab - quad)).reshape(B, N, 4).mean(axis=(0, 2))       # per output, over (B, C)
    loss = eng.forward_backward(pb, y)
    assert_close(loss, ref, 2e-5, 1e-7, "per-output Huber means at batch 4096")


def test_gradient_is_additive_over_graphs_and_invariant_to_their_order(setup):
    """d(loss)/dw of the batch == sum of the gradients of its two halves taken with the GLOBAL denominator (what the
    data-parallel all-reduce relies on), and does not depend on the order of the graphs in the batch."""
    spec, w, x, e, adj, y = setup
    eng = _engine(spec, w)
    pb = PackedBatch.from_dense(x, e, adj)
    eng.forward_backward(pb, y)
    g_full = eng.get_grad_flat().astype(np.float64)
    acc = np.zeros_like(g_full)
    yb = y.reshape(B, N, 4)
    for r in range(2):
        sh = pb.shard(r, 2)
        eng.forward_backward(sh, yb[r * B // 2:(r + 1) * B // 2].reshape(-1, 4), n_global=B)
        acc += eng.get_grad_flat()
    assert_grad_close(acc, g_full, "sum of half-batch gradients")
    perm = np.random.default_rng(3).permutation(B)
    pbp = PackedBatch.from_dense(x[perm], e[perm], adj[perm])
    q = eng.forward(pb).reshape(B, N, 4)
    qp = eng.forward(pbp).reshape(B, N, 4)
    assert np.array_equal(qp, q[perm])                          # a graph's output does not depend on its neighbours in the batch
    eng.forward_backward(pbp, yb[perm].reshape(-1, 4))
    assert_grad_close(eng.get_grad_flat(), g_full, "gradient of the permuted batch")


def test_graph_replay_is_bitwise_eager_and_steps_are_reproducible(setup):
    spec, w, x, e, adj, y = setup
    outs = []
    for use_graph in (False, True, True):
        eng = _engine(spec, w, use_graph=use_graph)
        pb = PackedBatch.from_dense(x, e, adj)
        for _ in range(3):
            loss = eng.train_step(pb, y)
        outs.append((np.asarray(loss), eng.get_flat()))
    for loss, flat in outs[1:]:
        assert np.array_equal(loss, outs[0][0]) and np.array_equal(flat, outs[0][1])
    assert np.all(np.isfinite(outs[0][1])) and not np.array_equal(outs[0][1], v2xgnn.keras_list_to_flat(spec, w))
