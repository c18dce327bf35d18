"""GPU: whole-model parity (forward, per-output loss, every gradient array) against the float64 oracle over a spread of
shapes -- link counts 1..40 (1 and 2 links: graphs without edges), all feature widths incl. the wide path, 1..4
message-passing layers, per-node and shared weights, batches that are not multiples of any tile (1, 17, 130), the
reference topology and random adjacencies.  `tests/sweep_shapes.py` runs the full 504-shape grid.

Every shape is ONE seeded draw and that draw counts.  A ReLU whose pre-activation lies within fp32 rounding of 0 is gated
differently by fp32 and fp64 arithmetic (one row's contribution appears / disappears from a gradient summed over few
rows): such units are found explicitly from the oracle's pre-activations and the oracle's reverse pass takes their gates
the kernels' way (tests/util.py, assert_grads_match_oracle); nothing is redrawn, no tolerance is widened."""
import os

import numpy as np
import pytest

import v2xgnn
from v2xgnn import GnnSpec, PackedBatch, GnnEngine
from oracle import compact as oc
from util import (ospec, f32_params, random_inputs, oracle_step, assert_fwd_close, assert_close, assert_grads_match_oracle,
                  FWD_RTOL, FWD_ATOL)

pytestmark = pytest.mark.gpu

SHAPES = [  # N, F, L, shared, B, reference topology
    (1, 16, 2, False, 17, False), (1, 64, 1, True, 130, False), (2, 32, 2, False, 130, False), (2, 128, 1, True, 1, False),
    (3, 16, 4, False, 17, True), (3, 64, 2, True, 130, True), (7, 32, 1, False, 130, False), (7, 128, 2, False, 17, True),
    (7, 16, 4, True, 1, False), (20, 16, 4, False, 130, False), (20, 32, 2, True, 17, True), (20, 64, 4, False, 1, True),
    (20, 128, 4, True, 17, False), (33, 16, 1, False, 130, False), (33, 32, 1, False, 130, True), (33, 64, 4, True, 130, False),
    (33, 128, 2, False, 1, True), (40, 16, 2, True, 130, False), (40, 32, 1, False, 130, False), (40, 64, 2, True, 130, False),
    (40, 64, 2, False, 17, True), (40, 128, 1, False, 17, False), (40, 32, 4, False, 17, True), (33, 64, 2, False, 17, False),
]


def _check(N, F, L, shared, B, topo, seed):
    rng = np.random.default_rng(seed)
    spec = GnnSpec(n_nodes=N, feat_dim=F, n_mp_layers=L, share_weights=shared)
    P = f32_params(spec, rng)
    x, e, adj = random_inputs(rng, B, N, ref_topology=topo and N > 2)
    pb = PackedBatch.from_dense(x, e, adj)
    # (the loss below is differentiated at the q of `forward`: keep `forward` on the training path's kernels -- the
    #  few-graph predict kernel has its own summation order and its own test, test_small_predict_*)
    os.environ["V2X_SMALL_PREDICT"] = "0"
    try:
        eng = GnnEngine(spec)
    finally:
        del os.environ["V2X_SMALL_PREDICT"]
    eng.set_weights(oc.params_to_list(P))
    graph = ((np.arange(B + 1) * N).astype(np.int32), pb.row_ptr, pb.col_idx)
    q = eng.forward(pb)
    # targets around the kernels' own q, so that both Huber branches occur whatever the scale of q; the loss is
    # differentiated at that q (forward parity is asserted separately)
    y = (q + rng.normal(0, 1.2, size=q.shape)).astype(np.float32)
    step = oracle_step(spec, P, x.reshape(B * N, -1), e.reshape(B * N, -1), graph, y, q_at=q)
    assert_fwd_close(q, step['q'], "forward")
    loss = eng.forward_backward(pb, y)
    assert_close(loss, step['loss'], 2e-4, 1e-6, "per-output Huber loss")
    assert_grads_match_oracle(v2xgnn.flat_to_keras_list(spec, eng.get_grad_flat()), P, step, "N=%d F=%d L=%d B=%d" % (N, F, L, B))
    eng.close()


@pytest.mark.parametrize("N,F,L,shared,B,topo", SHAPES)
def test_model_parity_over_shapes(N, F, L, shared, B, topo):
    _check(N, F, L, shared, B, topo, seed=7 * N + F + L + B)


GROUPS = [  # per-node weights, whole 16-graph groups, F <= 64, L >= 1: the training step hands h_L, a_L and gha over
            # FRAGMENT-major (frag_layout in csrc/v2xgnn.hip); with the reference topology also the complement aggregation,
            # whose backward takes its masks from the forward
    (4, 16, 2, False, 32, True), (4, 16, 1, False, 16, False), (7, 32, 1, False, 16, False), (7, 32, 2, False, 48, True),
    (20, 64, 2, False, 48, True), (20, 64, 1, False, 16, True), (20, 64, 4, False, 16, False), (20, 16, 3, False, 32, True),
    (28, 32, 2, False, 16, True), (32, 16, 2, False, 16, False),
]


@pytest.mark.parametrize("N,F,L,shared,B,topo", GROUPS)
def test_model_parity_whole_groups(N, F, L, shared, B, topo):
    _check(N, F, L, shared, B, topo, seed=7 * N + F + L + B)


SMALL = [  # N, F, L, shared, B, reference topology: forwards of at most 256 node rows run k_predict_small
    (1, 16, 1, False, 1, False), (2, 32, 2, False, 3, False), (4, 16, 2, False, 1, True), (4, 16, 2, True, 64, True),
    (7, 32, 3, False, 5, False), (20, 64, 2, False, 1, True), (20, 64, 2, False, 12, True), (20, 64, 4, True, 2, True),
    (20, 32, 2, False, 10, False), (32, 64, 1, False, 8, True), (28, 16, 4, False, 9, False), (20, 64, 2, False, 13, True),
    (40, 32, 2, False, 1, True), (33, 64, 1, True, 2, False),      # more than 32 links: not eligible, training-path kernels
]


@pytest.mark.parametrize("N,F,L,shared,B,topo", SMALL)
def test_small_predict_vs_oracle_and_training_path(N, F, L, shared, B, topo):
    """The one-launch few-graph forward (csrc/kernels_small.hpp: workgroup per node, grid barriers between the layers)
    against the float64 oracle and against the training path's forward; repeated calls (the barrier counters must clean
    up after themselves) and a batch just above the row limit (13 x 20 = 260 rows: training-path kernels)."""
    rng = np.random.default_rng(7 * N + F + L + B)
    spec = GnnSpec(n_nodes=N, feat_dim=F, n_mp_layers=L, share_weights=shared)
    P = f32_params(spec, rng)
    x, e, adj = random_inputs(rng, B, N, ref_topology=topo and N > 2)
    pb = PackedBatch.from_dense(x, e, adj)
    small = GnnEngine(spec)
    os.environ["V2X_SMALL_PREDICT"] = "0"
    try:
        plain = GnnEngine(spec)
    finally:
        del os.environ["V2X_SMALL_PREDICT"]
    for eng in (small, plain):
        eng.set_weights(oc.params_to_list(P))
    M = oc.csr_to_matrix((np.arange(B + 1) * N).astype(np.int32), pb.row_ptr, pb.col_idx, dtype=np.float64)
    q_ref, _ = oc.forward(ospec(spec), P, x.reshape(B * N, -1).astype(np.float64), e.reshape(B * N, -1).astype(np.float64), M)
    scale = max(1.0, np.abs(q_ref).max())
    qp = plain.forward(pb)
    for rep in range(3):
        q = small.forward(pb)
        assert np.all(np.abs(q - q_ref) <= FWD_RTOL * np.abs(q_ref) + FWD_ATOL * scale), "call %d vs oracle" % rep
        assert np.all(np.abs(q - qp) <= FWD_RTOL * np.abs(qp) + FWD_ATOL * scale), "call %d vs training path" % rep
    if B * N > 256 or N > 32:
        assert np.array_equal(q, qp)                      # not eligible: both engines run the same kernels
    # a fit step after a small predict still works (the predict saved nothing for a backward pass) and moves the weights
    y = (q_ref + rng.normal(0, 1.0, size=q_ref.shape)).astype(np.float32)
    w0 = small.get_flat().copy()
    small.train_step(pb, y)
    plain.train_step(pb, y)
    assert not np.array_equal(small.get_flat(), w0)
    assert np.array_equal(small.get_flat(), plain.get_flat())
    q2, qp2 = small.forward(pb), plain.forward(pb)
    assert np.all(np.abs(q2 - qp2) <= FWD_RTOL * np.abs(qp2) + FWD_ATOL * max(1.0, np.abs(qp2).max()))
    small.close()
    plain.close()


def test_small_predict_soak_bitwise_repeatable_across_launches_models_and_batch_sizes():
    """The one-launch predict exchanges rows as data-tagged words whose tag comes from a per-graph departure counter
    (csrc/kernels_small.hpp): 3,000 predicts of 1..12 graphs on two models sharing the GPU, eager and as replayed hipGraphs,
    interleaved with fit steps and weight copies.  Between two weight changes every repeat of a forward must return the
    SAME BITS as the first one -- a row taken from an earlier launch or stage would show."""
    import torch
    N, F = 20, 64
    spec = GnnSpec(n_nodes=N, feat_dim=F)
    rng = np.random.default_rng(0)
    with torch.cuda.stream(torch.cuda.Stream()):
        a, b = GnnEngine(spec, use_graph=True), GnnEngine(spec)
        w = oc.params_to_list(f32_params(spec, rng))
        a.set_weights(w)
        b.set_weights(w)
        batches = {}
        for B in range(1, 13):
            x, e, adj = random_inputs(rng, B, N, ref_topology=True)
            batches[B] = PackedBatch.from_dense(x, e, adj)
        xt, et, at = random_inputs(rng, 64, N, ref_topology=True)
        train = a.to_device(PackedBatch.from_dense(xt, et, at))
        yt = torch.from_numpy(rng.normal(2.5, 1.0, size=(64 * N, 4)).astype(np.float32)).cuda()
        ref = {}
        for it in range(3000):
            B = 1 + it % 12
            eng = a if it % 3 else b
            q = eng.forward(batches[B])
            key = (id(eng), B)
            if key in ref:
                assert np.array_equal(q, ref[key]), ("predict changed without a weight change", it, B)
            else:
                assert np.all(np.isfinite(q))
                ref[key] = q.copy()
            if it % 50 == 0:
                a.train_step(train, yt, want_loss=False)
                ref = {k: v for k, v in ref.items() if k[0] != id(a)}
            if it % 500 == 0:
                b.copy_weights_from(a)
                ref = {k: v for k, v in ref.items() if k[0] != id(b)}
        torch.cuda.synchronize()
        a.close()
        b.close()


def _exchange_counters(eng, n):
    """torch view (no copy) of the first n 64-bit departure counters of the one-launch predict's exchange."""
    import torch
    ptr = int(eng._lib.v2x_debug_exchange_counters(eng._h))

    class _H(object):
        pass
    h = _H()
    h.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (ptr, False), "version": 2, "strides": None}
    return torch.as_tensor(h, device="cuda:%d" % eng.device)


def test_small_predict_out_of_step_exchange_times_out_instead_of_hanging():
    """VERDICT r03 item 8 / ADVICE: the tagged-word polls of k_predict_small were unbounded.  Here the departure counter of a
    graph is bumped by a multiple of N from another stream WHILE predicts run, so workgroups of one launch read different
    epochs and wait for tags nobody writes.  The launch must terminate (bounded polls), the call must report V2X_ESTATE, the
    library must re-arm the exchange by itself, and the next predicts must be right again.  Also: a device batch with a source
    id outside its graph is reported (V2X_EINVAL) instead of polling a foreign row, and v2x_reset_exchange works on request."""
    import time
    import torch
    import v2xgnn
    N, F, B = 20, 64, 4
    spec = GnnSpec(n_nodes=N, feat_dim=F)
    rng = np.random.default_rng(3)
    eng = GnnEngine(spec)
    eng.set_weights(oc.params_to_list(f32_params(spec, rng)))
    x, e, adj = random_inputs(rng, B, N, ref_topology=True)
    pb = PackedBatch.from_dense(x, e, adj)
    q0 = eng.forward(pb)
    assert eng.path_info(pb)["graph_layers"].startswith("fused")
    cnt = _exchange_counters(eng, 256)
    torch.cuda.synchronize()
    assert int(cnt[0].item()) == N and int(cnt[B - 1].item()) == N and int(cnt[B].item()) == 0     # one launch: N departures per graph
    # two NON-default streams: the legacy default stream would serialise the predict behind the bumps
    side, main = torch.cuda.Stream(), torch.cuda.Stream()
    seen = 0
    t_max = 0.0
    # 2,000 bumps of whole epochs as ONE replayable graph: they run back to back on the GPU (~5 ms), not at the pace the
    # host can enqueue them (issued one by one they are long done when the predict starts)
    bumps = torch.cuda.CUDAGraph()
    with torch.cuda.graph(bumps, stream=side):
        for _ in range(2000):
            cnt[:B].add_(7 * N)
    for attempt in range(100):
        with torch.cuda.stream(side):
            bumps.replay()
        t0 = time.perf_counter()
        try:
            with torch.cuda.stream(main):
                q = eng.forward(pb)
            assert q.shape == q0.shape
        except v2xgnn.lib.V2XError as exc:
            assert "timed out" in str(exc)
            seen += 1
        t_max = max(t_max, time.perf_counter() - t0)
        torch.cuda.synchronize()
        if seen >= 2:
            break
    assert seen >= 1, "100 predicts raced against 200,000 counter bumps and none saw an out-of-step exchange"
    assert t_max < 20.0, "a timed-out predict took %.1f s" % t_max
    torch.cuda.synchronize()
    # the library re-armed the exchange when it reported the time-out: predicts are right again, bit for bit
    for _ in range(5):
        assert np.array_equal(eng.forward(pb), q0)
    eng.reset_exchange()
    torch.cuda.synchronize()
    assert int(cnt[0].item()) == 0
    assert np.array_equal(eng.forward(pb), q0)
    # a device batch with a source id outside its graph: reported, no hang
    db = eng.to_device(pb)
    db.col_idx[5] = N + 3
    out = eng.forward(db)
    with pytest.raises(ValueError, match="source id outside its graph"):
        eng.check_errors()
    assert out.shape == (B * N, 4)
    assert np.array_equal(eng.forward(pb), q0)
    eng.close()
