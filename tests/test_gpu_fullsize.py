"""GPU, BASELINE.json's sizes.  Element-wise parity with the float64 oracle -- forward, per-output Huber loss, EVERY
gradient array, the weights after one Keras-Adam step -- at configs[1] in full (20 links, feat_dim 64, 2 layers, batch
4096, per-node and shared weights: what `Model.fit` computes at /root/reference/BS_brain.py:147-179,218-223), at the
per-GPU share of configs[3] (100 links x 256 features x 3 layers, 1024 graphs) and of configs[4] (2048 ragged graphs of
8-128 links, shared weights); then size-independent properties of the same path (additivity over shards, order
invariance, hipGraph replay == eager).  The oracle costs seconds at configs[1] and about a minute at configs[3]."""
import os

import numpy as np
import pytest

import v2xgnn
from v2xgnn import GnnSpec, PackedBatch, GnnEngine
from oracle import compact as oc
from oracle.keras_semantics import KerasAdam
from util import MAX_GATE_FLIPS, assert_close, assert_grad_close, assert_fwd_close, assert_grads_match_oracle, f32_params, oracle_step

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


def _training_engine(spec, edge_gather=False):
    """forward() on the training path's kernels: the loss is differentiated at the q of `forward` (the few-graph predict
    kernel has its own summation order and its own tests).  edge_gather: the fused graph layers aggregate with the general
    edge-index gather / segment sum (V2X_FUSED_COMPL=0, read at create) instead of through the complement."""
    os.environ["V2X_SMALL_PREDICT"] = "0"
    if edge_gather:
        os.environ["V2X_FUSED_COMPL"] = "0"
    try:
        return GnnEngine(spec)
    finally:
        del os.environ["V2X_SMALL_PREDICT"]
        os.environ.pop("V2X_FUSED_COMPL", None)


def _parity_with_oracle(spec, P, pb, x, e, graph, what, adam=True, engine=None, one_call_step=False):
    """forward / loss / every gradient array / (one Adam step's weights) of the engine against the float64 oracle on the
    same fp32-rounded inputs and weights.  x, e: node rows [R, 9], [R, 4].  Targets: the engine's own q + N(0, 1.2), so
    that both branches of the Huber loss are taken whatever the magnitude of q (random weights and up to 126-neighbour
    sums put |q| anywhere between 1 and 1e6; bench.py's N(2.5, 1) targets would clip every error)."""
    eng = _training_engine(spec) if engine is None else engine
    w0 = oc.params_to_list(P)
    eng.set_weights(w0)
    q = eng.forward(pb)
    y = (q + np.random.default_rng(99).normal(0, 1.2, size=q.shape)).astype(np.float32)
    n_den = pb.n_rows if spec.variable_graphs else None
    ref = oracle_step(spec, P, x, e, graph, y, q_at=q, n_denominator=n_den)
    assert_fwd_close(q, ref['q'], what + ": forward q")
    loss = eng.forward_backward(pb, y, n_global=n_den)
    assert_close(loss, ref['loss'], 2e-4, 1e-6, what + ": per-output Huber loss")
    got = v2xgnn.flat_to_keras_list(spec, eng.get_grad_flat())
    g_ref, n_cand, n_flip = assert_grads_match_oracle(got, P, ref, what)
    print("%s: every gradient element within tolerance of the oracle's (%d ReLU gates at rounding distance of 0 taken the "
          "kernels' way, of %d candidates)" % (what, n_flip, n_cand))
    assert n_flip <= MAX_GATE_FLIPS, (what, "the checker resolved %d ReLU gates the kernels' way (of %d candidates): too many "
                                            "to be rounding at the gate" % (n_flip, n_cand))
    if adam:
        if one_call_step:
            # the weights after ONE fit call (v2x_train_step) from the same start: on the wide path Adam then runs in the
            # weight-gradient launch's epilogue (WideWgradArgs::adam), with the merged launch's row split, XCD map and grid of
            # THIS batch size -- not the Adam launch that apply_gradients() is
            assert eng.get_optimizer_state()[2] == 0
            eng.set_weights(w0)
            eng.train_step(pb, y, n_global=n_den)
        else:
            eng.apply_gradients()
        params = oc.cast_params(P, np.float64)
        KerasAdam().step(oc.param_arrays(params), oc.param_arrays(g_ref))
        n_loose = n_two_steps = 0
        w_after = eng.get_weights()
        # (1) the update itself, exactly: the weights after the step against Keras Adam (float64) applied to the ENGINE'S OWN
        # gradient -- whatever the sign of a noise-level entry is, the step taken must be the step of the gradient the kernels
        # computed: a wrong-sign, doubled or skipped update of a few weights (e.g. in the row split of the merged weight-gradient
        # launch whose epilogue runs Adam) fails here, where the looser oracle comparison below could let it through
        own = oc.cast_params(P, np.float64)
        KerasAdam().step(oc.param_arrays(own), oc.param_arrays(oc.params_from_list(ref['os'], got, np.float64)))
        for i, (a, b) in enumerate(zip(w_after, oc.params_to_list(own))):
            err = np.abs(a.astype(np.float64) - b)
            assert (err <= 3e-6 + 1e-5 * np.abs(b)).all(), (what, "weights vs Adam of the engine's own gradient", i, err.max())
        for i, (a, b, g, ge) in enumerate(zip(w_after, oc.params_to_list(params), oc.params_to_list(g_ref), got)):
            # (2) against the oracle's weights.  Adam's first step is sign-like (lr_t * m / sqrt(v) = +-1e-3 whatever |g|): where
            # |g| is at rounding-noise level its direction is not determined by fp32 arithmetic -- those entries are compared to
            # within one full step (as in test_train_steps_vs_oracle)
            scale = np.abs(g).max() or 1.0
            noise = 1e-4 * scale
            tight = np.abs(g) > noise
            err = np.abs(a.astype(np.float64) - b)
            assert (err[tight] <= 2e-5 + 2e-4 * np.abs(b[tight])).all(), (what, "weights", i, err[tight].max())
            # a noise-level gradient whose SIGN the two summation orders disagree on moves the weight by lr either way: two full
            # steps apart at most.  Allowed only where BOTH gradients are at noise level and really have opposite signs, and only
            # for a vanishing share of the noise-level entries
            two = (~tight) & (err > 1.1e-3)
            assert (err[~tight] <= 2.1e-3).all(), (what, "weights (sign of g undetermined)", i, err[~tight].max())
            assert (np.abs(ge[two]) <= 2 * noise).all() and (ge[two].astype(np.float64) * g[two] <= 0).all(), \
                (what, "two steps apart without opposite noise-level gradients", i, int(two.sum()))
            n_loose += int((~tight).sum())
            n_two_steps += int(two.sum())
        assert n_two_steps <= max(8, 1e-3 * n_loose), (what, "noise-level gradients with the other sign", n_two_steps, n_loose)
        assert eng.get_optimizer_state()[2] == 1
    eng.close()


@pytest.mark.parametrize("aggregation", ["edge-bitset-walk", "complement"])
@pytest.mark.parametrize("shared", [False, True])
def test_cfg2_full_size_vs_oracle(shared, aggregation):
    """BASELINE configs[1] in full: B = 4096 graphs of 20 links, F = 64, L = 2 -- the 256 lock-stepped fused workgroups,
    the 11 x 24-tile shares of k_mlp_train_wg, the 6 x 704-row chunks and 11-slab sums of k_wgrad / k_reduce_adam -- with
    BOTH aggregation forms of the fused graph layers: the general edge-index gather / segment sum north_star names (the
    form bench.py's `value` runs) and the complement rewriting (the default for these complete-minus-two graphs)."""
    import bench
    rng = np.random.default_rng(2025 + shared)
    x, e, adj, _ = bench.synth_batch(rng, B, N)
    spec = GnnSpec(n_nodes=N, feat_dim=F, share_weights=shared)
    P = f32_params(spec, rng)
    pb = PackedBatch.from_dense(x, e, adj)
    graph = ((np.arange(B + 1) * N).astype(np.int32), pb.row_ptr, pb.col_idx)
    eng = _training_engine(spec, edge_gather=aggregation == "edge-bitset-walk")
    info = eng.path_info(pb)
    assert info["graph_layers"] == "fused" and info["aggregation"] == aggregation, info
    _parity_with_oracle(spec, P, pb, x.reshape(B * N, -1), e.reshape(B * N, -1), graph,
                        "configs[1] B=4096 %s, %s" % ("shared" if shared else "per-node", aggregation), engine=eng)


@pytest.mark.parametrize("share,variant", [(512, "fused(split5)"), (1024, "fused(split4)"), (2048, "fused")])
def test_cfg2_shares_vs_oracle(share, variant):
    """The per-GPU shares of BASELINE configs[1]'s global batch of 4096 at 8 / 4 / 2 GPUs: the 512- and
    1024-graph shares run the split-tile fused graph layers (5 / 4 workgroups per 16-graph tile, stage rows handed over as
    tagged words, csrc/kernels_fused_split.hpp), the 2048-graph share whole-tile workgroups.  Forward, per-output Huber
    loss (mean over the GLOBAL batch is the caller's n_global; here the share's own), every gradient array and the weights
    after one Keras-Adam step against the float64 oracle."""
    import bench
    rng = np.random.default_rng(3000 + share)
    x, e, adj, _ = bench.synth_batch(rng, share, N)
    spec = GnnSpec(n_nodes=N, feat_dim=F)
    P = f32_params(spec, rng)
    pb = PackedBatch.from_dense(x, e, adj)
    graph = ((np.arange(share + 1) * N).astype(np.int32), pb.row_ptr, pb.col_idx)
    eng = _training_engine(spec, edge_gather=True)
    info = eng.path_info(pb)
    assert info["graph_layers"] == variant and info["aggregation"] == "edge-bitset-walk", info
    _parity_with_oracle(spec, P, pb, x.reshape(share * N, -1), e.reshape(share * N, -1), graph,
                        "configs[1] share of %d graphs, %s" % (share, variant), engine=eng)


def test_cfg3_share_vs_oracle():
    """The per-GPU share of BASELINE configs[3]: 1024 graphs x 100 links x 256 features x 3 layers (wide-feature path:
    tiled MFMA GEMMs, MFMA aggregation of dense graphs, in-place weight gradients), incl. the weights after one fit call
    whose Adam runs in the merged weight-gradient launch's epilogue, against Keras Adam on the oracle's gradients."""
    import bench
    n, f, l, b = 100, 256, 3, 1024
    rng = np.random.default_rng(43)
    x, e, adj, _ = bench.synth_batch(rng, b, n)
    spec = GnnSpec(n_nodes=n, feat_dim=f, n_mp_layers=l)
    P = f32_params(spec, rng)
    pb = PackedBatch.from_dense(x, e, adj)
    del adj
    graph = ((np.arange(b + 1) * n).astype(np.int32), pb.row_ptr, pb.col_idx)
    _parity_with_oracle(spec, P, pb, x.reshape(b * n, -1), e.reshape(b * n, -1), graph, "configs[3] share", adam=True, one_call_step=True)


def test_cfg4_share_vs_oracle():
    """The per-GPU share of BASELINE configs[4]: 2048 graphs of 8-128 links packed with CSR offsets, shared weights, one
    Huber mean over all rows x channels."""
    import bench
    b = 2048
    sizes, offs, row_ptr, col_idx, x, e, _ = bench.synth_ragged(np.random.default_rng(44), b, 8, 128)
    spec = GnnSpec(n_nodes=1, feat_dim=64, share_weights=True, variable_graphs=True)
    P = f32_params(spec, np.random.default_rng(45))
    pb = PackedBatch(b, 0, v2xgnn.pack_xe(x, e), row_ptr, col_idx, graph_off=offs)
    _parity_with_oracle(spec, P, pb, x, e, (offs, pb.row_ptr, pb.col_idx), "configs[4] share")


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
