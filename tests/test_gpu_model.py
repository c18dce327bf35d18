"""End-to-end parity of the drop-in boundary (BS / GnnQModel / GnnEngine) on the GPU:
golden vectors produced by the reference's own model code, oracle forward / backward / Adam,
the Keras-like call surface and its error behaviour."""
import contextlib
import os

import numpy as np
import pytest

import v2xgnn
from v2xgnn import GnnSpec, PackedBatch, GnnEngine, BS
from oracle import compact as oc
from util import (ospec, random_inputs, f32_params, assert_close, assert_fwd_close, assert_grad_close,
                  golden_forward_cases, golden_keras_list, golden_feed, GOLDEN)

pytestmark = pytest.mark.gpu

CASES = [  # N, F, L, shared, B
    (4, 16, 2, False, 1),
    (4, 16, 2, False, 512),       # the reference's own training configuration (RL_Train_main.py:29-31)
    (20, 64, 2, False, 96),
    (20, 64, 2, True, 96),
    (6, 32, 3, False, 50),
    # wide-feature path (feat_dim >= 128, kernels_wide.hpp)
    (5, 128, 2, False, 140),
    (12, 256, 3, True, 150),
    (100, 256, 3, False, 6),      # BASELINE config 4 shape: 100 links, feat_dim 256, 3 layers
]


def test_golden_forward_through_bs_predict():
    """Outputs of the reference's own `_create_model` code (tests/golden/make_golden.py) for the
    same injected weights and the same dict payload, through `BS.predict` (online and target)."""
    f, cases = golden_forward_cases()
    brain = BS(4, 3, 1, 16, 1, 4, seed=0)
    for case in cases:
        feed = golden_feed(f, case)
        for tag, target in (("online", False), ("target", True)):
            model = brain.target_model if target else brain.model
            # cast the float64 golden weights to the fp32 the engine stores
            model.set_weights(golden_keras_list(f, case, tag))
            out = brain.predict(feed, target=target)
            assert isinstance(out, list) and len(out) == 4
            for k in range(4):
                ref = f['%s/%s/out/%d' % (case, tag, k)]
                assert out[k].dtype == np.float32 and out[k].flags.writeable
                # golden is float64 arithmetic on float64 weights: allow fp32 rounding of both
                assert_close(out[k], ref, 5e-4, 5e-5, "golden %s/%s out %d" % (case, tag, k))


@pytest.mark.parametrize("N,F,L,shared,B", CASES)
def test_forward_and_gradients_vs_oracle(N, F, L, shared, B):
    spec = GnnSpec(n_nodes=N, feat_dim=F, n_mp_layers=L, share_weights=shared)
    rng = np.random.default_rng(100 + N + F + B)
    P = f32_params(spec, rng)
    x, e, adj = random_inputs(rng, B, N)
    eng = GnnEngine(spec)
    eng.set_weights(oc.params_to_list(P))
    pb = PackedBatch.from_dense(x, e, adj)
    graph = ((np.arange(B + 1) * N).astype(np.int32), pb.row_ptr, pb.col_idx)
    M = oc.csr_to_matrix(*graph, dtype=np.float64)
    os_ = ospec(spec)
    q_ref, cache = oc.forward(os_, P, x.reshape(B * N, -1).astype(np.float64), e.reshape(B * N, -1).astype(np.float64), M)
    q = eng.forward(pb)
    assert_fwd_close(q, q_ref, "forward q")
    # device-resident batch gives the same bits as the host batch
    db = eng.to_device(pb)
    q2 = eng.forward(db).cpu().numpy()
    assert np.array_equal(q, q2)

    y = (q_ref + rng.normal(0, 1.2, size=q_ref.shape)).astype(np.float32)
    # Huber's gradient is clip(q - y): with |q| in the hundreds..thousands (random weights, the constant
    # 10 dBm power feature, 18-neighbour sums) the fp32 rounding of q itself moves q - y by ~1e-3 relative.
    # Forward parity is asserted above; the backward is checked for the SAME q the kernels differentiate.
    loss_ref, dq = oc.huber_loss_and_grad(os_, q.astype(np.float64), y.astype(np.float64))
    g_ref = oc.backward(os_, P, cache, dq)
    loss = eng.forward_backward(pb, y)
    assert_close(loss, loss_ref, 2e-4, 1e-6, "per-output huber loss")
    got = v2xgnn.flat_to_keras_list(spec, eng.get_grad_flat())
    ref = oc.params_to_list(g_ref)
    assert len(got) == len(ref)
    for i, (a, b) in enumerate(zip(got, ref)):
        assert_grad_close(a, b, "gradient array %d" % i)


def test_gradient_shards_sum_to_global_gradient():
    """Data-parallel contract: per-shard gradients taken with the GLOBAL batch size
    in the Huber mean SUM to the full-batch gradient."""
    N, F, B = 20, 64, 64
    spec = GnnSpec(n_nodes=N, feat_dim=F)
    rng = np.random.default_rng(7)
    P = f32_params(spec, rng)
    x, e, adj = random_inputs(rng, B, N)
    eng = GnnEngine(spec)
    eng.set_weights(oc.params_to_list(P))
    pb = PackedBatch.from_dense(x, e, adj)
    y = rng.normal(2.5, 1.0, size=(B * N, 4)).astype(np.float32)
    full_loss = eng.forward_backward(pb, y)
    g_full = eng.get_grad_flat().astype(np.float64)
    acc = np.zeros_like(g_full)
    loss_acc = np.zeros(N)
    for r in range(4):
        sh = pb.shard(r, 4)
        ys = y.reshape(B, N, 4)[r * 16:(r + 1) * 16].reshape(-1, 4)
        loss_acc += eng.forward_backward(sh, ys, n_global=B)
        acc += eng.get_grad_flat()
    assert_grad_close(acc, g_full, "sum of shard gradients")
    assert_close(loss_acc, full_loss, 1e-5, 1e-7, "sum of shard losses")


@pytest.mark.parametrize("N,F,L,shared,B", [(4, 16, 2, False, 512), (20, 64, 2, False, 64), (20, 64, 2, True, 64)])
def test_train_steps_vs_oracle(N, F, L, shared, B):
    """3 x Model.fit(one batch) == 3 x (forward, Huber, backward, Keras Adam) of the oracle."""
    spec = GnnSpec(n_nodes=N, feat_dim=F, n_mp_layers=L, share_weights=shared)
    rng = np.random.default_rng(5 + N + B)
    P = f32_params(spec, rng)
    eng = GnnEngine(spec)
    eng.set_weights(oc.params_to_list(P))
    om = oc.OracleModel(ospec(spec), P, dtype=np.float64)
    for step in range(3):
        x, e, adj = random_inputs(rng, B, N)
        pb = PackedBatch.from_dense(x, e, adj)
        graph = ((np.arange(B + 1) * N).astype(np.int32), pb.row_ptr, pb.col_idx)
        y = rng.normal(2.5, 1.0, size=(B * N, 4)).astype(np.float32)
        # oracle gradient BEFORE its update, to know which entries Adam's sign-like first steps
        # make ill-conditioned (|g| ~ eps)
        _, g_ref, _ = om.loss_and_grads(x.reshape(B * N, -1), e.reshape(B * N, -1), graph, y)
        loss_ref = om.train_step(x.reshape(B * N, -1), e.reshape(B * N, -1), graph, y)
        loss = eng.train_step(pb, y)
        assert_close(loss, loss_ref, 5e-4, 1e-6, "loss at step %d" % step)
        got = eng.get_weights()
        ref = oc.params_to_list(om.params)
        gl = oc.params_to_list(g_ref)
        for i, (a, b, g) in enumerate(zip(got, ref, gl)):
            # Adam divides by sqrt(v)+1e-7: where |g| is at rounding-noise level the update direction is
            # not determined by fp32 arithmetic; compare those entries to within one full step (lr)
            scale = np.abs(g).max() or 1.0
            tight = np.abs(g) > 1e-4 * scale
            err = np.abs(a.astype(np.float64) - b)
            assert (err[tight] <= 2e-5 + 2e-4 * np.abs(b[tight])).all(), ("weights", i, step, err[tight].max())
            assert (err[~tight] <= 1.1e-3 * (step + 1)).all(), ("weights(ill-cond)", i, step, err[~tight].max())
    m, v, it = eng.get_optimizer_state()
    assert it == 3


def test_graph_replay_matches_eager_bitwise():
    import torch
    N, F, B = 20, 64, 128
    spec = GnnSpec(n_nodes=N, feat_dim=F)
    rng = np.random.default_rng(2)
    P = f32_params(spec, rng)
    x, e, adj = random_inputs(rng, B, N)
    y = rng.normal(2.5, 1.0, size=(B * N, 4)).astype(np.float32)
    pb = PackedBatch.from_dense(x, e, adj)
    res = []
    for use_graph in (False, True):
        eng = GnnEngine(spec, use_graph=use_graph)
        eng.set_weights(oc.params_to_list(P))
        db = eng.to_device(pb)
        yd = torch.from_numpy(y).cuda()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(4):
                loss = eng.train_step(db, yd)
            q = eng.forward(db)
        st.synchronize()
        res.append((eng.get_flat(), loss.cpu().numpy(), q.cpu().numpy()))
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)


def test_bs_call_surface_and_errors(tmp_path):
    """Constructor attributes (BS_brain.py:95-104), fit History keys (:835-837), target sync (:237-239),
    save/load_weights round trip (:863,:1254), ValueError on malformed payloads like Keras."""
    brain = BS(4, 3, 1, 16, 1, 4, seed=3)
    assert (brain.num_D2D, brain.num_CH, brain.num_Feedback, brain.num_Neighbor) == (4, 4, 16, 1)
    assert (brain.num_One_Node_Input, brain.num_One_Edge_Input, brain.num_One_D2D_Input, brain.num_D2D_Input) == (9, 4, 13, 68)
    a = np.load(os.path.join(GOLDEN, "golden_agent_n4.npz"))
    x = {k[len('fit_x/'):]: a[k] for k in a.files if k.startswith('fit_x/')}
    y = {k[len('fit_y/'):]: a[k] for k in a.files if k.startswith('fit_y/')}
    w0 = brain.model.get_weights()
    assert len(w0) == 80
    p_online = brain.predict(x)
    p_target = brain.predict(x, target=True)
    assert not np.allclose(p_online[0], p_target[0])          # independently initialised (BS_brain.py:105-106)
    hist = brain.train_dnn(x, y, 32)
    for k in range(1, 5):
        v = hist.history['D%d_Decide_Output_loss' % k]
        assert len(v) == 1 and np.isfinite(v[0]) and v[0] > 0
    assert abs(hist.history['loss'][0] - sum(hist.history['D%d_Decide_Output_loss' % k][0] for k in range(1, 5))) < 1e-6
    w1 = brain.model.get_weights()
    assert any(not np.array_equal(u, v) for u, v in zip(w0, w1))
    brain.update_target_model()
    for u, v in zip(brain.model.get_weights(), brain.target_model.get_weights()):
        assert np.array_equal(u, v)
    path = str(tmp_path / "Q-Network_model_weights-Episode-5-Step-20-Batch-512.h5")
    brain.model.save_weights(path)
    other = BS(4, 3, 1, 16, 1, 4, seed=99)
    other.model.load_weights(path)
    for u, v in zip(brain.model.get_weights(), other.model.get_weights()):
        assert np.array_equal(u, v)
    q1 = brain.predict_one_step({k[len('init/'):]: a[k] for k in a.files if k.startswith('init/')})
    assert len(q1) == 4 and q1[0].shape == (1, 4)
    # malformed payloads
    bad = dict(x)
    del bad['D2_Edge_Input']
    with pytest.raises(ValueError):
        brain.predict(bad)
    bad = dict(x)
    bad['D1_Node_Input'] = bad['D1_Node_Input'][:, :5]
    with pytest.raises(ValueError):
        brain.predict(bad)
    bad = dict(x)
    A = bad['Adjacency_Matrix'].copy()
    A[0, 0, 5] = 1.0            # not kron(Adj, I_F)
    bad['Adjacency_Matrix'] = A
    with pytest.raises(ValueError):
        brain.predict(bad)
    with pytest.raises(ValueError):
        brain.model.set_weights(w0[:-1])


def test_replay_targets_from_captured_agent_payload():
    """The x/y payload the reference's Agent.replay hands to fit (captured with the real simulator)
    trains the engine the same way it trains the oracle (one Adam step, same losses)."""
    a = np.load(os.path.join(GOLDEN, "golden_agent_n4.npz"))
    x = {k[len('fit_x/'):]: a[k] for k in a.files if k.startswith('fit_x/')}
    y = {k[len('fit_y/'):]: a[k] for k in a.files if k.startswith('fit_y/')}
    spec = GnnSpec()
    brain = BS(4, 3, 1, 16, 1, 4, seed=1)
    P = oc.params_from_list(ospec(spec), [w.astype(np.float64) for w in brain.model.get_weights()], np.float64)
    om = oc.OracleModel(ospec(spec), P, dtype=np.float64)
    xs, es, nbr, adj = v2xgnn.feed_to_arrays(spec, x)
    B = xs.shape[0]
    pb = PackedBatch.from_dense(xs, es, adj)
    graph = ((np.arange(B + 1) * 4).astype(np.int32), pb.row_ptr, pb.col_idx)
    yt = np.stack([y['D%d_Decide_Output' % k] for k in range(1, 5)], axis=1).reshape(-1, 4)
    loss_ref = om.train_step(xs.reshape(B * 4, -1).astype(np.float32), es.reshape(B * 4, -1).astype(np.float32), graph,
                             yt.astype(np.float32))
    hist = brain.train_dnn(x, y, B)
    got = [hist.history['D%d_Decide_Output_loss' % k][0] for k in range(1, 5)]
    assert_close(got, loss_ref, 5e-4, 1e-6, "losses on the captured replay payload")


def test_grad_tensor_aliases_library_memory():
    """The torch view handed to the RCCL all-reduce must alias (not copy) the engine's gradient buffer."""
    import torch
    eng = GnnEngine(GnnSpec())
    t = eng.grad_tensor()
    assert t.data_ptr() == int(eng._lib.v2x_grad_ptr(eng._h)) and t.numel() == eng.n_params and t.is_cuda
    t.fill_(3.0)
    torch.cuda.synchronize()
    assert (eng.get_grad_flat() == 3.0).all()


@pytest.mark.parametrize("dense", [False, True])
def test_ragged_graphs_shared_weights_vs_oracle(dense):
    """BASELINE config 5 shape: graphs of different sizes packed with graph_off, shared weights.  Forward, loss and
    every gradient against the CSR oracle (single Huber mean over all rows x channels).  dense: the reference
    topology (in-degree n-2) -> MFMA aggregation; otherwise sparse random graphs -> gather form."""
    from oracle.spec import GnnSpec as OSpec
    F = 64
    spec = GnnSpec(n_nodes=1, feat_dim=F, share_weights=True, variable_graphs=True)
    osp = OSpec(n_nodes=1, feat_dim=F, share_weights=True)
    rng = np.random.default_rng(42)
    sizes = [8, 33, 1, 128, 17, 64, 9, 100, 2, 40]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    R = int(offs[-1])
    row_ptr, cols, max_e = [0], [], 0
    for n in sizes:
        if dense:
            adj = ~np.eye(n, dtype=bool)
            for qq in range(n):
                if n > 1:
                    adj[rng.choice([p for p in range(n) if p != qq]), qq] = False
        else:
            adj = rng.uniform(size=(n, n)) < min(0.5, 6.0 / max(n, 1))
        e_g = 0
        for q in range(n):
            src = np.nonzero(adj[:, q])[0]
            cols.append(src)
            row_ptr.append(row_ptr[-1] + len(src))
            e_g += len(src)
        max_e = max(max_e, e_g)
    col_idx = np.concatenate(cols).astype(np.int32)
    x = np.concatenate([rng.normal(0.84, 0.39, size=(R, 4)), rng.normal(0.6, 0.21, size=(R, 4)), np.full((R, 1), 10.0)], 1).astype(np.float32)
    e = rng.normal(0.88, 0.11, size=(R, 4)).astype(np.float32)
    pb = PackedBatch(len(sizes), 0, v2xgnn.pack_xe(x, e), np.array(row_ptr, np.int32), col_idx, max_e,
                     graph_off=offs, max_nodes=max(sizes))
    P = f32_params(spec, rng)
    eng = GnnEngine(spec)
    eng.set_weights(oc.params_to_list(P))
    M = oc.csr_to_matrix(offs, pb.row_ptr, pb.col_idx, np.float64)
    q_ref, cache = oc.forward(osp, P, x.astype(np.float64), e.astype(np.float64), M)
    q = eng.forward(pb)
    assert_fwd_close(q, q_ref, "ragged forward")
    y = (q_ref + rng.normal(0, 1.2, size=q_ref.shape)).astype(np.float32)
    err = q.astype(np.float64) - y
    ab = np.abs(err)
    quad = np.minimum(ab, 1.0)
    loss_ref = (0.5 * quad * quad + (ab - quad)).sum() / (R * 4)
    dq = np.clip(err, -1, 1) / (R * 4)
    g_ref = oc.backward(osp, P, cache, dq)
    loss = eng.forward_backward(pb, y)
    assert loss.shape == (1,)
    assert_close(loss, [loss_ref], 2e-4, 1e-7, "ragged loss")
    got = v2xgnn.flat_to_keras_list(spec, eng.get_grad_flat())
    for i, (a, b) in enumerate(zip(got, oc.params_to_list(g_ref))):
        assert_grad_close(a, b, "ragged gradient array %d" % i)


def test_rccl_path_single_rank(tmp_path):
    """bench.py launched the way the driver launches it (torch.distributed.run, backend nccl == RCCL) with one
    rank: process-group init, the in-place all-reduce on the engine's aliased gradient buffer and the split
    forward_backward / apply_gradients step all run for real; multi-rank numerics are covered by the gloo test."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(GOLDEN.rstrip('/')).rsplit('/tests', 1)[0]
    env = dict(os.environ, V2X_FORCE_DP="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--min-seconds", "0",
           "--batch", "256", "--no-cpu-baseline", "--no-roofline", "--no-dropin", "--no-other-workloads"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 1 and res["value"] > 0 and res["config"]["global_batch"] == 256
    # the launch form of the timed region (--launch auto): both were probed, the faster one is named
    assert res["config"]["launch"] in ("eager", "hipGraph replay"), res["config"]["launch"]
    probe = res["config"]["launch_probe_ms"]
    assert set(probe) == {"hipGraph replay", "eager"} and all(v > 0 for v in probe.values()), probe
    assert (res["config"]["launch"] == "eager") == (probe["eager"] < probe["hipGraph replay"]), (res["config"]["launch"], probe)
    for forced, name in (("graph", "hipGraph replay"), ("eager", "eager")):
        out = subprocess.run(cmd + ["--launch", forced], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        r2 = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert r2["config"]["launch"] == name and r2["config"]["launch_probe_ms"] is None, r2["config"]


def test_ragged_large_batch_loss_is_split_reduced():
    """> 16k node rows with one Huber mean: the loss reduction is cut into ranges combined by the last workgroup.
    Checked against numpy on the engine's own q, twice (the arrival counter must re-arm itself)."""
    import bench
    rng = np.random.default_rng(9)
    sizes, offs, row_ptr, col_idx, x, e, y = bench.synth_ragged(rng, 500, 8, 128)
    assert offs[-1] > 16384
    spec = GnnSpec(n_nodes=1, feat_dim=64, share_weights=True, variable_graphs=True)
    pb = PackedBatch(len(sizes), 0, v2xgnn.pack_xe(x, e), row_ptr, col_idx, int((sizes * (sizes - 2)).max()),
                     graph_off=offs, max_nodes=int(sizes.max()))
    eng = GnnEngine(spec)
    eng.set_weights(oc.params_to_list(f32_params(spec, rng)))
    q = eng.forward(pb).astype(np.float64)
    ab = np.abs(q - y)
    quad = np.minimum(ab, 1.0)
    ref = (0.5 * quad * quad + (ab - quad)).mean()
    for _ in range(2):
        loss = eng.forward_backward(pb, y)
        assert_close(loss, [ref], 1e-5, 1e-7, "split loss reduction")


def test_graph_cache_survives_workspace_growth():
    """use_graph: a small batch is captured, a larger batch then grows (re-allocates) the activation workspace, and
    the small batch comes back -- the replayed graph must not reference the freed buffers."""
    spec = GnnSpec(n_nodes=20, feat_dim=64)
    rng = np.random.default_rng(77)
    P = f32_params(spec, rng)
    ref = GnnEngine(spec)                                   # eager reference
    ref.set_weights(oc.params_to_list(P))
    eng = GnnEngine(spec, use_graph=True)
    eng.set_weights(oc.params_to_list(P))
    import torch
    stream = torch.cuda.Stream()
    batches = {}
    for B in (2, 300, 2, 1200, 300, 2):
        if B not in batches:
            x, e, adj = random_inputs(rng, B, 20)
            y = rng.normal(2.5, 1.0, size=(B * 20, 4)).astype(np.float32)
            pb = PackedBatch.from_dense(x, e, adj)
            batches[B] = (pb, y, eng.to_device(pb), torch.from_numpy(y).cuda())      # device copies: STABLE pointers
            torch.cuda.synchronize()                        # (the copies ran on the default stream)
        pb, y, db, yd = batches[B]
        with torch.cuda.stream(stream):                     # graphs are only captured on a non-default stream
            q = eng.forward(db)
            la = eng.forward_backward(db, yd)
        stream.synchronize()
        assert np.array_equal(q.cpu().numpy(), ref.forward(pb)), B
        lb = ref.forward_backward(pb, y)
        assert np.array_equal(la.cpu().numpy(), lb) and np.array_equal(eng.get_grad_flat(), ref.get_grad_flat()), B


def test_graph_replay_of_train_steps_at_alternating_batch_sizes():
    """train_step keeps Adam outside the captured graph; the slab counts the Adam launch sums must be those of the
    REPLAYED weight-gradient launch, not of whichever batch size ran last."""
    import torch
    spec = GnnSpec(n_nodes=20, feat_dim=64)
    rng = np.random.default_rng(5)
    P = f32_params(spec, rng)
    ref = GnnEngine(spec)
    ref.set_weights(oc.params_to_list(P))
    eng = GnnEngine(spec, use_graph=True)
    eng.set_weights(oc.params_to_list(P))
    stream = torch.cuda.Stream()
    batches = {}
    for B in (2400, 40, 2400, 40, 2400):                     # 4 resp. 1 weight-gradient slabs per slot
        if B not in batches:
            x, e, adj = random_inputs(rng, B, 20)
            y = rng.normal(2.5, 1.0, size=(B * 20, 4)).astype(np.float32)
            pb = PackedBatch.from_dense(x, e, adj)
            batches[B] = (pb, y, eng.to_device(pb), torch.from_numpy(y).cuda())
            torch.cuda.synchronize()
        pb, y, db, yd = batches[B]
        with torch.cuda.stream(stream):
            la = eng.train_step(db, yd)
        stream.synchronize()
        lb = ref.train_step(pb, y)
        assert np.array_equal(la.cpu().numpy(), lb), B
        assert np.array_equal(eng.get_flat(), ref.get_flat()), B


@pytest.mark.parametrize("use_graph", [False, True])
def test_wide_fit_step_with_adam_in_the_weight_gradient_launch(use_graph):
    """BASELINE configs[3]'s shape (100 links x 256 features x 3 layers, per-node weights): a fit step applies Keras Adam
    in the epilogue of the merged weight-gradient launch for the layers that launch writes in place (csrc/kernels_wide.hpp,
    WideWgradArgs::adam) and k_reduce_adam covers the rest.  Three steps must equal (a) the split step forward_backward +
    apply_gradients, where k_reduce_adam updates everything from the gradient buffer, and (b) the oracle's Keras Adam on the
    engine's own gradients; eager and as a replayed hipGraph (lr_t reaches the replayed kernels through device memory)."""
    import torch
    from oracle.keras_semantics import KerasAdam
    N, F, L, B = 100, 256, 3, 8
    spec = GnnSpec(n_nodes=N, feat_dim=F, n_mp_layers=L)
    rng = np.random.default_rng(321)
    P = f32_params(spec, rng)
    x, e, adj = random_inputs(rng, B, N, ref_topology=True)
    pb = PackedBatch.from_dense(x, e, adj)
    y = rng.normal(0.0, 1.0, size=(B * N, 4)).astype(np.float32)
    ctx = torch.cuda.stream(torch.cuda.Stream()) if use_graph else contextlib.nullcontext()
    with ctx:
        fused, split = GnnEngine(spec, use_graph=use_graph), GnnEngine(spec, use_graph=use_graph)
        for eng in (fused, split):
            eng.set_weights(oc.params_to_list(P))
        db = fused.to_device(pb)
        yd = torch.from_numpy(y).cuda()
        names = set()
        ref = oc.cast_params(P, np.float64)
        opt = KerasAdam()
        for step in range(3):
            if step == 0 and not use_graph:
                fused.profile(True)
            lf = fused.train_step(db, yd)
            if step == 0 and not use_graph:
                names = set(fused.profile_read())
                fused.profile(False)
            ls = split.forward_backward(db, yd)
            g = v2xgnn.flat_to_keras_list(spec, split.get_grad_flat())
            split.apply_gradients()
            torch.cuda.synchronize()
            assert np.allclose(lf.cpu().numpy(), ls.cpu().numpy(), rtol=1e-5, atol=1e-7)
            wf, ws = fused.get_flat(), split.get_flat()
            # same gradients, same Adam expressions; the two kernels may contract their fused multiply-adds differently, and
            # from the second step on the weights they differentiate at differ by that rounding: where a moment is at
            # rounding-noise level the direction of Adam's (sign-like) step is not determined -- a handful of entries
            err = np.abs(wf - ws)
            tol = 2e-6 if step == 0 else 1e-5
            assert (err > tol).mean() <= (0.0 if step == 0 else 1e-5), (step, err.max(), (err > tol).sum())
            opt.step(oc.param_arrays(ref), oc.param_arrays(oc.params_from_list(ospec(spec), g, np.float64)))
            for i, (a, b) in enumerate(zip(fused.get_weights(), oc.params_to_list(ref))):
                e2 = np.abs(a.astype(np.float64) - b)
                assert (e2 > 1e-5 * (step + 1)).mean() <= (0.0 if step == 0 else 1e-4), (step, i, e2.max())
        mf, vf, itf = fused.get_optimizer_state()
        ms, vs, its = split.get_optimizer_state()
        assert itf == its == 3
        # (second and third gradient taken at weights that differ by the rounding above: compare on the arrays' own scale)
        assert np.abs(mf - ms).max() <= 1e-4 * np.abs(ms).max() and np.abs(vf - vs).max() <= 1e-4 * np.abs(vs).max()
        if not use_graph:
            assert "k_wgrad_wide_all" in names and "k_wgrad_gnn" not in names, names
        fused.close()
        split.close()
