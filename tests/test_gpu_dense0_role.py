"""GPU: Dense-0's weight gradient as a role of the graph layers' weight-gradient launch (round 6: the shares of the metric's
global batch).  At <= 2 tiles per wave k_mlp_train_wg leaves dW0 to k_wgrad (kernels_mlpwg.hpp WG0 = false writes the gated
dz1 rows; kernels.hpp WG_KIND_DENSE0 / WG_KIND_DENSE0_FRAG reads h_L | x | a_L, the latter from the fragment-major hand-over
of the fused graph-layer kernels).  Both forms against each other and against the float64 oracle (TF autodiff of the
K.dot of /root/reference/BS_brain.py:176 under fit, :218-223).  Also the opt-in streamed MLP (kernels_mlpstream.hpp, V2X_MLP_STREAM=1:
one wave per (slot, tile), all four Dense weight gradients as roles of k_wgrad; measured slower, profiles/r06_mlp_stream_ab.txt)."""
import os

import numpy as np
import pytest

import v2xgnn
from v2xgnn import GnnSpec, PackedBatch, GnnEngine
from util import f32_params, random_inputs, oracle_step, assert_grads_match_oracle, assert_close, assert_fwd_close
from oracle import compact as oc

pytestmark = pytest.mark.gpu


class _env(object):
    def __init__(self, **kw):
        self.kw = {k: str(v) for k, v in kw.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update(self.kw)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _grads(spec, weights, pb, y, wg0, n_global=None, stream=0, **create_env):
    """(path_info, q, loss, flat gradient, names of the launches) of one forward_backward with V2X_MLP_WG0 = wg0 (and the streamed
    MLP of kernels_mlpstream.hpp forced on / off)"""
    with _env(V2X_FUSED_COMPL=0, **create_env):
        eng = GnnEngine(spec)
    eng.set_weights(weights)
    with _env(V2X_MLP_WG0=wg0, V2X_MLP_STREAM=stream):
        info = eng.path_info(pb)
        q = eng.forward(pb)
        eng.profile(True)
        loss = eng.forward_backward(pb, y, n_global=n_global)
        names = set(eng.profile_read())
        eng.profile(False)
    g = eng.get_grad_flat()
    eng.close()
    return info, q, loss, g, names


CASES = [  # N, F, L, B, shared weights, what the hand-over between the graph layers and the MLP is
    (20, 64, 2, 64, False, "fragment-major"),      # fused graph layers (split tiles at this size), per-node weights
    (20, 64, 2, 512, False, "fragment-major"),     # the 8-GPU share of the metric's batch
    (20, 64, 3, 48, False, "fragment-major"),      # three stages: the embed gradient does not divide over them -> stays in-kernel
    (30, 64, 2, 32, False, "row-major"),           # 30 links: layer-wise graph layers, row-major h_L / a_L
    (20, 64, 2, 16, True, "row-major"),            # shared weights: one slot, 320 node rows
    (12, 64, 1, 40, False, "row-major"),           # 40 graphs: not whole 16-graph groups -> row-major hand-over, rows not whole 16-row tiles
]


@pytest.mark.parametrize("N,F,L,B,share,handoff", CASES)
def test_dense0_role_equals_in_kernel_gradient_and_oracle(N, F, L, B, share, handoff):
    spec = GnnSpec(n_nodes=N, feat_dim=F, n_mp_layers=L, share_weights=share)
    rng = np.random.default_rng(900 + N + L + B)
    P = f32_params(spec, rng)
    weights = oc.params_to_list(P)
    x, e, adj = random_inputs(rng, B, N, ref_topology=True)
    pb = PackedBatch.from_dense(x, e, adj)
    q0 = None
    with _env(V2X_FUSED_COMPL=0):
        probe = GnnEngine(spec)
    probe.set_weights(weights)
    q0 = probe.forward(pb)
    probe.close()
    y = (q0 + rng.normal(0, 1.2, size=q0.shape)).astype(np.float32)
    info_in, q_in, loss_in, g_in, names_in = _grads(spec, weights, pb, y, 1)
    info_out, q_out, loss_out, g_out, names_out = _grads(spec, weights, pb, y, 0)
    assert info_in["dense0_dw"] == "k_mlp_train_wg" and "k_mlp_train_wg" in names_in and "k_wgrad_gnn_d0" not in names_in, (info_in, names_in)
    assert info_in["handoff"] == handoff, info_in
    n_idx = B * N if share else B
    can = (F // 16) % L == 0 and n_idx % 16 == 0
    if can:
        assert info_out["dense0_dw"] == "k_wgrad" and {"k_mlp_train_wg123", "k_wgrad_gnn_d0"} <= names_out, (info_out, names_out)
    else:
        assert info_out["dense0_dw"] == "k_mlp_train_wg" and "k_mlp_train_wg" in names_out, (info_out, names_out)
    # same forward, same loss, same data gradients: Dense 1..3 come out of the MLP launch in the same order either way (bitwise);
    # Dense-0's block and the graph layers' (their launch is cut into other row chunks when Dense-0 rides along) differ by the
    # summation order over rows only
    assert np.array_equal(q_in, q_out) and np.array_equal(loss_in, loss_out)
    li, lo = v2xgnn.flat_to_keras_list(spec, g_in), v2xgnn.flat_to_keras_list(spec, g_out)
    shapes = v2xgnn.keras_list_shapes(spec)
    n_slots = 1 if share else N
    first_dense0 = 4 * n_slots * (L + 1)
    for i, (a, b) in enumerate(zip(li, lo)):
        if i < first_dense0 + 2 * n_slots:
            scale = max(np.abs(a).max(), 1e-30)
            assert np.abs(a - b).max() <= 2e-5 * scale, (i, shapes[i], np.abs(a - b).max(), scale)
        else:
            assert np.array_equal(a, b), (i, shapes[i])
    # ... and both against the oracle
    graph = ((np.arange(B + 1) * N).astype(np.int32), pb.row_ptr, pb.col_idx)
    ref = oracle_step(spec, P, x.reshape(B * N, -1), e.reshape(B * N, -1), graph, y, q_at=q_out)
    assert_fwd_close(q_out, ref['q'], "forward")
    assert_close(loss_out, ref['loss'], 2e-4, 1e-6, "loss")
    assert_grads_match_oracle(lo, P, ref, "Dense-0 as a k_wgrad role")
    assert_grads_match_oracle(li, P, ref, "Dense-0 in k_mlp_train_wg")
    # ... and the streamed MLP (kernels_mlpstream.hpp: one wave per (slot, tile), ALL Dense weight gradients as roles of k_wgrad):
    # the same forward, loss and data gradients bit for bit, every weight gradient up to the order of its sum over rows
    info_s, q_s, loss_s, g_s, names_s = _grads(spec, weights, pb, y, 0, stream=1)
    if can and not share and L + 5 <= 8:
        assert info_s["mlp"] == "stream" and {"k_mlp_stream", "k_wgrad_gnn_d0123"} <= names_s, (info_s, names_s)
        assert "k_mlp_train_wg123" not in names_s and "k_mlp_train_wg" not in names_s, names_s
    else:
        assert info_s["mlp"] == "train_wg" and "k_mlp_stream" not in names_s, (info_s, names_s)
    assert np.array_equal(q_in, q_s) and np.array_equal(loss_in, loss_s)
    ls = v2xgnn.flat_to_keras_list(spec, g_s)
    for i, (a, b) in enumerate(zip(li, ls)):
        scale = max(np.abs(a).max(), 1e-30)
        assert np.abs(a - b).max() <= 2e-5 * scale, (i, shapes[i], np.abs(a - b).max(), scale)
    assert_grads_match_oracle(ls, P, ref, "the streamed MLP")


def test_dense0_role_is_chosen_by_batch_size_and_replays_as_a_graph():
    """The library's own choice: the 512-, 1024- and 2048-graph shares of the metric's batch hand Dense-0 to k_wgrad, 4096
    graphs keep it in the MLP launch; a captured fit step (hipGraph replay) gives the eager step's weights bit for bit."""
    import torch
    import bench
    N, F = 20, 64
    spec = GnnSpec(n_nodes=N, feat_dim=F)
    rng = np.random.default_rng(5)
    weights = oc.params_to_list(f32_params(spec, rng))
    for B, want in ((512, "k_wgrad"), (1024, "k_wgrad"), (2048, "k_wgrad"), (4096, "k_mlp_train_wg")):
        x, e, adj, y = bench.synth_batch(rng, B, N)
        pb = PackedBatch.from_dense(x, e, adj)
        with _env(V2X_FUSED_COMPL=0):
            eager, graph = GnnEngine(spec), GnnEngine(spec, use_graph=True)
        assert eager.path_info(pb)["dense0_dw"] == want, (B, eager.path_info(pb))
        if B > 2048:
            eager.close(); graph.close()
            continue
        eager.set_weights(weights); graph.set_weights(weights)
        db = graph.to_device(pb)
        yd = torch.from_numpy(y).cuda()
        with torch.cuda.stream(torch.cuda.Stream()):
            for _ in range(3):
                lg = graph.train_step(db, yd)
            torch.cuda.synchronize()
        for _ in range(3):
            le = eager.train_step(pb, y)
        assert np.array_equal(lg.cpu().numpy(), le), B
        assert np.array_equal(graph.get_flat(), eager.get_flat()), B
        eager.close(); graph.close()
