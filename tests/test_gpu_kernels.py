"""Per-kernel parity of the HIP path (called through the C ABI) against the CPU oracle."""
import ctypes as C

import numpy as np
import pytest

import v2xgnn
from v2xgnn import GnnSpec, PackedBatch, GnnEngine
from v2xgnn import lib as vlib
from v2xgnn.engine import DeviceBatch, _batch_struct
from oracle import compact as oc
from oracle.keras_semantics import KerasAdam
from util import (ospec, random_inputs, f32_params, assert_close, assert_fwd_close, assert_grad_close)

pytestmark = pytest.mark.gpu


def _t(a, dev="cuda:0"):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _sync():
    import torch
    torch.cuda.synchronize()


def _variable_batch(rng, sizes, F, density=0.4):
    """Ragged batch: graphs of different node counts packed with graph_off (BASELINE config 5)."""
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    R = int(offs[-1])
    row_ptr = [0]
    cols = []
    max_e = 0
    for n in sizes:
        adj = rng.uniform(size=(n, n)) < density
        e_g = 0
        for q in range(n):
            src = np.nonzero(adj[:, q])[0]
            cols.append(src)
            row_ptr.append(row_ptr[-1] + len(src))
            e_g += len(src)
        max_e = max(max_e, e_g)
    col_idx = np.concatenate(cols).astype(np.int32) if cols else np.zeros(0, np.int32)
    xe = np.zeros((R, 16), np.float32)
    return PackedBatch(len(sizes), 0, xe, np.array(row_ptr, np.int32), col_idx, max_e,
                       graph_off=offs, max_nodes=int(max(sizes)))


@pytest.mark.parametrize("B,N,F,ref_topo", [(1, 4, 16, True), (7, 4, 16, False), (64, 20, 64, True),
                                            (5, 33, 64, False), (3, 100, 32, True), (130, 20, 64, False),
                                            (2, 100, 256, True), (3, 200, 64, True), (3, 200, 64, False),
                                            (2, 129, 128, True)])
def test_agg_fwd_bwd_fixed(B, N, F, ref_topo):
    """(N >= 32 dense graphs run k_agg_dense: the reference topology through the complement -- column sum minus the
    non-neighbour rows --, random adjacencies of density 0.5 as an MFMA product; 129 and 200 links: masks of 5 / 7 words)"""
    rng = np.random.default_rng(B * 1000 + N + F)
    _, _, adj = random_inputs(rng, B, N, ref_topology=ref_topo)
    row_ptr, col_idx, max_e = v2xgnn.adj_to_csr(adj)
    pb = PackedBatch(B, N, np.zeros((B * N, 16), np.float32), row_ptr, col_idx, max_e)
    h = rng.normal(size=(B * N, F)).astype(np.float32)
    M = oc.csr_to_matrix((np.arange(B + 1) * N).astype(np.int32), row_ptr, col_idx, np.float64)
    lib = vlib.load_library()
    db = DeviceBatch(pb, "cuda:0")
    s = _batch_struct(db)
    import torch
    hd, out = _t(h), torch.empty((B * N, F), dtype=torch.float32, device="cuda:0")
    vlib.check(lib, lib.v2x_agg_fwd(C.byref(s), N, F, hd.data_ptr(), out.data_ptr(), None))
    _sync()
    assert_close(out.cpu().numpy(), M @ h.astype(np.float64), 1e-5, 1e-5, "agg fwd")
    vlib.check(lib, lib.v2x_agg_bwd(C.byref(s), N, F, hd.data_ptr(), out.data_ptr(), None))
    _sync()
    assert_close(out.cpu().numpy(), M.T @ h.astype(np.float64), 1e-5, 1e-5, "agg bwd (transpose)")


def test_agg_variable_sizes():
    """Ragged graphs (8..128 nodes) incl. an isolated-node graph and an empty-edge graph."""
    rng = np.random.default_rng(3)
    sizes = [8, 128, 1, 17, 64, 9, 33, 100, 2, 5, 77]
    F = 64
    pb = _variable_batch(rng, sizes, F)
    R = pb.n_rows
    h = rng.normal(size=(R, F)).astype(np.float32)
    M = oc.csr_to_matrix(pb.graph_off, pb.row_ptr, pb.col_idx, np.float64)
    lib = vlib.load_library()
    db = DeviceBatch(pb, "cuda:0")
    s = _batch_struct(db)
    import torch
    hd, out = _t(h), torch.empty((R, F), dtype=torch.float32, device="cuda:0")
    vlib.check(lib, lib.v2x_agg_fwd(C.byref(s), 0, F, hd.data_ptr(), out.data_ptr(), None))
    _sync()
    assert_close(out.cpu().numpy(), M @ h.astype(np.float64), 1e-5, 1e-5, "agg fwd ragged")
    vlib.check(lib, lib.v2x_agg_bwd(C.byref(s), 0, F, hd.data_ptr(), out.data_ptr(), None))
    _sync()
    assert_close(out.cpu().numpy(), M.T @ h.astype(np.float64), 1e-5, 1e-5, "agg bwd ragged")


def test_agg_mixed_forms_in_one_ragged_batch():
    """k_agg_dense decides PER GRAPH between the complement walk (at most 8 non-edges per row: the reference topology, complete
    graphs, a graph of one node) and the MFMA product (random adjacency of density 0.5, an edgeless graph) -- all in one batch."""
    rng = np.random.default_rng(17)
    sizes = [64, 40, 1, 128, 33, 2, 96, 50]
    kinds = ['ref', 'rand', 'ref', 'ref', 'rand', 'ref', 'full', 'empty']
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    row_ptr, cols, max_e = [0], [], 0
    for n, kind in zip(sizes, kinds):
        if kind == 'ref':
            adj = ~np.eye(n, dtype=bool)
            for q in range(n):
                if n > 1:
                    adj[rng.choice([p for p in range(n) if p != q]), q] = False
        elif kind == 'full':
            adj = ~np.eye(n, dtype=bool)
        elif kind == 'empty':
            adj = np.zeros((n, n), bool)
        else:
            adj = rng.uniform(size=(n, n)) < 0.5
        e_g = 0
        for q in range(n):
            src = np.nonzero(adj[:, q])[0]
            cols.append(src)
            row_ptr.append(row_ptr[-1] + len(src))
            e_g += len(src)
        max_e = max(max_e, e_g)
    col_idx = np.concatenate(cols).astype(np.int32)
    R, F = int(offs[-1]), 64
    pb = PackedBatch(len(sizes), 0, np.zeros((R, 16), np.float32), np.array(row_ptr, np.int32), col_idx, max_e,
                     graph_off=offs, max_nodes=max(sizes))
    h = rng.normal(size=(R, F)).astype(np.float32)
    M = oc.csr_to_matrix(pb.graph_off, pb.row_ptr, pb.col_idx, np.float64)
    lib = vlib.load_library()
    import torch
    db = DeviceBatch(pb, "cuda:0")
    s = _batch_struct(db)
    hd, out = _t(h), torch.empty((R, F), dtype=torch.float32, device="cuda:0")
    vlib.check(lib, lib.v2x_agg_fwd(C.byref(s), 0, F, hd.data_ptr(), out.data_ptr(), None))
    _sync()
    assert_close(out.cpu().numpy(), M @ h.astype(np.float64), 1e-5, 2e-5, "agg fwd, mixed forms")
    vlib.check(lib, lib.v2x_agg_bwd(C.byref(s), 0, F, hd.data_ptr(), out.data_ptr(), None))
    _sync()
    assert_close(out.cpu().numpy(), M.T @ h.astype(np.float64), 1e-5, 2e-5, "agg bwd (transpose), mixed forms")


def test_agg_empty_graphs():
    """No edges at all: aggregation must be exactly zero."""
    B, N, F = 9, 4, 16
    pb = PackedBatch(B, N, np.zeros((B * N, 16), np.float32), np.zeros(B * N + 1, np.int32),
                     np.zeros(0, np.int32), 0)
    lib = vlib.load_library()
    db = DeviceBatch(pb, "cuda:0")
    s = _batch_struct(db)
    import torch
    hd = _t(np.random.default_rng(0).normal(size=(B * N, F)).astype(np.float32))
    out = torch.full((B * N, F), 7.0, dtype=torch.float32, device="cuda:0")
    vlib.check(lib, lib.v2x_agg_fwd(C.byref(s), N, F, hd.data_ptr(), out.data_ptr(), None))
    _sync()
    assert float(out.abs().max()) == 0.0


CASES = [  # N, F, L, shared, B
    (4, 16, 2, False, 1),
    (4, 16, 2, False, 37),
    (20, 64, 2, False, 70),
    (20, 64, 2, True, 70),
    (6, 32, 3, False, 130),
    (20, 64, 2, False, 4096 // 16),
    # wide-feature path (feat_dim >= 128, kernels_wide.hpp)
    (5, 128, 2, False, 140),
    (12, 256, 3, True, 150),
    (100, 256, 3, False, 6),      # BASELINE config 4 shape: 100 links, feat_dim 256, 3 layers
]


def _setup(N, F, L, shared, B, seed, with_nbr=False):
    spec = GnnSpec(n_nodes=N, feat_dim=F, n_mp_layers=L, share_weights=shared)
    rng = np.random.default_rng(seed)
    P = f32_params(spec, rng)
    x, e, adj = random_inputs(rng, B, N)
    nbr = rng.normal(0, 0.5, size=(B, N, F)).astype(np.float32) if with_nbr else None
    eng = GnnEngine(spec)
    eng.set_weights(oc.params_to_list(P))
    pb = PackedBatch.from_dense(x, e, adj, nbr)
    graph = ((np.arange(B + 1) * N).astype(np.int32), pb.row_ptr, pb.col_idx)
    M = oc.csr_to_matrix(*graph, dtype=np.float64)
    xr, er = x.reshape(B * N, -1).astype(np.float64), e.reshape(B * N, -1).astype(np.float64)
    nr = None if nbr is None else nbr.reshape(B * N, F).astype(np.float64)
    return spec, P, eng, pb, M, xr, er, nr


@pytest.mark.parametrize("N,F,L,shared,B", CASES)
@pytest.mark.parametrize("with_nbr", [False, True])
def test_node_update_fwd_bwd(N, F, L, shared, B, with_nbr):
    """GNNLayer.call of every stage (BS_brain.py:44-51) and its backward, one layer at a time."""
    import torch
    spec, P, eng, pb, M, xr, er, nr = _setup(N, F, L, shared, B, 11 + N + F + B, with_nbr)
    lib = eng._lib
    _, cache = oc.forward(ospec(spec), P, xr, er, M, nr)
    R = B * N
    xe = _t(pb.xe)
    rng = np.random.default_rng(5)
    for s in range(L + 1):
        hp = None if s == 0 else _t(cache['h'][s - 1].astype(np.float32))
        ap = (None if nr is None else _t(nr.astype(np.float32))) if s == 0 else _t(cache['a'][s - 1].astype(np.float32))
        # oracle on the SAME fp32 inputs the kernel sees
        gs = P['gnn'][s]
        hp64 = None if hp is None else hp.cpu().numpy().astype(np.float64)
        ap64 = None if ap is None else ap.cpu().numpy().astype(np.float64)
        u = xr if s == 0 else np.concatenate([hp64, xr], axis=1)
        pre = oc._slot_mm(u, gs['W1']) + oc._slot_mm(er, gs['W2']) + oc._slot_bias(gs['b'], R)
        if ap64 is not None:
            pre = pre + oc._slot_mm(ap64, gs['W3'])
        ref = np.maximum(pre, 0) if s < L else pre
        out = torch.empty((R, F), dtype=torch.float32, device="cuda:0")
        p = lambda t: None if t is None else t.data_ptr()
        vlib.check(lib, lib.v2x_node_update_fwd(eng._h, s, R, xe.data_ptr(), p(hp), p(ap), out.data_ptr(), None), eng._h)
        _sync()
        assert_fwd_close(out.cpu().numpy(), ref, "node_update fwd stage %d" % s)

        # backward of the same layer
        dpre = rng.normal(size=(R, F)).astype(np.float32)
        d64 = dpre.astype(np.float64)
        S = spec.n_slots
        ref_dW = np.concatenate([oc._slot_wgrad(u, d64, S), oc._slot_wgrad(er, d64, S),
                                 oc._slot_wgrad(ap64, d64, S) if ap64 is not None else np.zeros((S, F, F))], axis=1)
        ref_db = oc._slot_bgrad(d64, S)
        gout = torch.zeros(eng.n_params, dtype=torch.float32, device="cuda:0")
        dh = torch.empty((R, F), dtype=torch.float32, device="cuda:0")
        da = torch.empty((R, F), dtype=torch.float32, device="cuda:0")
        dd = _t(dpre)
        vlib.check(lib, lib.v2x_node_update_bwd(eng._h, s, R, xe.data_ptr(), p(hp), p(ap), dd.data_ptr(),
                                                dh.data_ptr() if s else None, da.data_ptr() if s else None,
                                                gout.data_ptr(), None), eng._h)
        _sync()
        g = gout.cpu().numpy()
        k_real = ref_dW.shape[1]
        per_slot = k_real * F + F
        off = sum((spec.stage_in_a(t) + spec.edge_in + F) * F + F for t in range(s)) * S
        got = g[off:off + S * per_slot].reshape(S, per_slot)
        assert_grad_close(got[:, :k_real * F].reshape(S, k_real, F), ref_dW, "dW stage %d" % s)
        assert_grad_close(got[:, k_real * F:], ref_db, "db stage %d" % s)
        if s:
            assert_grad_close(dh.cpu().numpy(), oc._slot_mm_t(d64, gs['W1'])[:, :F], "dh_prev stage %d" % s)
            assert_grad_close(da.cpu().numpy(), oc._slot_mm_t(d64, gs['W3']), "dagg_prev stage %d" % s)


@pytest.mark.parametrize("N,F,L,shared,B", CASES)
def test_mlp_fwd_and_huber_bwd(N, F, L, shared, B):
    """Dense 80-40-20-C (BS_brain.py:176-179) forward; Huber (:86-87) + backward."""
    import torch
    spec, P, eng, pb, M, xr, er, nr = _setup(N, F, L, shared, B, 23 + N + F + B)
    lib = eng._lib
    R = B * N
    rng = np.random.default_rng(9)
    h = rng.normal(0.3, 1.0, size=(R, F)).astype(np.float32)
    a = rng.normal(0.0, 3.0, size=(R, F)).astype(np.float32)
    z = [np.concatenate([xr, h.astype(np.float64), a.astype(np.float64)], axis=1)]
    for i in range(4):
        d = P['dense'][i]
        pre = oc._slot_mm(z[i], d['W']) + oc._slot_bias(d['b'], R)
        z.append(np.maximum(pre, 0) if i < 3 else pre)
    xe, hd, ad = _t(pb.xe), _t(h), _t(a)
    q = torch.empty((R, spec.n_channels), dtype=torch.float32, device="cuda:0")
    vlib.check(lib, lib.v2x_mlp_fwd(eng._h, R, xe.data_ptr(), hd.data_ptr(), ad.data_ptr(), q.data_ptr(), None), eng._h)
    _sync()
    assert_fwd_close(q.cpu().numpy(), z[4], "mlp fwd")

    # targets: some inside the quadratic zone of Huber, some in the linear zone
    y = (z[4] + rng.normal(0, 1.5, size=z[4].shape)).astype(np.float32)
    os_ = ospec(spec)
    qk = q.cpu().numpy().astype(np.float64)      # differentiate at the kernel's own q (clip boundaries)
    loss_ref, dq = oc.huber_loss_and_grad(os_, qk, y.astype(np.float64))
    grads = {}
    g = dq
    S = spec.n_slots
    for i in range(3, -1, -1):
        if i < 3:
            g = g * (z[i + 1] > 0)
        grads[i] = (oc._slot_wgrad(z[i], g, S), oc._slot_bgrad(g, S))
        g = oc._slot_mm_t(g, P['dense'][i]['W'])
    Dn = spec.node_in
    gout = torch.zeros(eng.n_params, dtype=torch.float32, device="cuda:0")
    dh = torch.empty((R, F), dtype=torch.float32, device="cuda:0")
    da = torch.empty((R, F), dtype=torch.float32, device="cuda:0")
    n_out = 1 if spec.variable_graphs else N
    loss = torch.empty(n_out, dtype=torch.float32, device="cuda:0")
    yd = _t(y)
    vlib.check(lib, lib.v2x_mlp_huber_bwd(eng._h, R, B, xe.data_ptr(), hd.data_ptr(), ad.data_ptr(), yd.data_ptr(),
                                          dh.data_ptr(), da.data_ptr(), gout.data_ptr(), loss.data_ptr(), None), eng._h)
    _sync()
    assert_close(loss.cpu().numpy(), loss_ref, 1e-4, 1e-6, "huber loss per output")
    assert_grad_close(dh.cpu().numpy(), g[:, Dn:Dn + F], "mlp dh")
    assert_grad_close(da.cpu().numpy(), g[:, Dn + F:], "mlp dagg")
    flat = gout.cpu().numpy()
    gl = v2xgnn.flat_to_keras_list(spec, flat)
    base = (L + 1) * S * 4
    for i in range(4):
        for k in range(S):
            assert_grad_close(gl[base + (i * S + k) * 2], grads[i][0][k], "dense %d kernel grad slot %d" % (i, k))
            assert_grad_close(gl[base + (i * S + k) * 2 + 1], grads[i][1][k], "dense %d bias grad slot %d" % (i, k))


def test_adam_step_matches_keras_formula():
    """keras.optimizers.Adam(lr=1e-3, beta_1=0.5, beta_2=0.999), eps inside sqrt()+eps (Appendix B.6)."""
    import torch
    rng = np.random.default_rng(1)
    n = 10007                                  # not a multiple of 4 / 256: tail handling
    p = rng.normal(size=n).astype(np.float32)
    lib = vlib.load_library()
    pd, md, vd = _t(p), torch.zeros(n, device="cuda:0"), torch.zeros(n, device="cuda:0")
    ref = [p.copy()]
    opt = KerasAdam()
    for t in range(1, 6):
        g = (rng.normal(size=n) * (rng.uniform(size=n) < 0.8)).astype(np.float32)   # includes exact zeros
        opt.step(ref, [g])
        gd = _t(g)
        vlib.check(lib, lib.v2x_adam_step(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), n, t,
                                          1e-3, 0.5, 0.999, 1e-7, None))
        _sync()
        assert_close(pd.cpu().numpy(), ref[0], 1e-6, 1e-6, "adam params after step %d" % t)
    assert_close(md.cpu().numpy(), opt.m[0], 1e-6, 1e-7, "adam m")
    assert_close(vd.cpu().numpy(), opt.v[0], 1e-6, 1e-9, "adam v")
