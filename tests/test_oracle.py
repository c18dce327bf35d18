"""CPU: the oracle against the golden vectors, against itself (three formulations), and its
hand-written backward against torch autograd and finite differences."""
import numpy as np
import pytest
import torch

from oracle.spec import GnnSpec as OSpec
from oracle import compact as oc, literal as ol
from oracle.keras_semantics import KerasAdam, huber_mean, huber_grad, glorot_uniform
from util import golden_forward_cases, golden_keras_list, golden_feed


def test_param_counts_match_survey():
    assert OSpec().params_per_slot == 9456 and OSpec().n_params == 37824
    s2 = OSpec(n_nodes=20, feat_dim=64)
    assert s2.params_per_slot == 38352 and s2.n_params == 767040
    assert len(oc.params_to_list(oc.init_params(OSpec(), np.random.default_rng(0)))) == 80


def test_oracle_reproduces_reference_model_code_outputs():
    """Golden vectors = the reference's own _create_model / GNNLayer.call / AggLayer.call executed in
    the build container (tests/golden/make_golden.py).  Both formulations must reproduce them."""
    f, cases = golden_forward_cases()
    spec = OSpec()
    assert set(cases) == {'env_b1', 'replay_b32', 'synthetic_b6'}
    for case in cases:
        feed = golden_feed(f, case)
        for tag in ('online', 'target'):
            P = oc.params_from_list(spec, golden_keras_list(f, case, tag), np.float64)
            outs = ol.forward_literal(spec, P, feed)
            A = feed['Adjacency_Matrix']
            adj = A[:, ::16, ::16]
            assert np.array_equal(np.kron(adj, np.eye(16)), A)            # SURVEY.md A.2 identity
            x = np.stack([feed['D%d_Node_Input' % k] for k in range(1, 5)], 1)
            e = np.stack([feed['D%d_Edge_Input' % k] for k in range(1, 5)], 1)
            nb = np.stack([feed['D%d_Neighbor_Input' % k] for k in range(1, 5)], 1)
            M = oc.csr_to_matrix(*oc.adj_to_csr(adj), dtype=np.float64)
            q, _ = oc.forward(spec, P, x.reshape(-1, 9), e.reshape(-1, 4), M, nb.reshape(-1, 16))
            q = q.reshape(-1, 4, 4)
            for k in range(4):
                ref = f['%s/%s/out/%d' % (case, tag, k)]
                assert np.abs(outs[k] - ref).max() < 1e-12
                assert np.abs(q[:, k] - ref).max() < 1e-12


def _torch_forward(spec, tp, x, e, adj, nbr=None):
    L = spec.n_mp_layers
    it = iter(tp)
    gn = [dict(W1=next(it), W2=next(it), W3=next(it), b=next(it)) for _ in range(L + 1)]
    dn = [dict(W=next(it), b=next(it)) for _ in range(4)]
    X, E, A = torch.tensor(x), torch.tensor(e), torch.tensor(adj)
    if spec.share_weights:
        mm = lambda a, W: a @ W[0]
        bb = lambda b: b[0]
    else:
        mm = lambda a, W: torch.einsum('bnk,nkf->bnf', a, W)
        bb = lambda b: b[None]
    pre = mm(X, gn[0]['W1']) + mm(E, gn[0]['W2']) + bb(gn[0]['b'])
    if nbr is not None:
        pre = pre + mm(torch.tensor(nbr), gn[0]['W3'])
    h = torch.relu(pre)
    a = torch.einsum('bpq,bpf->bqf', A, h)
    for s in range(1, L + 1):
        pre = mm(torch.cat([h, X], -1), gn[s]['W1']) + mm(E, gn[s]['W2']) + mm(a, gn[s]['W3']) + bb(gn[s]['b'])
        h = torch.relu(pre) if s < L else pre
        a = torch.einsum('bpq,bpf->bqf', A, h)
    z = torch.cat([X, h, a], -1)
    for i in range(4):
        z = mm(z, dn[i]['W']) + bb(dn[i]['b'])
        if i < 3:
            z = torch.relu(z)
    return z


@pytest.mark.parametrize("N,F,L,shared,with_nbr", [(4, 16, 2, False, False), (4, 16, 2, False, True),
                                                   (7, 32, 3, False, False), (5, 16, 2, True, True),
                                                   (20, 64, 2, False, False)])
def test_backward_matches_torch_autograd(N, F, L, shared, with_nbr):
    spec = OSpec(n_nodes=N, feat_dim=F, n_mp_layers=L, share_weights=shared)
    rng = np.random.default_rng(N * 7 + F)
    P = oc.init_params(spec, rng, random_bias=True)
    B = 6
    x, e = rng.normal(size=(B, N, 9)), rng.normal(size=(B, N, 4))
    adj = (rng.uniform(size=(B, N, N)) < 0.6).astype(np.float64)
    nbr = rng.normal(size=(B, N, F)) if with_nbr else None
    M = oc.csr_to_matrix(*oc.adj_to_csr(adj), dtype=np.float64)
    q, cache = oc.forward(spec, P, x.reshape(B * N, -1), e.reshape(B * N, -1), M,
                          None if nbr is None else nbr.reshape(B * N, F))
    y = q + rng.normal(0, 1.5, size=q.shape)
    loss, dq = oc.huber_loss_and_grad(spec, q, y)
    G = oc.backward(spec, P, cache, dq)
    tp = [torch.tensor(a, requires_grad=True) for a in oc.param_arrays(P)]
    qt = _torch_forward(spec, tp, x, e, adj, nbr)
    assert np.abs(qt.detach().numpy().reshape(B * N, -1) - q).max() < 1e-11
    Y = torch.tensor(y.reshape(B, N, 4))
    per = [torch.nn.functional.huber_loss(qt[:, k], Y[:, k], delta=1.0) for k in range(N)]
    sum(per).backward()
    assert np.allclose(loss, [p.item() for p in per], atol=1e-13)
    for a, t in zip(oc.param_arrays(G), tp):
        gt = t.grad.numpy() if t.grad is not None else np.zeros(a.shape)
        assert np.abs(a - gt).max() < 1e-11


def test_backward_finite_differences():
    spec = OSpec(n_nodes=4, feat_dim=16)
    rng = np.random.default_rng(4)
    P = oc.init_params(spec, rng, random_bias=True)
    B = 3
    x, e = rng.normal(size=(B * 4, 9)), rng.normal(size=(B * 4, 4))
    M = oc.csr_to_matrix(*oc.adj_to_csr(oc.random_topology(rng, B, 4)), dtype=np.float64)
    y = rng.normal(size=(B * 4, 4))

    def total():
        q, _ = oc.forward(spec, P, x, e, M)
        return oc.huber_loss_and_grad(spec, q, y)[0].sum()

    q, cache = oc.forward(spec, P, x, e, M)
    G = oc.backward(spec, P, cache, oc.huber_loss_and_grad(spec, q, y)[1])
    for arr, g in zip(oc.param_arrays(P), oc.param_arrays(G)):
        idx = tuple(rng.integers(0, s) for s in arr.shape)
        old = arr[idx]
        h = 1e-6
        arr[idx] = old + h
        lp = total()
        arr[idx] = old - h
        lm = total()
        arr[idx] = old
        assert abs((lp - lm) / (2 * h) - g[idx]) < 1e-6


def test_keras_adam_and_huber_semantics():
    # hand-computable known answers
    y, p = np.array([[0.0, 0.0, 0.0, 0.0]]), np.array([[0.5, -2.0, 1.0, 3.0]])
    assert abs(huber_mean(y, p) - (0.125 + 1.5 + 0.5 + 2.5) / 4) < 1e-15
    assert np.allclose(huber_grad(y, p), np.array([[0.5, -1.0, 1.0, 1.0]]) / 4)
    opt = KerasAdam()
    w = [np.array([1.0, -1.0])]
    g = [np.array([0.3, -0.2])]
    opt.step(w, g)
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.5)
    m, v = 0.5 * g[0], 0.001 * g[0] ** 2
    assert np.allclose(w[0], np.array([1.0, -1.0]) - lr_t * m / (np.sqrt(v) + 1e-7), atol=1e-15)
    lim = np.sqrt(6 / (25 + 16))
    W = glorot_uniform(np.random.default_rng(0), (25, 16))
    assert W.min() >= -lim and W.max() <= lim and abs(W.std() - lim / np.sqrt(3)) < 0.02


def test_captured_agent_payload_contract():
    """Fixture captured from the reference Agent on the real simulator: pins the caller-side
    contract (BS_brain.py:441-445 adjacency, :492-493 kron, :684-692 target rule)."""
    import os
    from util import GOLDEN
    a = np.load(os.path.join(GOLDEN, 'golden_agent_n4.npz'))
    dest = a['destinations']
    A = a['init/Adjacency_Matrix']
    assert A.shape == (1, 64, 64)
    adj = A[0, ::16, ::16]
    exp = np.ones((4, 4)) - np.eye(4)
    for q in range(4):
        exp[dest[q], q] = 0
    assert np.array_equal(adj, exp) and np.array_equal(np.kron(adj, np.eye(16)), A[0])
    assert (adj.sum(axis=0) == 2).all()                        # in-degree N-2
    for k in range(1, 5):
        assert a['init/D%d_Node_Input' % k].shape == (1, 9) and a['init/D%d_Node_Input' % k][0, 8] == 10.0
        assert not a['init/D%d_Neighbor_Input' % k].any()      # always zeros (:478-490)
    # target rule
    gamma = float(a['gamma'])
    B = a['fit_y/D1_Decide_Output'].shape[0]
    for k in range(4):
        p, p_, y = a['replay_p/%d' % k], a['replay_p_next/%d' % k], a['fit_y/D%d_Decide_Output' % (k + 1)]
        changed = np.abs(y - p) > 0
        assert (changed.sum(axis=1) <= 1).all()
        assert y.shape == (B, 4)
    # every fit input equals the online-predict input (states) and is float64
    for key in [k for k in a.files if k.startswith('fit_x/')]:
        assert a[key].dtype == np.float64
        assert np.array_equal(a[key], a['replay_s/' + key[len('fit_x/'):]])
    assert 0 < gamma < 1


@pytest.mark.parametrize("N,F,L,with_nbr", [(4, 16, 2, False), (4, 16, 2, True), (6, 32, 3, False)])
def test_literal_formulation_fit_step_equals_the_compact_one(N, F, L, with_nbr):
    """The reference's own formulation (per-node weights, dict inputs, dense kron(Adj, I_F) contracted with batch_dot:
    BS_brain.py:44-76,117-208) and the compact node-row / CSR formulation are the same function: outputs, losses,
    every gradient and the weights after two Keras-Adam steps agree to rounding.  (The literal fit step is what the
    bench times as CPU legs C0 / C1.)"""
    from oracle import literal as ol
    from oracle.keras_semantics import KerasAdam
    spec = OSpec(n_nodes=N, feat_dim=F, n_mp_layers=L)
    rng = np.random.default_rng(N + F)
    P = oc.init_params(spec, rng, random_bias=True)
    B = 5
    x, e = rng.normal(size=(B, N, 9)), rng.normal(size=(B, N, 4))
    adj = (rng.uniform(size=(B, N, N)) < 0.6).astype(np.float64)
    nbr = rng.normal(size=(B, N, F)) if with_nbr else None
    feed = ol.feed_from_compact(spec, x, e, adj, nbr)
    graph = oc.adj_to_csr(adj)
    M = oc.csr_to_matrix(*graph, dtype=np.float64)
    flat = lambda a: None if a is None else a.reshape(B * N, -1)
    q, cache = oc.forward(spec, P, flat(x), flat(e), M, flat(nbr))
    ql, lcache = ol.forward_literal_cached(spec, P, feed)
    assert np.abs(np.stack(ql, 1).reshape(B * N, -1) - q).max() < 1e-11
    assert all(np.abs(a - b).max() < 1e-12 for a, b in zip(ql, ol.forward_literal(spec, P, feed)))
    y = q + rng.normal(0, 1.5, size=q.shape)
    loss, dq = oc.huber_loss_and_grad(spec, q, y)
    G = oc.backward(spec, P, cache, dq)
    yl = [y.reshape(B, N, -1)[:, k] for k in range(N)]
    Gl = ol.backward_literal(spec, P, lcache, [dq.reshape(B, N, -1)[:, k] for k in range(N)])
    for a, b in zip(oc.param_arrays(G), oc.param_arrays(Gl)):
        assert np.abs(a - b).max() < 1e-11
    # two optimizer steps
    Pc, Pl = oc.cast_params(P, np.float64), oc.cast_params(P, np.float64)
    om = oc.OracleModel(spec, Pc, dtype=np.float64)
    opt = KerasAdam()
    for _ in range(2):
        lc = om.train_step(flat(x), flat(e), graph, y, flat(nbr))
        ll = ol.train_step_literal(spec, Pl, opt, feed, yl)
        assert np.abs(lc - ll).max() < 1e-12
    for a, b in zip(oc.param_arrays(om.params), oc.param_arrays(Pl)):
        assert np.abs(a - b).max() < 1e-10


@pytest.mark.parametrize("shared", [False, True])
def test_rounding_distance_relu_gates_are_identified_and_nothing_else_is_excused(shared):
    """tests/util.assert_grads_match_oracle (how the GPU parity tests treat ReLUs whose pre-activation lies within fp32
    rounding of 0): a gradient computed with some candidate gates taken the other way is accepted, the flipped gates are
    identified exactly; a gradient that is wrong in any other way is still rejected."""
    import copy
    from util import assert_grads_match_oracle, ospec
    N, F, L, B = 5, 16, 2, 40
    spec = OSpec(n_nodes=N, feat_dim=F, n_mp_layers=L, share_weights=shared)
    rng = np.random.default_rng(11)
    P = oc.init_params(spec, rng, np.float64, random_bias=True)
    x = rng.normal(0.8, 0.4, size=(B * N, 9))
    e = rng.normal(0.9, 0.1, size=(B * N, 4))
    graph = oc.adj_to_csr(oc.random_topology(rng, B, N))
    M = oc.csr_to_matrix(*graph, dtype=np.float64)
    q, cache = oc.forward(spec, P, x, e, M)
    y = q + rng.normal(0, 1.2, size=q.shape)
    _, dq = oc.huber_loss_and_grad(spec, q, y)
    # candidates: one or two units per ReLU tensor (two in the same column of dense layer 0 when weights are shared)
    units = [(0, 0 * N + 2, 3), (1, 1 * N + 1, 5), (2, 1 * N + 4, 70), (2, 7 * N + 4, 70), (3, 2 * N + 0, 11), (4, 3 * N + 3, 7),
             (1, 9 * N + 1, 5), (0, 12 * N + 0, 13)]
    cache = dict(cache)
    cache['relu_pre'] = [p.copy() for p in cache['relu_pre']]
    for t, r, f in units:
        cache['relu_pre'][t][r, f] = 1e-9 * np.sign(cache['relu_pre'][t][r, f])         # at rounding distance of 0
    probe = {}
    ref = oc.backward(spec, P, cache, dq, probe=probe)
    step = {'os': spec, 'grads': ref, 'cache': cache, 'dq': dq, 'pre_gate': probe['pre_gate']}
    want = [units[0], units[2], units[3], units[5], units[6]]                           # the gates "the kernels" took the other way
    c2 = copy.deepcopy(cache)
    hold = [c2['h'][0], c2['h'][1], c2['z'][1], c2['z'][2], c2['z'][3]]
    for t, r, f in want:
        hold[t][r, f] = 0.0 if hold[t][r, f] > 0 else 1e-300
    flipped = oc.backward(spec, P, c2, dq)
    noise = lambda a: a * (1 + 1e-5 * rng.uniform(-1, 1, size=a.shape))                # fp32-like rounding noise
    got = [noise(a) for a in oc.params_to_list(flipped)]
    assert any(np.abs(a - b).max() > 1e-3 * np.abs(b).max() for a, b in zip(got, oc.params_to_list(ref)) if b.size and np.abs(b).max() > 0)
    used, n_cand, n_flip = assert_grads_match_oracle(got, P, step, "flipped candidates")
    assert n_cand >= len(units) and n_flip == len(want)          # (a few units of the draw are candidates by themselves)
    for a, b in zip(oc.params_to_list(used), oc.params_to_list(flipped)):
        assert np.array_equal(a, b)
    # an untouched gradient needs no resolution at all
    assert assert_grads_match_oracle([noise(a) for a in oc.params_to_list(ref)], P, step)[1:] == (0, 0)
    # ... and errors that are not flipped candidate gates are rejected: a wrong scale of one array, a non-candidate gate
    bad = [a.copy() for a in got]
    bad[5] *= 1.01
    with pytest.raises(AssertionError):
        assert_grads_match_oracle(bad, P, step, "scaled array")
    c3 = copy.deepcopy(c2)
    r_nc, f_nc = np.argwhere(c3['h'][1] > 0.5)[0]
    c3['h'][1][r_nc, f_nc] = 0.0                                                        # a healthy unit's gate closed
    with pytest.raises(AssertionError):
        assert_grads_match_oracle([noise(a) for a in oc.params_to_list(oc.backward(spec, P, c3, dq))], P, step, "non-candidate gate")


def test_sharded_oracle_step_equals_the_single_process_step():
    """oracle/parallel.py (bench.py's all-cores CPU leg): 2 worker processes over graph shards, gradients summed, one Adam
    update == OracleModel.train_step on the whole batch (up to the summation order of the gradient)."""
    from oracle.parallel import ShardedOracle
    import bench
    N, F, B = 5, 16, 24
    kw = dict(n_nodes=N, feat_dim=F, n_mp_layers=2, share_weights=False)
    x, e, adj, y = bench.synth_batch(np.random.default_rng(3), B, N)
    spec = OSpec(**kw)
    om = oc.OracleModel(spec, oc.init_params(spec, np.random.default_rng(1001), np.float32), dtype=np.float32)
    _, g_ref, _ = om.loss_and_grads(x.reshape(B * N, -1), e.reshape(B * N, -1), oc.adj_to_csr(adj), y)
    so = ShardedOracle(kw, x, e, adj, y, workers=2, seed=1001)
    try:
        assert np.array_equal(so.flat, np.concatenate([a.ravel() for a in oc.param_arrays(om.params)]))
        so.step()
        g = so.slabs.sum(axis=0)
        ref = np.concatenate([a.ravel() for a in oc.param_arrays(g_ref)])
        assert np.allclose(g, ref, rtol=1e-4, atol=1e-6 * np.abs(ref).max())
        assert so.opt.iterations == 1
    finally:
        so.close()
