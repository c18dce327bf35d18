"""Shared helpers of the test-suite (tests may use the oracle; the product may not)."""
import itertools
import os

import numpy as np

from oracle.spec import GnnSpec as OSpec
from oracle import compact as oc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# fp32 parity tolerances of the HIP path against the float64 oracle evaluated on the same
# fp32-rounded inputs and weights (north_star: "within a stated fp32 tolerance").
# forward:  |got - ref| <= FWD_RTOL*|ref| + FWD_ATOL*max(1, max|ref|)   (sums of ~150 fp32 products
# with cancellation: the absolute part has to follow the magnitude of the terms)
FWD_RTOL, FWD_ATOL = 2e-4, 2e-6
GRAD_RTOL, GRAD_ATOL_REL = 5e-4, 2e-5     # atol = GRAD_ATOL_REL * max|reference|


def ospec(spec):
    return OSpec(n_nodes=spec.n_nodes, n_channels=spec.n_channels, feat_dim=spec.feat_dim,
                 n_mp_layers=spec.n_mp_layers, share_weights=spec.share_weights)


def random_inputs(rng, B, N, C=4, ref_topology=True, density=0.5):
    """Feature statistics of SURVEY.md 8(d3) (measured from the reference simulator)."""
    x = np.concatenate([rng.normal(0.84, 0.39, size=(B, N, C)), rng.normal(0.60, 0.21, size=(B, N, C)),
                        np.full((B, N, 1), 10.0)], axis=2)
    e = rng.normal(0.88, 0.11, size=(B, N, C))
    if ref_topology and N > 2:
        adj = oc.random_topology(rng, B, N)
    else:
        adj = (rng.uniform(size=(B, N, N)) < density).astype(np.float64)
    return x.astype(np.float32), e.astype(np.float32), adj


def fixed_indegree_adj(rng, B, N, degree):
    """adj[b, p, q] = 1 for exactly `degree` sources p != q of every destination q (sparse interference graphs: the
    degree-aware walks of the fused kernels)."""
    adj = np.zeros((B, N, N), np.float64)
    for b in range(B):
        for q in range(N):
            others = np.delete(np.arange(N), q)
            adj[b, rng.choice(others, size=min(degree, N - 1), replace=False), q] = 1.0
    return adj


def f32_params(spec, rng, random_bias=True):
    """Oracle params whose values are exactly representable in fp32 (kept as float64 arrays)."""
    P = oc.init_params(ospec(spec), rng, np.float64, random_bias=random_bias)
    return oc.cast_params(oc.cast_params(P, np.float32), np.float64)


def assert_close(got, ref, rtol, atol, what=""):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = np.abs(got - ref)
    bound = atol + rtol * np.abs(ref)
    bad = err > bound
    assert not bad.any(), "%s: %d/%d out of tolerance, max err %.3e (ref scale %.3e)" % (
        what, bad.sum(), bad.size, err.max(), np.abs(ref).max())


def assert_fwd_close(got, ref, what=""):
    scale = max(1.0, float(np.abs(ref).max()))
    assert_close(got, ref, FWD_RTOL, FWD_ATOL * scale, what)


def assert_grad_close(got, ref, what=""):
    scale = float(np.abs(ref).max()) or 1.0
    assert_close(got, ref, GRAD_RTOL, GRAD_ATOL_REL * scale, what)


def golden_forward_cases():
    f = np.load(os.path.join(GOLDEN, "golden_forward_n4.npz"))
    cases = sorted({k.split('/')[0] for k in f.files})
    return f, cases


def golden_keras_list(f, case, tag):
    """Weights of a golden case, re-ordered from the reference's creation order
    (GNN: stage-major node-minor; Dense: node-major layer-minor) to the Keras-shaped list
    order used by the engine and the oracle (Dense: layer-major node-minor)."""
    ws = [f['%s/%s/w/%03d' % (case, tag, i)] for i in range(80)]
    lst = ws[:48]
    for i in range(4):
        for k in range(4):
            lst += [ws[48 + k * 8 + i * 2], ws[48 + k * 8 + i * 2 + 1]]
    return lst


def golden_feed(f, case):
    pre = case + '/in/'
    return {k[len(pre):]: f[k] for k in f.files if k.startswith(pre)}


# ------------------------------------------------------------------------------------------------ ReLU gates at rounding distance of 0
# A ReLU whose float64 pre-activation lies within fp32 rounding error of 0 is gated one way by the oracle and possibly
# the other way by fp32 arithmetic: the unit's whole gradient contribution appears / disappears, which no rounding
# tolerance covers.  That is conditioning of the DRAW, not of the kernels.  Nothing is redrawn and no tolerance is
# widened; instead the condition is made explicit:
#   (a) the CANDIDATE units are found from the oracle's own pre-activations (|pre| <= 4e-6 x the sum of the magnitudes of
#       the terms it adds up, sum_k |in_k| |W_kf| + |b_f|; AMBIG_RTOLS);
#   (b) if the kernels' gradient differs from the oracle's, the candidates whose gate the kernels took the other way are
#       identified from the residual in the unit's OWN weight column (flipping unit (row r, feature f) changes column f of
#       its layer's weight gradient by exactly +-(pre-gate gradient) x (the layer's input row r), bias included): the
#       smallest set of candidates that brings the column within the rounding tolerance, layer by layer in the order of
#       the reverse pass;
#   (c) the oracle's reverse pass is repeated with exactly those gates taken the kernels' way, and EVERY element of EVERY
#       gradient array must then agree within the plain rounding tolerance.
# A kernel error is not a linear combination of a handful of such columns: it fails (c) as before.
AMBIG_RTOLS = (4e-6, 2e-5)      # the candidate set: first the tight one, the wider one only if that does not explain the residual


def _graph_of_row(graph_off, R):
    return np.searchsorted(np.asarray(graph_off), np.arange(R), side='right') - 1


def _relu_layer_inputs(os_, cache, t):
    """ReLU tensor t (order of cache['relu_pre']) -> (kind, index, [(parameter name, input tensor)], post-ReLU tensor)."""
    L = os_.n_mp_layers
    if t >= L:
        i = t - L
        return 'dense', i, [('W', cache['z'][i])], cache['z'][i + 1]
    if t == 0:
        return 'gnn', 0, [('W1', cache['x']), ('W2', cache['e'])], cache['h'][0]
    return 'gnn', t, [('W1', np.concatenate([cache['h'][t - 1], cache['x']], axis=1)), ('W2', cache['e']),
                      ('W3', cache['a'][t - 1])], cache['h'][t]


def _slot_bias_abs(b, R):
    return np.abs(b[0])[None, :] if b.shape[0] == 1 else np.tile(np.abs(b), (R // b.shape[0], 1))


def resolve_relu_gates(os_, P, cache, dq, got, ref, pre_gate, rtol=AMBIG_RTOLS[0]):
    """got / ref: gradients with the structure of the parameters (kernels / oracle).  -> (gradients of the oracle with the
    candidate gates the kernels took the other way flipped, number of candidate units, number flipped).
    The ReLU tensors are visited in the order of the reverse pass (a gate only influences the gradients of its own layer and
    of the layers below it), the oracle's reverse pass being repeated after every layer in which gates were flipped, so that
    a unit's column is always compared with an oracle that already has the layers above it the kernels' way."""
    S = os_.n_slots
    c2 = dict(cache)
    c2['h'] = [v.copy() for v in cache['h']]
    c2['z'] = [v.copy() for v in cache['z']]
    n_cand = n_flip = 0
    for t in range(len(cache['relu_pre']) - 1, -1, -1):
        pre = cache['relu_pre'][t]
        kind, li, ins, post = _relu_layer_inputs(os_, c2, t)
        # a pre-activation is uncertain relative to the size of the TERMS it sums (rows of a 126-link graph carry
        # values 1e4 times those of a 2-link graph in the same ragged batch), not relative to the layer's largest value
        mag = np.abs(_slot_bias_abs(P[kind][li]['b'], pre.shape[0]))
        for name, a in ins:
            mag = mag + oc._slot_mm(np.abs(a), np.abs(P[kind][li][name]))
        rr, ff = np.nonzero(np.abs(pre) <= rtol * mag)
        n_cand += rr.size
        if rr.size == 0:
            continue
        slot = rr % S if S > 1 else np.zeros_like(rr)
        delta = np.where(post[rr, ff] > 0, -1.0, 1.0) * pre_gate[t][rr, ff]    # what flipping the unit adds to db[slot][f]
        key = slot * pre.shape[1] + ff
        flips = []
        for kq in np.unique(key):
            sel = np.nonzero(key == kq)[0]
            k, f = int(slot[sel[0]]), int(ff[sel[0]])
            resid = np.concatenate([got[kind][li][n][k][:, f] - ref[kind][li][n][k][:, f] for n, _ in ins]
                                   + [[got[kind][li]['b'][k][f] - ref[kind][li]['b'][k][f]]])
            V = np.stack([delta[j] * np.concatenate([a[rr[j]] for _, a in ins] + [[1.0]]) for j in sel], axis=1)
            refcol = np.concatenate([ref[kind][li][n][k][:, f] for n, _ in ins] + [[ref[kind][li]['b'][k][f]]])
            scale = np.concatenate([np.full(ref[kind][li][n][k].shape[0], np.abs(ref[kind][li][n]).max()) for n, _ in ins]
                                   + [[np.abs(ref[kind][li]['b']).max()]])
            tol = GRAD_RTOL * np.abs(refcol) + GRAD_ATOL_REL * scale
            # the SMALLEST set of candidates that brings the column within the rounding tolerance (none, if it already
            # is: rows' input vectors are nearly parallel, so "whatever reduces the residual most" would pick at random)
            best, best_norm = (), np.inf
            for size in range(0, min(sel.size, 3) + 1):
                hit = False
                for sub in itertools.combinations(range(sel.size), size):
                    d = resid - V[:, list(sub)].sum(axis=1)
                    if (np.abs(d) <= tol).all():
                        best, hit = sub, True
                        break
                    nd = float(np.linalg.norm(d / np.maximum(tol, 1e-300)))
                    if nd < best_norm:
                        best, best_norm = sub, nd
                if hit:
                    break
            flips += [(int(rr[sel[j]]), int(ff[sel[j]])) for j in best]
        if flips:
            for r, f in flips:
                post[r, f] = 0.0 if post[r, f] > 0 else 1e-300                 # gate closed / open; the value stays ~0
            probe = {}
            ref = oc.backward(os_, P, c2, dq, probe=probe)
            pre_gate = probe['pre_gate']
            n_flip += len(flips)
    return ref, n_cand, n_flip


def oracle_step(spec, P, x, e, graph, y, q_at=None, n_denominator=None):
    """Float64 oracle of one fit step's forward / Huber / backward on node-row inputs.  q_at: differentiate the loss at
    this q (the kernels' own fp32 output -- forward parity is asserted separately) instead of at the oracle's.
    -> dict(q, loss, grads (structure of the parameters), cache, dq, pre_gate)"""
    os_ = ospec(spec)
    M = oc.csr_to_matrix(*graph, dtype=np.float64)
    q_ref, cache = oc.forward(os_, P, np.asarray(x, np.float64), np.asarray(e, np.float64), M)
    q_use = q_ref if q_at is None else np.asarray(q_at, np.float64)
    if getattr(spec, 'variable_graphs', False):
        R, C = q_use.shape
        den = float((n_denominator or R) * C)
        err = q_use - y
        a_ = np.abs(err)
        quad = np.minimum(a_, 1.0)
        loss = np.array([(0.5 * quad * quad + (a_ - quad)).sum() / den])
        dq = np.clip(err, -1.0, 1.0) / den
    else:
        loss, dq = oc.huber_loss_and_grad(os_, q_use, np.asarray(y, np.float64), n_denominator)
    probe = {}
    g = oc.backward(os_, P, cache, dq, probe=probe)
    return {'q': q_ref, 'loss': loss, 'grads': g, 'cache': cache, 'dq': dq, 'pre_gate': probe['pre_gate'], 'os': os_}


def _grads_within_tolerance(got_list, ref_list):
    """-> None, or (array index, number of elements out of tolerance, size, max error, reference scale)."""
    for i, (a, b) in enumerate(zip(got_list, ref_list)):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        assert a.shape == b.shape, (i, a.shape, b.shape)
        scale = float(np.abs(b).max()) or 1.0
        err = np.abs(a - b)
        bad = err > GRAD_RTOL * np.abs(b) + GRAD_ATOL_REL * scale
        if bad.any():
            return i, int(bad.sum()), bad.size, float(err.max()), scale
    return None


MAX_GATE_FLIPS = 8      # gates a check may hand to the kernels; observed 0-3 per draw at 22 M ReLU units (assert it)


def assert_grads_match_oracle(got_list, P, step, what="", max_flips=MAX_GATE_FLIPS):
    """got_list: the kernels' gradient as a Keras-shaped list; step: oracle_step(...).  Every element of every array within
    GRAD_RTOL / GRAD_ATOL_REL of the oracle's gradient -- if need be of the oracle with the explicitly identified
    rounding-distance ReLU gates taken the kernels' way (header above).  -> (oracle gradients used (structure of the
    parameters), number of candidate units, number flipped)."""
    os_, ref0 = step['os'], step['grads']
    ref, n_cand, n_flip = ref0, 0, 0
    bad = _grads_within_tolerance(got_list, oc.params_to_list(ref0))
    if bad is not None:
        got = oc.params_from_list(os_, got_list, np.float64)
        n_units = sum(p.size for p in step['cache']['relu_pre'])
        for rtol in AMBIG_RTOLS:
            ref, n_cand, n_flip = resolve_relu_gates(os_, P, step['cache'], step['dq'], got, ref0, step['pre_gate'], rtol)
            assert n_cand <= 1e-3 * n_units, (what, "too many ReLU units at rounding distance of 0", n_cand, n_units)
            bad = _grads_within_tolerance(got_list, oc.params_to_list(ref))
            if bad is None:
                break
        assert bad is None, "%s: gradient array %d: %d/%d out of tolerance, max err %.3e (ref scale %.3e); %d of %d " \
            "candidate ReLU gates flipped" % ((what,) + bad + (n_flip, n_cand))
        assert n_flip <= max_flips, "%s: %d ReLU gates (of %d candidates) had to be taken the kernels' way -- more than " \
            "rounding at the gate explains" % (what, n_flip, n_cand)
    return ref, n_cand, n_flip
