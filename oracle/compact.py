"""Compact (node-row / CSR) restatement of the reference Q-network: forward, hand-written
backward, Huber loss, Keras Adam.  ORACLE -- test infrastructure only (see oracle/__init__).

Representation: all graphs of a batch are concatenated into R node rows (graph-major, node
order inside a graph = the reference's D1..DN order).  Adjacency is a CSR by DESTINATION:
for destination row q, `col_idx[row_ptr[q]:row_ptr[q+1]]` are the graph-local source nodes p
with Adj[p, q] == 1, ascending.  This equals the reference's
`AggLayer.call` (BS_brain.py:69-76) with A = kron(Adj, I_F):   agg_q = sum_p Adj[p,q] * h_p.

Weights ("Keras-shaped", with a leading slot axis S = N for the reference's per-node
weights, S = 1 for shared weights):
  params['gnn'][s]   = {'W1': [S, in_a, F], 'W2': [S, De, F], 'W3': [S, F, F], 'b': [S, F]}
       s = 0 embed (BS_brain.py:121-150), s = 1..L message passing (:154-164)
  params['dense'][i] = {'W': [S, in, out], 'b': [S, out]},  i = 0..3      (:175-200)
"""
import numpy as np
import scipy.sparse as sp

from .spec import GnnSpec
from .keras_semantics import glorot_uniform, KerasAdam, HUBER_DELTA


# ----------------------------------------------------------------------------- params
def init_params(spec: GnnSpec, rng, dtype=np.float64, random_bias=False):
    """glorot-uniform kernels, zero biases (GNNLayer.build BS_brain.py:26-41, Dense default).
    random_bias=True draws small non-zero biases so tests are sensitive to the bias path."""
    S, F, De = spec.n_slots, spec.feat_dim, spec.edge_in

    def bias(n):
        if random_bias:
            return rng.uniform(-0.1, 0.1, size=(S, n)).astype(dtype)
        return np.zeros((S, n), dtype)

    gnn = []
    for s in range(spec.n_mp_layers + 1):
        ia = spec.stage_in_a(s)
        gnn.append({'W1': glorot_uniform(rng, (S, ia, F), dtype),
                    'W2': glorot_uniform(rng, (S, De, F), dtype),
                    'W3': glorot_uniform(rng, (S, F, F), dtype),
                    'b': bias(F)})
    dense = [{'W': glorot_uniform(rng, (S, i, o), dtype), 'b': bias(o)} for i, o in spec.dense_dims]
    return {'gnn': gnn, 'dense': dense}


def params_to_list(params):
    """Keras-like flat list: stage-major, slot-minor [W1,W2,W3,b]; then dense-layer-major,
    slot-minor [kernel,bias].  N=4 reference => 80 arrays."""
    out = []
    for st in params['gnn']:
        for k in range(st['W1'].shape[0]):
            out += [st['W1'][k], st['W2'][k], st['W3'][k], st['b'][k]]
    for d in params['dense']:
        for k in range(d['W'].shape[0]):
            out += [d['W'][k], d['b'][k]]
    return out


def params_from_list(spec: GnnSpec, lst, dtype=None):
    S = spec.n_slots
    it = iter(lst)
    gnn = []
    for _ in range(spec.n_mp_layers + 1):
        per = [[next(it) for _ in range(4)] for _ in range(S)]
        gnn.append({n: np.stack([np.asarray(p[j], dtype) for p in per])
                    for j, n in enumerate(['W1', 'W2', 'W3', 'b'])})
    dense = []
    for _ in range(4):
        per = [[next(it) for _ in range(2)] for _ in range(S)]
        dense.append({n: np.stack([np.asarray(p[j], dtype) for p in per])
                      for j, n in enumerate(['W', 'b'])})
    return {'gnn': gnn, 'dense': dense}


def cast_params(params, dtype):
    return {'gnn': [{k: v.astype(dtype) for k, v in st.items()} for st in params['gnn']],
            'dense': [{k: v.astype(dtype) for k, v in d.items()} for d in params['dense']]}


def zeros_like_params(params):
    return {'gnn': [{k: np.zeros_like(v) for k, v in st.items()} for st in params['gnn']],
            'dense': [{k: np.zeros_like(v) for k, v in d.items()} for d in params['dense']]}


def param_arrays(params):
    """Leaf arrays in a fixed order (for optimizers / comparisons)."""
    out = []
    for st in params['gnn']:
        out += [st['W1'], st['W2'], st['W3'], st['b']]
    for d in params['dense']:
        out += [d['W'], d['b']]
    return out


# ----------------------------------------------------------------------------- graphs
def adj_to_csr(adj):
    """Dense Adj[B,N,N] (Adj[p,q]=1 iff p feeds q; BS_brain.py:441-445) -> CSR by destination
    with graph-local ascending sources.  Returns graph_off[B+1], row_ptr[R+1], col_idx[E]."""
    adj = np.asarray(adj)
    B, N, _ = adj.shape
    nz = adj != 0
    # destination-major: iterate q then p ascending
    dst_major = np.transpose(nz, (0, 2, 1))            # [B, q, p]
    deg = dst_major.sum(axis=2).reshape(-1)            # in-degree per destination row
    row_ptr = np.zeros(B * N + 1, np.int32)
    np.cumsum(deg, out=row_ptr[1:])
    col_idx = np.nonzero(dst_major)[2].astype(np.int32)
    graph_off = (np.arange(B + 1) * N).astype(np.int32)
    return graph_off, row_ptr, col_idx


def csr_to_matrix(graph_off, row_ptr, col_idx, dtype):
    """scipy CSR M (R x R): M[q_global, p_global] = 1, so agg = M @ h, agg^T = M.T @ g."""
    R = len(row_ptr) - 1
    deg = np.diff(row_ptr)
    row_graph = np.searchsorted(graph_off, np.arange(R), side='right') - 1
    base = np.repeat(graph_off[row_graph], deg)
    cols = col_idx.astype(np.int64) + base
    data = np.ones(len(cols), dtype)
    return sp.csr_matrix((data, cols, row_ptr.astype(np.int64)), shape=(R, R))


def random_topology(rng, B, N):
    """Reference-like topology (BS_brain.py:441-445, Environment.py:360-376): every link q
    has one receiver dest[q] != q; edge p->q iff p != q and p != dest[q] => in-degree N-2."""
    dest = rng.integers(0, N - 1, size=(B, N))
    dest = dest + (dest >= np.arange(N)[None, :])
    adj = np.ones((B, N, N)) - np.eye(N)[None]
    b_idx = np.repeat(np.arange(B), N)
    q_idx = np.tile(np.arange(N), B)
    adj[b_idx, dest.reshape(-1), q_idx] = 0.0
    return adj


# ----------------------------------------------------------------------------- slot matmul
def _slot_mm(x, W, _unused=None):
    """out[r] = x[r] @ W[slot(r)].  slots: None (S=1) or int N (row r uses slot r % N)."""
    if W.shape[0] == 1:
        return x @ W[0]
    N = W.shape[0]
    out = np.empty((x.shape[0], W.shape[2]), x.dtype)
    for k in range(N):
        out[k::N] = x[k::N] @ W[k]
    return out


def _slot_mm_t(g, W):
    """out[r] = g[r] @ W[slot(r)].T"""
    if W.shape[0] == 1:
        return g @ W[0].T
    N = W.shape[0]
    out = np.empty((g.shape[0], W.shape[1]), g.dtype)
    for k in range(N):
        out[k::N] = g[k::N] @ W[k].T
    return out


def _slot_wgrad(x, g, S):
    """dW[k] = sum_{r in slot k} x[r]^T g[r]"""
    if S == 1:
        return (x.T @ g)[None]
    return np.stack([x[k::S].T @ g[k::S] for k in range(S)])


def _slot_bgrad(g, S):
    if S == 1:
        return g.sum(axis=0)[None]
    return np.stack([g[k::S].sum(axis=0) for k in range(S)])


def _slot_bias(b, R):
    if b.shape[0] == 1:
        return b[0][None, :]
    return np.tile(b, (R // b.shape[0], 1))


# ----------------------------------------------------------------------------- forward
def forward(spec: GnnSpec, params, x, e, M, nbr=None):
    """x[R,Dn], e[R,De], M = csr_to_matrix(...), nbr[R,F] or None (the reference always feeds
    zeros: BS_brain.py:478-490).  Returns q[R,C] and the cache for backward.
    Stage wiring: BS_brain.py:147-179."""
    L, F = spec.n_mp_layers, spec.feat_dim
    R = x.shape[0]
    g0 = params['gnn'][0]
    pre = _slot_mm(x, g0['W1'], None) + _slot_mm(e, g0['W2'], None) + _slot_bias(g0['b'], R)
    if nbr is not None:
        pre = pre + _slot_mm(nbr, g0['W3'], None)
    h = [np.maximum(pre, 0)]                                    # :121 activation='relu'
    relu_pre = [pre]             # pre-activations of every ReLU (tests: which gates sit at fp32 rounding distance of 0)
    a = [M @ h[0]]                                              # :152
    for s in range(1, L + 1):
        gs = params['gnn'][s]
        u = np.concatenate([h[s - 1], x], axis=1)               # :154 concatenate([D, Node_Input])
        pre = (_slot_mm(u, gs['W1'], None) + _slot_mm(e, gs['W2'], None)
               + _slot_mm(a[s - 1], gs['W3'], None) + _slot_bias(gs['b'], R))
        h.append(np.maximum(pre, 0) if s < L else pre)          # :154 relu ... :161 linear
        if s < L:
            relu_pre.append(pre)
        a.append(M @ h[s])                                      # :159 / :166
    z = [np.concatenate([x, h[L], a[L]], axis=1)]               # :168-175
    for i in range(4):
        d = params['dense'][i]
        pre = _slot_mm(z[i], d['W'], None) + _slot_bias(d['b'], R)
        z.append(np.maximum(pre, 0) if i < 3 else pre)          # :176-179
        if i < 3:
            relu_pre.append(pre)
    q = z[4]
    cache = {'x': x, 'e': e, 'nbr': nbr, 'h': h, 'a': a, 'z': z, 'M': M, 'relu_pre': relu_pre}
    return q, cache


def huber_loss_and_grad(spec: GnnSpec, q, y, n_graphs_global=None):
    """Per-slot Huber mean (tf.losses.huber_loss, BS_brain.py:86-87, one loss per output
    :214) and dTotal/dq.  For fixed-N graphs slot k = output 'D{k+1}_Decide_Output';
    per-slot mean is over (B, C).  n_graphs_global: B of the GLOBAL batch under data
    parallelism (each rank differentiates its shard of the same global mean)."""
    N, C = spec.n_nodes, spec.n_channels
    R = q.shape[0]
    B = R // N if n_graphs_global is None else n_graphs_global
    err = q - y
    ab = np.abs(err)
    quad = np.minimum(ab, HUBER_DELTA)
    per_elem = 0.5 * quad * quad + HUBER_DELTA * (ab - quad)
    denom = q.dtype.type(B * C)
    loss = per_elem.reshape(-1, N, C).sum(axis=(0, 2)) / denom      # [N]
    dq = np.clip(err, -HUBER_DELTA, HUBER_DELTA) / denom
    return loss, dq.astype(q.dtype)


def backward(spec: GnnSpec, params, cache, dq, probe=None):
    """Hand-written reverse pass of `forward`; returns grads with the structure of params.
    probe: optional dict; receives 'pre_gate' = the gradient arriving at every ReLU BEFORE its gate is applied, in the
    order of cache['relu_pre'] (tests use it to bound what a gate at fp32 rounding distance of 0 can change)."""
    L, F, Dn = spec.n_mp_layers, spec.feat_dim, spec.node_in
    S = spec.n_slots
    x, e, nbr, h, a, z, M = (cache[k] for k in ['x', 'e', 'nbr', 'h', 'a', 'z', 'M'])
    Mt = M.T.tocsr()
    grads = zeros_like_params(params)
    pre_gate = [None] * (L + 3)
    g = dq
    for i in range(3, -1, -1):
        d = params['dense'][i]
        if i < 3:
            pre_gate[L + i] = g
            g = g * (z[i + 1] > 0)
        grads['dense'][i]['W'] = _slot_wgrad(z[i], g, S)
        grads['dense'][i]['b'] = _slot_bgrad(g, S)
        g = _slot_mm_t(g, d['W'])
    dh = g[:, Dn:Dn + F] + Mt @ g[:, Dn + F:]          # z0 = [x | h_L | a_L]
    for s in range(L, 0, -1):
        gs = params['gnn'][s]
        if s < L:
            pre_gate[s] = dh
        dpre = dh * (h[s] > 0) if s < L else dh
        u = np.concatenate([h[s - 1], x], axis=1)
        grads['gnn'][s]['W1'] = _slot_wgrad(u, dpre, S)
        grads['gnn'][s]['W2'] = _slot_wgrad(e, dpre, S)
        grads['gnn'][s]['W3'] = _slot_wgrad(a[s - 1], dpre, S)
        grads['gnn'][s]['b'] = _slot_bgrad(dpre, S)
        du = _slot_mm_t(dpre, gs['W1'])
        da = _slot_mm_t(dpre, gs['W3'])
        dh = du[:, :F] + Mt @ da
    pre_gate[0] = dh
    if probe is not None:
        probe['pre_gate'] = pre_gate
    dpre = dh * (h[0] > 0)
    grads['gnn'][0]['W1'] = _slot_wgrad(x, dpre, S)
    grads['gnn'][0]['W2'] = _slot_wgrad(e, dpre, S)
    if nbr is not None:
        grads['gnn'][0]['W3'] = _slot_wgrad(nbr, dpre, S)
    grads['gnn'][0]['b'] = _slot_bgrad(dpre, S)
    return grads


class OracleModel:
    """fit/predict on the compact representation; one `train_step` == one Keras
    `Model.fit(x, y, batch_size=len(x), epochs=1)` (BS_brain.py:218-223): forward, summed
    per-output Huber, backward, one Keras-Adam update."""

    def __init__(self, spec: GnnSpec, params, dtype=np.float32):
        self.spec = spec
        self.dtype = dtype
        self.params = cast_params(params, dtype)
        self.opt = KerasAdam()

    def predict(self, x, e, graph, nbr=None):
        M = csr_to_matrix(*graph, dtype=self.dtype)
        q, _ = forward(self.spec, self.params, x.astype(self.dtype), e.astype(self.dtype), M,
                       None if nbr is None else nbr.astype(self.dtype))
        return q

    def loss_and_grads(self, x, e, graph, y, nbr=None, n_graphs_global=None):
        M = csr_to_matrix(*graph, dtype=self.dtype)
        q, cache = forward(self.spec, self.params, x.astype(self.dtype), e.astype(self.dtype), M,
                           None if nbr is None else nbr.astype(self.dtype))
        loss, dq = huber_loss_and_grad(self.spec, q, y.astype(self.dtype), n_graphs_global)
        return loss, backward(self.spec, self.params, cache, dq), q

    def train_step(self, x, e, graph, y, nbr=None):
        loss, grads, _ = self.loss_and_grads(x, e, graph, y, nbr)
        self.opt.step(param_arrays(self.params), param_arrays(grads))
        return loss
