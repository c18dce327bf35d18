"""Literal restatement of the reference Keras graph in the reference's OWN formulation:
one weight set per node, dict-keyed inputs `D{k}_Node_Input` ..., dense
`Adjacency_Matrix = kron(Adj, I_F)` of shape [B, N*F, N*F] contracted with batch_dot.
ORACLE -- test infrastructure only.  Follows /root/reference/BS_brain.py line by line:
GNNLayer.call :44-51, AggLayer.call :69-76, _create_model wiring :117-208.
"""
import numpy as np


def gnn_layer(a, b, c, W1, W2, W3, bias, relu):
    """GNNLayer.call (BS_brain.py:47-50): act(K.dot(a,W1)+K.dot(b,W2)+K.dot(c,W3)+bias)."""
    out = a @ W1 + b @ W2 + c @ W3
    out = out + bias
    return np.maximum(out, 0) if relu else out


def agg_layer(D_list, A, F):
    """AggLayer.call (BS_brain.py:71-76): D = concat(D1..DN); out = batch_dot(D, A, axes=[1,1])
    i.e. out[b, j] = sum_i D[b, i] * A[b, i, j]; split back into N blocks of width F."""
    D = np.concatenate(D_list, axis=-1)
    out = np.einsum('bi,bij->bj', D, A)
    return [out[:, k * F:(k + 1) * F] for k in range(len(D_list))]


def forward_literal(spec, params, feed):
    """feed: dict with the reference's input names (BS_brain.py:117-144).
    params: oracle.compact structure with S == N (per-node weights).
    Returns list of N arrays [B, C] ordered like Model(outputs=...) (BS_brain.py:208)."""
    N, F, L = spec.n_nodes, spec.feat_dim, spec.n_mp_layers
    assert params['gnn'][0]['W1'].shape[0] == N, "literal formulation has per-node weights"
    A = feed['Adjacency_Matrix']
    node = [feed['D%d_Node_Input' % (k + 1)] for k in range(N)]
    edge = [feed['D%d_Edge_Input' % (k + 1)] for k in range(N)]
    nbr = [feed['D%d_Neighbor_Input' % (k + 1)] for k in range(N)]
    g = params['gnn']
    D = [gnn_layer(node[k], edge[k], nbr[k], g[0]['W1'][k], g[0]['W2'][k], g[0]['W3'][k],
                   g[0]['b'][k], True) for k in range(N)]                       # :147-150
    Agg = agg_layer(D, A, F)                                                    # :152
    for s in range(1, L + 1):
        D = [gnn_layer(np.concatenate([D[k], node[k]], axis=-1), edge[k], Agg[k],
                       g[s]['W1'][k], g[s]['W2'][k], g[s]['W3'][k], g[s]['b'][k], s < L)
             for k in range(N)]                                                 # :154-157 / :161-164
        Agg = agg_layer(D, A, F)                                                # :159 / :166
    outs = []
    for k in range(N):
        z = np.concatenate([node[k], np.concatenate([D[k], Agg[k]], axis=-1)], axis=-1)  # :168,:175
        for i in range(4):
            d = params['dense'][i]
            z = z @ d['W'][k] + d['b'][k]
            if i < 3:
                z = np.maximum(z, 0)                                            # :176-179
        outs.append(z)
    return outs


def feed_from_compact(spec, x, e, adj, nbr=None):
    """Build the reference's dict payload (float64, kron adjacency) from compact arrays
    x[B,N,Dn], e[B,N,De], adj[B,N,N]  -- the inverse of what Agent.replay builds
    (BS_brain.py:585-651)."""
    N, F = spec.n_nodes, spec.feat_dim
    B = x.shape[0]
    feed = {}
    for k in range(N):
        feed['D%d_Node_Input' % (k + 1)] = np.ascontiguousarray(x[:, k, :], dtype=np.float64)
        feed['D%d_Edge_Input' % (k + 1)] = np.ascontiguousarray(e[:, k, :], dtype=np.float64)
        feed['D%d_Neighbor_Input' % (k + 1)] = (np.zeros((B, F)) if nbr is None
                                                else np.ascontiguousarray(nbr[:, k, :], dtype=np.float64))
    feed['Adjacency_Matrix'] = np.kron(adj, np.eye(F))                          # :603
    return feed


# ----------------------------------------------------------------------------- fit step in the same formulation
def _batch_dot(D, A):
    """K.batch_dot(D[B,M], A[B,M,M], axes=[1,1]) (BS_brain.py:73) as a batched matrix product."""
    return np.matmul(D[:, None, :], A)[:, 0, :]


def forward_literal_cached(spec, params, feed):
    """forward_literal keeping what the reverse pass needs (same arithmetic, batched-matmul contraction)."""
    N, F, L = spec.n_nodes, spec.feat_dim, spec.n_mp_layers
    A = feed['Adjacency_Matrix']
    node = [feed['D%d_Node_Input' % (k + 1)] for k in range(N)]
    edge = [feed['D%d_Edge_Input' % (k + 1)] for k in range(N)]
    nbr = [feed['D%d_Neighbor_Input' % (k + 1)] for k in range(N)]
    g = params['gnn']
    D = [[gnn_layer(node[k], edge[k], nbr[k], g[0]['W1'][k], g[0]['W2'][k], g[0]['W3'][k], g[0]['b'][k], True)
          for k in range(N)]]
    split = lambda out: [out[:, k * F:(k + 1) * F] for k in range(N)]
    Agg = [split(_batch_dot(np.concatenate(D[0], axis=-1), A))]
    for s in range(1, L + 1):
        D.append([gnn_layer(np.concatenate([D[s - 1][k], node[k]], axis=-1), edge[k], Agg[s - 1][k],
                            g[s]['W1'][k], g[s]['W2'][k], g[s]['W3'][k], g[s]['b'][k], s < L) for k in range(N)])
        Agg.append(split(_batch_dot(np.concatenate(D[s], axis=-1), A)))
    zs = []
    for k in range(N):
        z = [np.concatenate([node[k], D[L][k], Agg[L][k]], axis=-1)]
        for i in range(4):
            d = params['dense'][i]
            pre = z[i] @ d['W'][k] + d['b'][k]
            z.append(np.maximum(pre, 0) if i < 3 else pre)
        zs.append(z)
    return [z[4] for z in zs], {'A': A, 'node': node, 'edge': edge, 'nbr': nbr, 'D': D, 'Agg': Agg, 'z': zs}


def backward_literal(spec, params, cache, dq):
    """Reverse pass of forward_literal_cached; dq = list of N arrays [B, C].  Returns gradients in the structure of
    `params` (what TF's autodiff of the reference graph computes: BS_brain.py:147-179 in reverse)."""
    N, F, L, Dn = spec.n_nodes, spec.feat_dim, spec.n_mp_layers, spec.node_in
    A, node, edge, nbr, D, Agg, zs = (cache[k] for k in ('A', 'node', 'edge', 'nbr', 'D', 'Agg', 'z'))
    grads = {'gnn': [{k: np.zeros_like(v) for k, v in gs.items()} for gs in params['gnn']],
             'dense': [{k: np.zeros_like(v) for k, v in d.items()} for d in params['dense']]}
    At = np.transpose(A, (0, 2, 1))
    dD, dAgg = [None] * N, [None] * N
    for k in range(N):
        dz = dq[k]
        for i in range(3, -1, -1):
            d = params['dense'][i]
            grads['dense'][i]['W'][k] = zs[k][i].T @ dz
            grads['dense'][i]['b'][k] = dz.sum(axis=0)
            dz = dz @ d['W'][k].T
            if i > 0:
                dz = dz * (zs[k][i] > 0)
        dD[k], dAgg[k] = dz[:, Dn:Dn + F], dz[:, Dn + F:]
    for s in range(L, -1, -1):
        back = _batch_dot(np.concatenate(dAgg, axis=-1), At)             # transpose of AggLayer.call
        gs = params['gnn'][s]
        for k in range(N):
            dpre = dD[k] + back[:, k * F:(k + 1) * F]
            if s < L:
                dpre = dpre * (D[s][k] > 0)
            a_in = np.concatenate([D[s - 1][k], node[k]], axis=-1) if s > 0 else node[k]
            c_in = Agg[s - 1][k] if s > 0 else nbr[k]
            grads['gnn'][s]['W1'][k] = a_in.T @ dpre
            grads['gnn'][s]['W2'][k] = edge[k].T @ dpre
            grads['gnn'][s]['W3'][k] = c_in.T @ dpre
            grads['gnn'][s]['b'][k] = dpre.sum(axis=0)
            if s > 0:
                dD[k] = dpre @ gs['W1'][k][:F].T
                dAgg[k] = dpre @ gs['W3'][k].T
    return grads


def train_step_literal(spec, params, opt, feed, y):
    """One Model.fit step (BS_brain.py:218-223) in the reference formulation: y = list of N target arrays [B, C].
    params are updated in place by `opt` (oracle.keras_semantics.KerasAdam).  Returns the N per-output Huber means."""
    from .keras_semantics import huber_mean, huber_grad
    from .compact import param_arrays
    q, cache = forward_literal_cached(spec, params, feed)
    loss = np.array([huber_mean(y[k], q[k]) for k in range(spec.n_nodes)])
    grads = backward_literal(spec, params, cache, [huber_grad(y[k], q[k]) for k in range(spec.n_nodes)])
    opt.step(param_arrays(params), param_arrays(grads))
    return loss
