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
