"""Host-side data formats either side of the hot path (numpy only, no GPU needed).

* dict payload of the reference (`D{k}_Node_Input`, `D{k}_Edge_Input`, `D{k}_Neighbor_Input`,
  `Adjacency_Matrix = kron(Adj, I_F)`; BS_brain.py:495-504, :642-651)  ->  compact packed batch
  (xe[R][16] fp32 + CSR by destination) consumed by the C ABI (include/v2xgnn.h).
* Keras-shaped weight lists  <->  the engine's flat parameter layout.
"""
import numpy as np

from .spec import GnnSpec, XE_WIDTH


# ------------------------------------------------------------------------------ inputs
def pack_xe(node, edge):
    """node[R,Dn], edge[R,De] -> xe[R,16] fp32 = [node | edge | 0 pad] (Keras casts the float64
    feed to floatx=float32 the same way)."""
    node = np.asarray(node)
    edge = np.asarray(edge)
    R, Dn = node.shape
    De = edge.shape[1]
    if Dn + De > XE_WIDTH:
        raise ValueError("node+edge width %d exceeds the packed row width %d" % (Dn + De, XE_WIDTH))
    xe = np.zeros((R, XE_WIDTH), np.float32)
    xe[:, :Dn] = node
    xe[:, Dn:Dn + De] = edge
    return xe


def adj_to_csr(adj):
    """Dense Adj[B,N,N] with Adj[p,q] != 0 iff node p feeds node q (BS_brain.py:441-445) ->
    CSR by destination: row_ptr[R+1] (global edge offsets), col_idx[E] (graph-local sources,
    ascending).  Equivalent to AggLayer with kron(Adj, I_F) (BS_brain.py:72-76)."""
    adj = np.asarray(adj)
    if adj.ndim != 3 or adj.shape[1] != adj.shape[2]:
        raise ValueError("adjacency must be [B,N,N], got %r" % (adj.shape,))
    B, N, _ = adj.shape
    vals = adj[adj != 0]
    if vals.size and not np.all(vals == 1):
        raise ValueError("adjacency entries must be 0 or 1 (the engine aggregates unweighted edges)")
    dst_major = np.transpose(adj != 0, (0, 2, 1))            # [B, q, p]
    deg = dst_major.sum(axis=2)
    row_ptr = np.zeros(B * N + 1, np.int32)
    np.cumsum(deg.reshape(-1), out=row_ptr[1:])
    col_idx = np.nonzero(dst_major)[2].astype(np.int32)
    max_edges = int(deg.sum(axis=1).max()) if B else 0
    return row_ptr, col_idx, max_edges


def kron_to_adj(A, F, validate=True):
    """Compress the reference's dense `Adjacency_Matrix` [B, N*F, N*F] = kron(Adj, I_F)
    (BS_brain.py:492-493, :603) back to Adj[B,N,N] by striding.  With validate=True the
    Kronecker structure is checked (anything else is not expressible as a graph)."""
    A = np.asarray(A)
    if A.ndim != 3 or A.shape[1] != A.shape[2] or A.shape[1] % F:
        raise ValueError("Adjacency_Matrix must be [B, N*F, N*F] with F=%d, got %r" % (F, A.shape))
    adj = A[:, ::F, ::F]
    if validate:
        N = A.shape[1] // F
        blocks = A.reshape(A.shape[0], N, F, N, F)
        diag = np.einsum('bpiqi->bpqi', blocks)
        if not (np.all(diag == adj[..., None]) and np.count_nonzero(A) == np.count_nonzero(adj) * F):
            raise ValueError("Adjacency_Matrix is not kron(Adj, I_F)")
    return adj


class PackedBatch(object):
    """Compact batch: xe[R,16] f32, optional nbr[R,F] f32, CSR, sizes.  Arrays are numpy (host)."""

    def __init__(self, n_graphs, n_nodes, xe, row_ptr, col_idx, max_edges, nbr=None, graph_off=None,
                 max_nodes=None):
        self.n_graphs = int(n_graphs)
        self.n_nodes = int(n_nodes)
        self.xe = np.ascontiguousarray(xe, np.float32)
        self.row_ptr = np.ascontiguousarray(row_ptr, np.int32)
        self.col_idx = np.ascontiguousarray(col_idx, np.int32)
        self.nbr = None if nbr is None else np.ascontiguousarray(nbr, np.float32)
        self.graph_off = None if graph_off is None else np.ascontiguousarray(graph_off, np.int32)
        self.max_edges = int(max_edges)
        self.max_nodes = int(max_nodes if max_nodes is not None else n_nodes)
        self.n_rows = self.xe.shape[0]
        self.n_edges = int(self.row_ptr[-1])
        if self.row_ptr.shape[0] != self.n_rows + 1:
            raise ValueError("row_ptr must have n_rows+1 entries")
        if self.col_idx.shape[0] != self.n_edges:
            raise ValueError("col_idx length %d != row_ptr[-1] %d" % (self.col_idx.shape[0], self.n_edges))

    @classmethod
    def from_dense(cls, x, e, adj, nbr=None):
        """x[B,N,Dn], e[B,N,De], adj[B,N,N], nbr[B,N,F] or None."""
        x = np.asarray(x)
        B, N = x.shape[0], x.shape[1]
        row_ptr, col_idx, max_edges = adj_to_csr(adj)
        xe = pack_xe(x.reshape(B * N, -1), np.asarray(e).reshape(B * N, -1))
        nb = None if nbr is None else np.asarray(nbr).reshape(B * N, -1)
        return cls(B, N, xe, row_ptr, col_idx, max_edges, nbr=nb)

    def shard(self, rank, world):
        """Contiguous shard of whole graphs for data parallelism (fixed-size graphs)."""
        if self.graph_off is not None:
            raise NotImplementedError("sharding of variable-size batches")
        if self.n_graphs % world:
            raise ValueError("batch %d not divisible by world size %d" % (self.n_graphs, world))
        b = self.n_graphs // world
        g0, g1 = rank * b, (rank + 1) * b
        r0, r1 = g0 * self.n_nodes, g1 * self.n_nodes
        e0, e1 = int(self.row_ptr[r0]), int(self.row_ptr[r1])
        deg_g = np.add.reduceat(np.diff(self.row_ptr[r0:r1 + 1]), np.arange(0, r1 - r0, self.n_nodes))
        return PackedBatch(b, self.n_nodes, self.xe[r0:r1], self.row_ptr[r0:r1 + 1] - e0,
                           self.col_idx[e0:e1], int(deg_g.max()),
                           nbr=None if self.nbr is None else self.nbr[r0:r1])


def feed_to_arrays(spec: GnnSpec, feed, validate_adjacency=True):
    """Reference dict payload -> (x[B,N,Dn], e[B,N,De], nbr[B,N,F] or None, adj[B,N,N]).
    Raises ValueError on a missing / mis-shaped key like Keras' input standardisation."""
    N, F, Dn, De = spec.n_nodes, spec.feat_dim, spec.node_in, spec.edge_in
    xs, es, ns = [], [], []
    B = None
    for k in range(1, N + 1):
        for kind, width, dst in (('Node', Dn, xs), ('Edge', De, es), ('Neighbor', F, ns)):
            name = 'D%d_%s_Input' % (k, kind)
            if name not in feed:
                raise ValueError('No data provided for "%s". Need data for each key' % name)
            arr = np.asarray(feed[name])
            if arr.ndim != 2 or arr.shape[1] != width:
                raise ValueError('Error when checking input: expected %s to have shape (%d,) but got array '
                                 'with shape %r' % (name, width, arr.shape[1:]))
            if B is None:
                B = arr.shape[0]
            elif arr.shape[0] != B:
                raise ValueError('All input arrays should have the same number of samples')
            dst.append(arr)
    if 'Adjacency_Matrix' not in feed:
        raise ValueError('No data provided for "Adjacency_Matrix". Need data for each key')
    A = np.asarray(feed['Adjacency_Matrix'])
    if A.shape != (B, N * F, N * F):
        raise ValueError('Error when checking input: expected Adjacency_Matrix to have shape (%d, %d) but got '
                         'array with shape %r' % (N * F, N * F, A.shape[1:]))
    x = np.stack(xs, axis=1)
    e = np.stack(es, axis=1)
    nbr = np.stack(ns, axis=1)
    if not nbr.any():
        nbr = None            # the reference always feeds zeros (BS_brain.py:478-490): skip that GEMM
    adj = kron_to_adj(A, F, validate=validate_adjacency)
    return x, e, nbr, adj


# ------------------------------------------------------------------------------ weights
def keras_list_shapes(spec: GnnSpec):
    """Shapes of the Keras-shaped weight list: stage-major, slot-minor [W1,W2,W3,bias], then
    Dense-layer-major, slot-minor [kernel,bias].  80 arrays for the reference (N=4)."""
    F, De, S = spec.feat_dim, spec.edge_in, spec.n_slots
    shapes = []
    for s in range(spec.n_mp_layers + 1):
        for _ in range(S):
            shapes += [(spec.stage_in_a(s), F), (De, F), (F, F), (F,)]
    for i, o in spec.dense_dims:
        for _ in range(S):
            shapes += [(i, o), (o,)]
    return shapes


def keras_list_to_flat(spec: GnnSpec, weights):
    """Keras-shaped list -> flat fp32 vector in the engine layout (include/v2xgnn.h):
    per layer, per slot: vstack(W1, W2, W3) then bias; Dense-0 rows permuted from Keras'
    [x | h | agg] (BS_brain.py:175) to the engine's [h | x | agg]."""
    shapes = keras_list_shapes(spec)
    if len(weights) != len(shapes):
        raise ValueError("You called `set_weights(weights)` with a weight list of length %d, but the model "
                         "was expecting %d weights." % (len(weights), len(shapes)))
    for w, shp in zip(weights, shapes):
        if tuple(np.shape(w)) != tuple(shp):
            raise ValueError("Layer weight shape %r not compatible with provided weight shape %r"
                             % (shp, tuple(np.shape(w))))
    F, Dn, S = spec.feat_dim, spec.node_in, spec.n_slots
    out = []
    it = iter(weights)
    for _ in range(spec.n_mp_layers + 1):
        for _ in range(S):
            W1, W2, W3, b = (np.asarray(next(it), np.float32) for _ in range(4))
            out += [W1.ravel(), W2.ravel(), W3.ravel(), b.ravel()]
    for li in range(4):
        for _ in range(S):
            W, b = (np.asarray(next(it), np.float32) for _ in range(2))
            if li == 0:
                W = np.concatenate([W[Dn:Dn + F], W[:Dn], W[Dn + F:]], axis=0)
            out += [W.ravel(), b.ravel()]
    flat = np.concatenate(out).astype(np.float32)
    assert flat.size == spec.n_params
    return flat


def flat_to_keras_list(spec: GnnSpec, flat):
    flat = np.asarray(flat, np.float32)
    if flat.size != spec.n_params:
        raise ValueError("flat parameter vector has %d entries, expected %d" % (flat.size, spec.n_params))
    F, Dn, De, S = spec.feat_dim, spec.node_in, spec.edge_in, spec.n_slots
    out = []
    pos = 0

    def take(shape):
        nonlocal pos
        n = int(np.prod(shape))
        a = flat[pos:pos + n].reshape(shape).copy()
        pos += n
        return a

    for s in range(spec.n_mp_layers + 1):
        ia = spec.stage_in_a(s)
        for _ in range(S):
            out += [take((ia, F)), take((De, F)), take((F, F)), take((F,))]
    for li, (i, o) in enumerate(spec.dense_dims):
        for _ in range(S):
            W = take((i, o))
            if li == 0:
                W = np.concatenate([W[F:F + Dn], W[:F], W[F + Dn:]], axis=0)
            out += [W, take((o,))]
    assert pos == flat.size
    return out
