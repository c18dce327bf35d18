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

    def __init__(self, n_graphs, n_nodes, xe, row_ptr, col_idx, max_edges=None, nbr=None, graph_off=None,
                 max_nodes=None):
        self.n_graphs = int(n_graphs)
        self.n_nodes = int(n_nodes)
        self.xe = np.ascontiguousarray(xe, np.float32)
        self.row_ptr = np.ascontiguousarray(row_ptr, np.int32)
        self.col_idx = np.ascontiguousarray(col_idx, np.int32)
        self.nbr = None if nbr is None else np.ascontiguousarray(nbr, np.float32)
        self.graph_off = None if graph_off is None else np.ascontiguousarray(graph_off, np.int32)
        self.n_rows = self.xe.shape[0]
        self.n_edges = int(self.row_ptr[-1])
        if self.row_ptr.shape[0] != self.n_rows + 1:
            raise ValueError("row_ptr must have n_rows+1 entries")
        if self.col_idx.shape[0] != self.n_edges:
            raise ValueError("col_idx length %d != row_ptr[-1] %d" % (self.col_idx.shape[0], self.n_edges))
        # The kernels size their LDS tiles from max_nodes / max_edges: both are derived from the batch itself, and
        # a caller-supplied value may only be LARGER (an understated value would overrun the tile).
        bounds = self.graph_bounds()
        if bounds[0] != 0 or bounds[-1] != self.n_rows or np.any(np.diff(bounds) <= 0):
            raise ValueError("graph_off must start at 0, end at n_rows and be strictly increasing")
        real_nodes = int(np.diff(bounds).max())
        real_edges = int(np.diff(self.row_ptr[bounds]).max())
        if max_nodes is not None and int(max_nodes) < real_nodes:
            raise ValueError("max_nodes %d is smaller than the largest graph (%d nodes)" % (max_nodes, real_nodes))
        if max_edges is not None and int(max_edges) < real_edges:
            raise ValueError("max_edges %d is smaller than the largest graph's edge count %d" % (max_edges, real_edges))
        self.max_nodes = int(max_nodes) if max_nodes is not None else real_nodes
        self.max_edges = int(max_edges) if max_edges is not None else real_edges

    @classmethod
    def trusted(cls, n_graphs, n_nodes, xe, row_ptr, col_idx, max_edges, nbr=None):
        """Fixed-size batch whose arrays were just produced by the library's own packer (v2x_pack_feed): contiguous,
        typed and consistent by construction, so the numpy-side checks of __init__ are skipped (the C ABI re-checks
        every host batch anyway)."""
        self = cls.__new__(cls)
        self.n_graphs, self.n_nodes = int(n_graphs), int(n_nodes)
        self.xe, self.row_ptr, self.col_idx, self.nbr, self.graph_off = xe, row_ptr, col_idx, nbr, None
        self.n_rows = xe.shape[0]
        self.n_edges = int(col_idx.shape[0])
        self.max_nodes, self.max_edges = int(n_nodes), int(max_edges)
        return self

    def graph_bounds(self):
        """First node row of every graph, plus n_rows: [B+1] int64."""
        if self.graph_off is not None:
            if self.graph_off.shape[0] != self.n_graphs + 1:
                raise ValueError("graph_off must have n_graphs+1 entries")
            return self.graph_off.astype(np.int64)
        if self.n_rows != self.n_graphs * self.n_nodes:
            raise ValueError("n_rows (%d) != n_graphs*n_nodes (%d*%d)" % (self.n_rows, self.n_graphs, self.n_nodes))
        return np.arange(self.n_graphs + 1, dtype=np.int64) * self.n_nodes

    def validate(self):
        """Full check of the CSR contract (include/v2xgnn.h): row_ptr monotone, sources inside their graph and
        strictly ascending within a row (=> no duplicate edges).  O(E) numpy; from_dense output satisfies it by
        construction."""
        rp = self.row_ptr.astype(np.int64)
        if rp[0] != 0 or np.any(np.diff(rp) < 0):
            raise ValueError("row_ptr must start at 0 and be non-decreasing")
        if self.n_edges == 0:
            return self
        bounds = self.graph_bounds()
        n_of_row = np.repeat(np.diff(bounds), np.diff(bounds))
        deg = np.diff(rp)
        n_of_edge = np.repeat(n_of_row, deg)
        col = self.col_idx.astype(np.int64)
        if np.any(col < 0) or np.any(col >= n_of_edge):
            raise ValueError("col_idx holds a source id outside its graph")
        first = np.zeros(self.n_edges, bool)
        first[rp[:-1][deg > 0]] = True
        if np.any((np.diff(col) <= 0) & ~first[1:]):
            raise ValueError("sources of a row must be strictly ascending (duplicate edges are not allowed)")
        return self

    @classmethod
    def from_dense(cls, x, e, adj, nbr=None):
        """x[B,N,Dn], e[B,N,De], adj[B,N,N], nbr[B,N,F] or None."""
        x = np.asarray(x)
        B, N = x.shape[0], x.shape[1]
        row_ptr, col_idx, max_edges = adj_to_csr(adj)
        xe = pack_xe(x.reshape(B * N, -1), np.asarray(e).reshape(B * N, -1))
        nb = None if nbr is None else np.asarray(nbr).reshape(B * N, -1)
        return cls(B, N, xe, row_ptr, col_idx, max_edges, nbr=nb)

    def shard_bounds(self, world, balance="edges"):
        """Graph-index boundaries [world+1] of `world` contiguous shards of WHOLE graphs (AggLayer only contracts inside
        a sample, BS_brain.py:73).  Fixed-size graphs: equal counts.  Variable-size graphs: balanced
        by the cumulative per-graph cost -- "edges" (gather work + node rows; default), "nodes" or "count" -- with at
        least one graph per shard."""
        B = self.n_graphs
        if world < 1 or world > B:
            raise ValueError("cannot cut %d graphs into %d shards" % (B, world))
        if self.graph_off is None or balance == "count":
            if self.graph_off is None and B % world:
                raise ValueError("batch %d not divisible by world size %d" % (B, world))
            return (np.arange(world + 1, dtype=np.int64) * B) // world
        bounds = self.graph_bounds()
        nodes = np.diff(bounds)
        cost = nodes.astype(np.float64)
        if balance == "edges":
            cost = cost + np.diff(self.row_ptr[bounds].astype(np.int64))
        elif balance != "nodes":
            raise ValueError("balance must be 'edges', 'nodes' or 'count'")
        cum = np.concatenate([[0.0], np.cumsum(cost)])
        targets = cum[-1] * np.arange(1, world) / world
        hi = np.clip(np.searchsorted(cum, targets, side="left"), 1, B)          # cum[hi] >= target
        cuts = np.where(targets - cum[hi - 1] < cum[hi] - targets, hi - 1, hi)  # the nearer graph boundary
        out = np.concatenate([[0], cuts, [B]]).astype(np.int64)
        for k in range(1, world):
            out[k] = min(max(out[k], out[k - 1] + 1), B - (world - k))
        return out

    def slice_graphs(self, g0, g1):
        """Sub-batch of the whole graphs [g0, g1) and the range of node rows it covers."""
        bounds = self.graph_bounds()
        r0, r1 = int(bounds[g0]), int(bounds[g1])
        e0, e1 = int(self.row_ptr[r0]), int(self.row_ptr[r1])
        goff = None if self.graph_off is None else (self.graph_off[g0:g1 + 1] - r0).astype(np.int32)
        sub = PackedBatch(g1 - g0, self.n_nodes, self.xe[r0:r1], self.row_ptr[r0:r1 + 1] - e0, self.col_idx[e0:e1],
                          nbr=None if self.nbr is None else self.nbr[r0:r1], graph_off=goff)
        return sub, (r0, r1)

    def shard(self, rank, world, balance="edges", with_rows=False):
        """Contiguous shard `rank` of `world` for data parallelism (see shard_bounds).  with_rows: also return the
        (first, last+1) node rows of the shard, for slicing the targets."""
        b = self.shard_bounds(world, balance)
        sub, rows = self.slice_graphs(int(b[rank]), int(b[rank + 1]))
        return (sub, rows) if with_rows else sub


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


class AdjacencyCache(object):
    """Which `Adjacency_Matrix` array OBJECTS have passed the full Kronecker-structure check.  One replay of the
    reference hands the same array object to predict and then to fit (BS_brain.py:603 -> :652, :716); re-reading its
    16.8 MB (4 links x 16 features x 512 samples) to re-prove what the first call proved is the largest single cost of
    the dict boundary.  An entry is a weak reference to the array plus its address / shape / dtype and a copy of the
    strided sample A[:, ::F, ::F] -- the only entries the engine ever uses; a later call with the same live object and an
    identical sample skips the scan of the other entries (they cannot have changed the result, only the verdict of the
    check).  Any difference re-validates."""

    def __init__(self, capacity=4):
        self.capacity = capacity
        self._entries = {}

    def hit(self, A, F):
        ent = self._entries.get(id(A))
        if ent is None:
            return False
        ref, addr, shape, dtype, sample = ent
        if ref() is not A or addr != A.ctypes.data or shape != A.shape or dtype != A.dtype:
            del self._entries[id(A)]
            return False
        return bool(np.array_equal(A[:, ::F, ::F], sample))

    def remember(self, A, F):
        import weakref
        try:
            ref = weakref.ref(A)
        except TypeError:
            return
        if len(self._entries) >= self.capacity:
            self._entries.pop(next(iter(self._entries)))
        self._entries[id(A)] = (ref, A.ctypes.data, A.shape, A.dtype, A[:, ::F, ::F].copy())


_F32, _F64 = np.dtype(np.float32), np.dtype(np.float64)


def _native_array(a):
    """C-contiguous float32 / float64 view of an input array (anything else is converted to float64 like np.asarray
    followed by Keras' own standardisation would)."""
    a = np.asarray(a)
    if (a.dtype != _F64 and a.dtype != _F32) or not a.flags.c_contiguous:
        a = np.ascontiguousarray(a, np.float64)
    return a


def feed_to_packed(spec: GnnSpec, feed, validate_adjacency=True, cache=None):
    """Reference dict payload -> PackedBatch in one pass of compiled host code (v2x_pack_feed, csrc/host_pack.hpp).
    Same contract and error behaviour as feed_to_arrays + PackedBatch.from_dense (the numpy definition the tests compare
    it against); `cache`: an AdjacencyCache shared by the models of one BS."""
    import ctypes as C
    from . import lib as _lib
    lib = _lib.load_library()
    N, F, Dn, De = spec.n_nodes, spec.feat_dim, spec.node_in, spec.edge_in
    arrs = [None] * (3 * N)
    B = None
    for k in range(N):
        for j, (kind, width) in enumerate((('Node', Dn), ('Edge', De), ('Neighbor', F))):
            name = 'D%d_%s_Input' % (k + 1, kind)
            if name not in feed:
                raise ValueError('No data provided for "%s". Need data for each key' % name)
            arr = _native_array(feed[name])
            if arr.ndim != 2 or arr.shape[1] != width:
                raise ValueError('Error when checking input: expected %s to have shape (%d,) but got array '
                                 'with shape %r' % (name, width, arr.shape[1:]))
            if B is None:
                B = arr.shape[0]
            elif arr.shape[0] != B:
                raise ValueError('All input arrays should have the same number of samples')
            arrs[j * N + k] = arr
    if 'Adjacency_Matrix' not in feed:
        raise ValueError('No data provided for "Adjacency_Matrix". Need data for each key')
    A0 = feed['Adjacency_Matrix']
    A = _native_array(A0)
    if A.shape != (B, N * F, N * F):
        raise ValueError('Error when checking input: expected Adjacency_Matrix to have shape (%d, %d) but got '
                         'array with shape %r' % (N * F, N * F, A.shape[1:]))
    if B == 0:
        raise ValueError('empty batch')
    check = bool(validate_adjacency)
    cached = check and cache is not None and A is A0 and cache.hit(A, F)
    ptrs = (C.c_void_p * (3 * N))(*[a.ctypes.data for a in arrs])
    flags = (C.c_uint8 * (3 * N + 1))(*([a.dtype == _F64 for a in arrs] + [A.dtype == _F64]))
    base = C.cast(ptrs, C.POINTER(C.c_void_p))
    step = C.sizeof(C.c_void_p)
    at = lambda i: C.cast(C.addressof(ptrs) + i * step, C.POINTER(C.c_void_p))
    f = _lib.Feed(B, N, F, Dn, De, base, at(N), at(2 * N), flags, A.ctypes.data)
    R = B * N
    xe = np.empty((R, XE_WIDTH), np.float32)
    row_ptr = np.empty(R + 1, np.int32)
    col_idx = np.empty(R * N, np.int32)
    nbr = np.empty((R, F), np.float32)
    info = np.zeros(3, np.int32)
    rc = lib.v2x_pack_feed(C.byref(f), int(check and not cached), xe.ctypes.data, row_ptr.ctypes.data, col_idx.ctypes.data,
                           nbr.ctypes.data, info.ctypes.data)
    _lib.check(lib, rc, None)
    if check and not cached and cache is not None and A is A0:
        cache.remember(A, F)
    return PackedBatch.trusted(B, N, xe, row_ptr, col_idx[:int(info[0])], int(info[1]), nbr if info[2] else None)


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
