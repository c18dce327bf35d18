"""CPU: host-side logic of the product package (packing, weight layout, error behaviour) and the
C-ABI shared library: it loads, exports every symbol include/v2xgnn.h declares, and refuses to
run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import v2xgnn
from v2xgnn import GnnSpec, PackedBatch
from v2xgnn import lib as vlib
from oracle import compact as oc, literal as ol
from oracle.spec import GnnSpec as OSpec
from util import golden_forward_cases, golden_feed, random_inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'v2xgnn.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(v2x_[a-z_0-9]+)\s*\(', hdr))
    assert len(declared) >= 20
    assert os.path.exists(vlib.library_path()), "build the HIP extension first (__graft_entry__.build())"
    lib = C.CDLL(vlib.library_path())
    for name in sorted(declared):
        assert hasattr(lib, name), "libv2xgnn.so does not export %s" % name
    bound = {n for n, _, _ in vlib.SYMBOLS}
    assert declared == bound, (declared - bound, bound - declared)
    assert b'gfx950' in vlib.load_library().v2x_version()


def test_simulator_library_exports_every_declared_symbol():
    """include/v2xsim.h (the batched simulator's helper library, callers' side) against libv2xsim.so and its ctypes binding"""
    from v2xgnn.rl import native_sim
    hdr = open(os.path.join(ROOT, 'include', 'v2xsim.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(v2xsim_[a-z_0-9]+)\s*\(', hdr))
    assert len(declared) >= 15
    path = os.path.join(os.path.dirname(vlib.library_path()), 'libv2xsim.so')
    assert os.path.exists(path), "build first (__graft_entry__.build())"
    lib = C.CDLL(path)
    for name in sorted(declared):
        assert hasattr(lib, name), "libv2xsim.so does not export %s" % name
    assert lib.v2xsim_abi() == 3 and native_sim.available()
    a = native_sim.AdvanceArgs()
    assert [f[0] for f in a._fields_][:4] == ["E", "n", "rb", "n_lanes"] and C.sizeof(a) == 4 * 4 + 3 * 8 + 6 * 8 + 5 * 8 + 24 * 8
    # v2xsim_rollout_args: 6 int32, 3 + 7 + 4 doubles, 6 + 22 + 14 pointers, step_no0, eps_last -- field for field the header's order
    r = native_sim.RolloutArgs()
    assert C.sizeof(r) == 6 * 4 + (3 + 7 + 4) * 8 + (6 + 22 + 14) * 8 + 8 + 8
    names = re.findall(r'\b([A-Za-z_0-9]+)\s*[;,]', re.search(r'typedef struct \{([^}]*)\} v2xsim_rollout_args;', hdr, re.S).group(1).replace('*', ' '))
    assert names == [f[0] for f in r._fields_], (names, [f[0] for f in r._fields_])


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(vlib.V2XError, match="no CPU fallback"):
        v2xgnn.GnnEngine(GnnSpec())
    with pytest.raises(vlib.V2XError):
        v2xgnn.BS(4, 3, 1, 16, 1, 4)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'globecom2020-resourceallocationgnn_amd')
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.hpp', '.h', '.cpp')):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), fn
                assert 'oracle/' not in src.replace('SURVEY', ''), fn
    # repo-wide: only tests/, the smoke entry and bench.py's cpu_baseline leg may touch the oracle
    allowed = {os.path.join(ROOT, 'bench.py'), os.path.join(ROOT, '__graft_entry__.py')}
    for dirpath, dirs, files in os.walk(ROOT):
        dirs[:] = [d for d in dirs if d not in ('.git', 'gpurun_out', '__pycache__', 'tests', 'oracle')]
        for fn in files:
            path = os.path.join(dirpath, fn)
            if fn.endswith('.py') and path not in allowed:
                assert not re.search(r'^\s*(from|import)\s+oracle\b', open(path).read(), flags=re.M), path


def test_spec_matches_reference_size_bookkeeping():
    s = GnnSpec()
    assert (s.node_in, s.edge_in, s.n_params) == (9, 4, 37824)
    assert GnnSpec(n_nodes=20, feat_dim=64).n_params == 767040
    assert GnnSpec(n_nodes=20, feat_dim=64, share_weights=True).n_params == 38352
    assert len(v2xgnn.keras_list_shapes(s)) == 80


def test_weight_layout_roundtrip_and_dense0_permutation():
    spec = GnnSpec(n_nodes=3, feat_dim=16)
    rng = np.random.default_rng(0)
    ws = [rng.normal(size=s).astype(np.float32) for s in v2xgnn.keras_list_shapes(spec)]
    flat = v2xgnn.keras_list_to_flat(spec, ws)
    back = v2xgnn.flat_to_keras_list(spec, flat)
    assert all(np.array_equal(a, b) for a, b in zip(ws, back))
    # flat layout as documented in include/v2xgnn.h: stage 0, slot 0 = vstack(W1, W2, W3), bias
    n0 = (9 + 4 + 16) * 16
    assert np.array_equal(flat[:n0].reshape(29, 16), np.vstack(ws[0:3]))
    assert np.array_equal(flat[n0:n0 + 16], ws[3])
    # Dense-0 rows are stored [h | x | agg]
    S = spec.n_slots
    off = sum((spec.stage_in_a(s) + 4 + 16) * 16 + 16 for s in range(3)) * S
    W = ws[3 * S * 4]                      # Keras order [x(9) | h(16) | agg(16)]
    got = flat[off:off + 41 * 80].reshape(41, 80)
    assert np.array_equal(got[:16], W[9:25]) and np.array_equal(got[16:25], W[:9]) and np.array_equal(got[25:], W[25:])
    with pytest.raises(ValueError):
        v2xgnn.keras_list_to_flat(spec, ws[:-1])
    bad = list(ws)
    bad[0] = bad[0][:, :5]
    with pytest.raises(ValueError):
        v2xgnn.keras_list_to_flat(spec, bad)


def test_pack_and_csr_match_reference_adjacency_semantics():
    rng = np.random.default_rng(1)
    B, N, F = 5, 4, 16
    x, e, adj = random_inputs(rng, B, N)
    pb = PackedBatch.from_dense(x, e, adj)
    assert pb.xe.shape == (B * N, 16) and pb.xe.dtype == np.float32
    assert np.array_equal(pb.xe[:, :9], x.reshape(-1, 9)) and np.array_equal(pb.xe[:, 9:13], e.reshape(-1, 4))
    assert not pb.xe[:, 13:].any()
    # CSR by destination == AggLayer with kron(Adj, I_F)  (BS_brain.py:72-76)
    h = rng.normal(size=(B, N, F))
    ref = ol.agg_layer([h[:, k] for k in range(N)], np.kron(adj, np.eye(F)), F)
    M = oc.csr_to_matrix((np.arange(B + 1) * N).astype(np.int32), pb.row_ptr, pb.col_idx, np.float64)
    got = (M @ h.reshape(B * N, F)).reshape(B, N, F)
    for k in range(N):
        assert np.allclose(got[:, k], ref[k], atol=1e-13)
    assert pb.max_edges == 8 and pb.n_edges == B * 8
    # oracle's own CSR builder agrees with the product's
    g = oc.adj_to_csr(adj)
    assert np.array_equal(g[1], pb.row_ptr) and np.array_equal(g[2], pb.col_idx)
    with pytest.raises(ValueError):
        v2xgnn.adj_to_csr(adj * 2.0)


def test_feed_to_arrays_on_reference_payloads_and_errors():
    f, cases = golden_forward_cases()
    spec = GnnSpec()
    for case in cases:
        feed = golden_feed(f, case)
        x, e, nbr, adj = v2xgnn.feed_to_arrays(spec, feed)
        assert x.shape[1:] == (4, 9) and e.shape[1:] == (4, 4) and adj.shape[1:] == (4, 4)
        assert (nbr is None) == (case != 'synthetic_b6')
        # inverse: compact arrays rebuild the reference's dict payload exactly
        back = ol.feed_from_compact(OSpec(), x, e, adj, nbr)
        for k, v in feed.items():
            assert np.array_equal(back[k], v), k
    feed = golden_feed(f, 'env_b1')
    for mutate in ('drop', 'shape', 'kron', 'adjshape'):
        bad = dict(feed)
        if mutate == 'drop':
            del bad['D3_Neighbor_Input']
        elif mutate == 'shape':
            bad['D1_Edge_Input'] = np.zeros((1, 5))
        elif mutate == 'kron':
            A = bad['Adjacency_Matrix'].copy()
            A[0, 1, 0] = 1
            bad['Adjacency_Matrix'] = A
        else:
            bad['Adjacency_Matrix'] = np.zeros((1, 32, 32))
        with pytest.raises(ValueError):
            v2xgnn.feed_to_arrays(spec, bad)


def test_batch_shard_partitions_graphs():
    rng = np.random.default_rng(2)
    B, N = 8, 5
    x, e, adj = random_inputs(rng, B, N, ref_topology=False)
    pb = PackedBatch.from_dense(x, e, adj)
    parts = [pb.shard(r, 4) for r in range(4)]
    assert sum(p.n_graphs for p in parts) == B and sum(p.n_edges for p in parts) == pb.n_edges
    assert np.array_equal(np.concatenate([p.xe for p in parts]), pb.xe)
    assert np.array_equal(np.concatenate([p.col_idx for p in parts]), pb.col_idx)
    for r, p in enumerate(parts):
        assert p.row_ptr[0] == 0 and p.row_ptr[-1] == p.n_edges
        sub = PackedBatch.from_dense(x[2 * r:2 * r + 2], e[2 * r:2 * r + 2], adj[2 * r:2 * r + 2])
        assert np.array_equal(sub.row_ptr, p.row_ptr) and sub.max_edges == p.max_edges
    with pytest.raises(ValueError):
        pb.shard(0, 3)


def test_keras_hdf5_weight_files_roundtrip(tmp_path):
    """h5weights.py writes / reads the layout of Keras `save_weights` through libhdf5 itself: layer_names /
    weight_names attributes, nested '<layer>/<layer>/<weight>:0' float32 datasets; order-based matching."""
    from v2xgnn import h5weights
    if not h5weights.available():
        pytest.skip("libhdf5 not loadable on this host")
    rng = np.random.default_rng(0)
    layers = [('D1_GNN', [('D1_GNN/W1:0', rng.normal(size=(9, 16))), ('D1_GNN/W2:0', rng.normal(size=(4, 16))),
                          ('D1_GNN/W3:0', rng.normal(size=(16, 16))), ('D1_GNN/bias:0', rng.normal(size=(16,)))]),
              ('dense_1', [('dense_1/kernel:0', rng.normal(size=(41, 80))), ('dense_1/bias:0', np.zeros(80))]),
              ('D1_Decide_Output', [('D1_Decide_Output/kernel:0', rng.normal(size=(20, 4))),
                                    ('D1_Decide_Output/bias:0', rng.normal(size=(4,)))])]
    path = str(tmp_path / 'w.h5')
    h5weights.save_keras_weights(path, layers)
    assert h5weights.is_hdf5(path)
    back = h5weights.load_keras_weights(path)
    assert [ln for ln, _ in back] == [ln for ln, _ in layers]
    for (_, ws0), (_, ws1) in zip(layers, back):
        assert [n for n, _ in ws0] == [n for n, _ in ws1]
        for (_, a), (_, b) in zip(ws0, ws1):
            assert b.dtype == np.float32 and np.array_equal(np.asarray(a, np.float32), b)
    # structural check against the on-disk names: the nested group h5py creates for 'layer/W1:0'
    import ctypes as C
    h = h5weights._load()
    f = h.H5Fopen(path.encode(), 0, 0)
    d = h.H5Dopen2(f, b"/D1_GNN/D1_GNN/W1:0", 0)
    assert f >= 0 and d >= 0
    h.H5Dclose(d)
    h.H5Fclose(f)
    assert not h5weights.is_hdf5(__file__)


def test_keras_layer_table_matches_weight_list():
    """Layer / weight names of the HDF5 export follow the reference model (BS_brain.py:121-200) and cover
    get_weights() exactly, in order."""
    from v2xgnn.bs_brain import GnnQModel
    from v2xgnn.packing import keras_list_shapes
    for spec in (GnnSpec(), GnnSpec(n_nodes=20, feat_dim=64), GnnSpec(n_nodes=3, feat_dim=32, n_mp_layers=3, share_weights=True)):
        m = GnnQModel.__new__(GnnQModel)
        m.spec, m.model_index = spec, 0
        table = m.keras_layer_table()
        assert sum(len(w) for _, w in table) == len(keras_list_shapes(spec))
        assert len({ln for ln, _ in table}) == len(table)
    m.spec = GnnSpec()
    t = m.keras_layer_table()
    assert t[0] == ('D1_GNN', ['D1_GNN/W1:0', 'D1_GNN/W2:0', 'D1_GNN/W3:0', 'D1_GNN/bias:0'])
    assert t[4][0] == 'gnn_layer_1' and t[11][0] == 'gnn_layer_8'
    # Keras numbers auto-named layers in CREATION order: node 1's Dense 80/40/20 are dense_1/2/3, node 2's dense_4/5/6
    # (BS_brain.py:176-186); get_weights() lists them depth-sorted (all 80-wide layers first)
    assert [t[i][0] for i in (12, 13, 14, 15)] == ['dense_1', 'dense_4', 'dense_7', 'dense_10']     # the four 80-wide layers
    assert [t[i][0] for i in (16, 20)] == ['dense_2', 'dense_3'] and t[23][0] == 'dense_12'
    assert t[-1] == ('D4_Decide_Output', ['D4_Decide_Output/kernel:0', 'D4_Decide_Output/bias:0'])
    m.model_index = 1                       # the target network is the second model of the session (BS_brain.py:106)
    t = m.keras_layer_table()
    assert t[4][0] == 'gnn_layer_9' and t[12][0] == 'dense_13' and t[0][0] == 'D1_GNN'


def test_fit_minibatches_epochs_and_shuffle():
    """Model.fit semantics above the engine (bs_brain.GnnQModel.fit): minibatches of `batch_size` in order or in the
    order of ONE np.random.shuffle per epoch (Keras shuffles with the global numpy RNG), `epochs` passes, History
    entries = sample-weighted means of the minibatch losses.  Engine injected: the float64 oracle."""
    from oracle_engine import OracleEngine
    from v2xgnn.bs_brain import GnnQModel
    from v2xgnn.packing import PackedBatch as PB
    spec = GnnSpec(n_nodes=4, feat_dim=16)
    rng = np.random.default_rng(8)
    B = 40
    x, e, adj = random_inputs(rng, B, 4)
    feed = {}
    for k in range(4):
        feed['D%d_Node_Input' % (k + 1)] = x[:, k, :].astype(np.float64)
        feed['D%d_Edge_Input' % (k + 1)] = e[:, k, :].astype(np.float64)
        feed['D%d_Neighbor_Input' % (k + 1)] = np.zeros((B, 16))
    feed['Adjacency_Matrix'] = np.kron(adj, np.eye(16))
    yt = rng.normal(2.5, 1.0, size=(B, 4, 4))
    y = {'D%d_Decide_Output' % (k + 1): yt[:, k, :] for k in range(4)}
    for shuffle in (False, True):
        model = GnnQModel(spec, seed=3, engine=OracleEngine(spec))
        ref = OracleEngine(spec)
        ref.set_weights(model.get_weights())
        np.random.seed(21)
        hist = model.fit(feed, y, batch_size=16, epochs=2, shuffle=shuffle)
        np.random.seed(21)
        want = []
        for ep in range(2):
            idx = np.arange(B)
            if shuffle:
                np.random.shuffle(idx)
            tot = np.zeros(4)
            for s in range(0, B, 16):
                sel = idx[s:s + 16]
                loss = ref.train_step(PB.from_dense(x[sel], e[sel], adj[sel]), yt[sel].reshape(-1, 4))
                tot += np.asarray(loss) * len(sel)
            want.append(tot / B)
        assert hist.epoch == [0, 1] and len(hist.history['loss']) == 2
        for ep in range(2):
            got = [hist.history['D%d_Decide_Output_loss' % (k + 1)][ep] for k in range(4)]
            assert np.allclose(got, want[ep], rtol=1e-6)                 # (the feed is rounded to fp32 when packed)
            assert np.isclose(hist.history['loss'][ep], np.sum(want[ep]), rtol=1e-6)
        for a, b in zip(model.get_weights(), ref.get_weights()):
            assert np.allclose(a, b, rtol=1e-5, atol=1e-7)


def test_ragged_shards_are_whole_graphs_balanced_by_cost():
    """PackedBatch.shard on variable-size batches: contiguous, disjoint, complete, every shard a valid
    batch of its own, cost (edges + nodes) within one largest graph of the ideal share."""
    import bench
    rng = np.random.default_rng(5)
    sizes, offs, row_ptr, col_idx, x, e, y = bench.synth_ragged(rng, 257, 8, 128)
    pb = PackedBatch(len(sizes), 0, v2xgnn.pack_xe(x, e), row_ptr, col_idx, graph_off=offs).validate()
    assert pb.max_nodes == sizes.max() and pb.max_edges == (sizes * (sizes - 2)).max()
    cost = sizes + sizes * (sizes - 2)
    for world in (2, 3, 8):
        b = pb.shard_bounds(world)
        assert b[0] == 0 and b[-1] == pb.n_graphs and np.all(np.diff(b) >= 1)
        shards = [pb.shard(r, world, with_rows=True) for r in range(world)]
        assert sum(s.n_graphs for s, _ in shards) == pb.n_graphs
        assert [r for _, r in shards][0][0] == 0 and all(shards[i][1][1] == shards[i + 1][1][0] for i in range(world - 1))
        for r, (s, (r0, r1)) in enumerate(shards):
            s.validate()
            assert np.array_equal(s.xe, pb.xe[r0:r1]) and s.graph_off[0] == 0 and s.graph_off[-1] == r1 - r0
            share = cost[b[r]:b[r + 1]].sum()
            assert abs(share - cost.sum() / world) <= cost.max(), (world, r)
    # equal-count sharding is what the bench used to do: visibly worse balance on the same batch
    by_count = pb.shard_bounds(8, balance="count")
    worst = lambda bb: max(cost[bb[i]:bb[i + 1]].sum() for i in range(8))
    assert worst(pb.shard_bounds(8)) <= worst(by_count)


def test_packed_batch_rejects_understated_tile_sizes_and_broken_csr():
    rng = np.random.default_rng(6)
    x = rng.normal(size=(3, 5, 9)); e = rng.normal(size=(3, 5, 4))
    adj = (rng.uniform(size=(3, 5, 5)) < 0.6).astype(float)
    pb = PackedBatch.from_dense(x, e, adj).validate()
    with pytest.raises(ValueError, match="max_edges"):
        PackedBatch(3, 5, pb.xe, pb.row_ptr, pb.col_idx, max_edges=pb.max_edges - 1)
    with pytest.raises(ValueError, match="max_nodes"):
        PackedBatch(3, 5, pb.xe, pb.row_ptr, pb.col_idx, max_nodes=4)
    bad = pb.col_idx.copy(); bad[0] = 5
    with pytest.raises(ValueError, match="outside"):
        PackedBatch(3, 5, pb.xe, pb.row_ptr, bad).validate()
    deg = np.diff(pb.row_ptr)
    r = int(np.argmax(deg >= 2))
    bad = pb.col_idx.copy(); bad[pb.row_ptr[r] + 1] = bad[pb.row_ptr[r]]
    with pytest.raises(ValueError, match="ascending"):
        PackedBatch(3, 5, pb.xe, pb.row_ptr, bad).validate()


# ------------------------------------------------------------------------------ drop-in boundary: compiled packer
def _ref_payload(rng, B, N, F, p_edge=0.6, nbr=False, dtype=np.float64):
    adj = (rng.random((B, N, N)) < p_edge).astype(np.float64)
    feed = {}
    for k in range(N):
        feed['D%d_Node_Input' % (k + 1)] = rng.normal(size=(B, 9)).astype(dtype)
        feed['D%d_Edge_Input' % (k + 1)] = rng.normal(size=(B, 4)).astype(dtype)
        feed['D%d_Neighbor_Input' % (k + 1)] = (rng.normal(size=(B, F)) if nbr else np.zeros((B, F))).astype(dtype)
    feed['Adjacency_Matrix'] = np.kron(adj, np.eye(F)).astype(dtype)
    return feed, adj


@pytest.mark.parametrize("B,N,F,dtype,nbr", [(1, 4, 16, np.float64, False), (37, 4, 16, np.float64, False),
                                              (5, 4, 16, np.float32, True), (3, 20, 64, np.float64, False),
                                              (600, 4, 16, np.float64, True)])
def test_native_packer_equals_the_numpy_definition(B, N, F, dtype, nbr):
    """v2x_pack_feed (csrc/host_pack.hpp: the reference's dict payload -> packed batch in one pass of compiled host code) against
    packing.feed_to_arrays + PackedBatch.from_dense, the numpy definition: identical xe rows, CSR, max_edges, neighbour input."""
    from v2xgnn.packing import feed_to_packed, feed_to_arrays
    spec = GnnSpec(n_nodes=N, feat_dim=F)
    feed, adj = _ref_payload(np.random.default_rng(B + N), B, N, F, nbr=nbr, dtype=dtype)
    x, e, nb, a = feed_to_arrays(spec, feed, True)
    ref = PackedBatch.from_dense(x, e, a, nb)
    for validate in (True, False):
        got = feed_to_packed(spec, feed, validate)
        for f in ('xe', 'row_ptr', 'col_idx'):
            assert np.array_equal(getattr(ref, f), getattr(got, f)), f
        assert (got.n_graphs, got.n_nodes, got.n_rows, got.n_edges, got.max_nodes, got.max_edges) == \
               (ref.n_graphs, ref.n_nodes, ref.n_rows, ref.n_edges, ref.max_nodes, ref.max_edges)
        assert (got.nbr is None) == (ref.nbr is None) and (ref.nbr is None or np.array_equal(got.nbr, ref.nbr))
        got.validate()
    # non-contiguous / integer inputs are standardised like np.asarray would
    feed2 = dict(feed)
    feed2['D1_Node_Input'] = np.asfortranarray(feed['D1_Node_Input'])
    feed2['Adjacency_Matrix'] = feed['Adjacency_Matrix'].astype(np.int64)
    assert np.array_equal(feed_to_packed(spec, feed2, True).col_idx, ref.col_idx)


def test_native_packer_rejects_what_the_numpy_definition_rejects():
    from v2xgnn.packing import feed_to_packed
    spec = GnnSpec()
    B, N, F = 70, 4, 16
    feed, adj = _ref_payload(np.random.default_rng(5), B, N, F)
    NF = N * F

    def bad_adj(mutate):
        A = feed['Adjacency_Matrix'].copy()
        mutate(A)
        return dict(feed, Adjacency_Matrix=A)
    cases = {
        "off-diagonal entry inside a block": lambda A: A.__setitem__((69, 0, 5), 1.0),
        "off-diagonal entry, last row": lambda A: A.__setitem__((33, NF - 1, 0), 1e-300),
        "diagonal of a block not constant": lambda A: A.__setitem__((12, 1 * F + 3, 2 * F + 3), 1.0 - A[12, 1 * F + 3, 2 * F + 3]),
        "NaN off the diagonals": lambda A: A.__setitem__((0, 2, 9), np.nan),
        "weighted edge": lambda A: A.__setitem__((4, slice(None), slice(None)), 2.0 * A[4]),
    }
    for name, mut in cases.items():
        with pytest.raises(ValueError):
            feed_to_packed(spec, bad_adj(mut), True)
        with pytest.raises(ValueError):                      # the numpy definition agrees
            x, e, nb, a = v2xgnn.feed_to_arrays(spec, bad_adj(mut), True)
            PackedBatch.from_dense(x, e, a, nb)
    # a negative zero off the diagonals IS kron(Adj, I_F) (numpy: -0.0 == 0)
    ok = bad_adj(lambda A: A.__setitem__((3, 0, 5), -0.0))
    assert np.array_equal(feed_to_packed(spec, ok, True).col_idx, feed_to_packed(spec, feed, True).col_idx)
    # ... and so is a negative zero ON a block's diagonal where the adjacency entry is 0 (the compiled packer used
    # to compare the diagonal by bit pattern and rejected it); a -1 where the entry is 1 is a different VALUE: rejected by both
    p0, q0 = [int(v[0]) for v in np.nonzero(adj[3] == 0)]
    p1, q1 = [int(v[0]) for v in np.nonzero(adj[3] == 1)]
    ok = bad_adj(lambda A: A.__setitem__((3, p0 * F + 5, q0 * F + 5), -0.0))
    assert np.array_equal(feed_to_packed(spec, ok, True).col_idx, feed_to_packed(spec, feed, True).col_idx)
    x_, e_, nb_, a_ = v2xgnn.feed_to_arrays(spec, ok, True)
    assert np.array_equal(PackedBatch.from_dense(x_, e_, a_, nb_).col_idx, feed_to_packed(spec, feed, True).col_idx)
    neg = bad_adj(lambda A: A.__setitem__((3, p1 * F + 5, q1 * F + 5), -1.0))
    with pytest.raises(ValueError):
        feed_to_packed(spec, neg, True)
    with pytest.raises(ValueError):
        x_, e_, nb_, a_ = v2xgnn.feed_to_arrays(spec, neg, True)
        PackedBatch.from_dense(x_, e_, a_, nb_)
    for mutate in ('drop', 'shape', 'adjshape', 'samples'):
        bad = dict(feed)
        if mutate == 'drop':
            del bad['D3_Neighbor_Input']
        elif mutate == 'shape':
            bad['D1_Edge_Input'] = np.zeros((B, 5))
        elif mutate == 'samples':
            bad['D2_Node_Input'] = np.zeros((B + 1, 9))
        else:
            bad['Adjacency_Matrix'] = np.zeros((B, 32, 32))
        with pytest.raises(ValueError):
            feed_to_packed(spec, bad, True)


def test_adjacency_cache_skips_only_what_it_has_seen():
    """One replay of the reference hands the same Adjacency_Matrix object to predict and to fit (BS_brain.py:603 -> :652, :716):
    the second call skips the Kronecker scan, but only for the same live object with an unchanged strided sample."""
    from v2xgnn.packing import feed_to_packed, AdjacencyCache
    spec = GnnSpec()
    feed, adj = _ref_payload(np.random.default_rng(6), 16, 4, 16)
    cache = AdjacencyCache()
    A = feed['Adjacency_Matrix']
    assert not cache.hit(A, 16)
    ref = feed_to_packed(spec, feed, True, cache)
    assert cache.hit(A, 16)
    assert np.array_equal(feed_to_packed(spec, dict(feed), True, cache).col_idx, ref.col_idx)
    # an equal COPY is another object: not a hit
    assert not cache.hit(A.copy(), 16)
    # the engine-visible entries changed in place: not a hit, and the new adjacency is what gets packed
    A[2, 0 * 16:1 * 16, 1 * 16:2 * 16] = (1.0 - A[2, 0, 16]) * np.eye(16)
    assert not cache.hit(A, 16)
    got = feed_to_packed(spec, feed, True, cache)
    x, e, nb, a = v2xgnn.feed_to_arrays(spec, feed, True)
    assert np.array_equal(got.col_idx, PackedBatch.from_dense(x, e, a, nb).col_idx)
    # a broken structure in a NEW object is still caught with the cache in place
    B2 = A.copy()
    B2[1, 0, 7] = 1.0
    with pytest.raises(ValueError):
        feed_to_packed(spec, dict(feed, Adjacency_Matrix=B2), True, cache)
    # the cache holds weak references only
    import gc
    n0 = len(cache._entries)
    del A, feed, got
    gc.collect()
    assert all(ent[0]() is None for ent in cache._entries.values()) or len(cache._entries) <= n0


def test_bench_gpus_2_without_a_launcher_starts_two_ranks():
    """`python bench.py --gpus 2` with no torch.distributed.run around it must start its ranks itself (it used to
    SystemExit).  Without a GPU each rank stops at the engine's no-CPU-fallback check -- AFTER the launch: the launcher's
    failure report names bench.py's ranks (it terminates the second rank as soon as the first has failed, so only one of them
    is sure to get its message out), and the exit code is the launcher's."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""
    env["CUDA_VISIBLE_DEVICES"] = ""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode != 0
    assert out.stderr.count("bench.py needs a GPU") >= 1, out.stderr[-2000:]
    assert "local_rank" in out.stderr or "ChildFailedError" in out.stderr, out.stderr[-2000:]     # torch.distributed.run's report
    assert "launch with torch.distributed.run" not in out.stderr
