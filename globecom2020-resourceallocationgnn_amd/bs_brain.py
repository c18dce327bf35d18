"""Drop-in mirror of the reference's `BS` Q-network wrapper (BS_brain.py:90-239) and of the part
of the Keras `Model` API its callers use (`fit`, `predict`, `get_weights`, `set_weights`,
`save_weights`, `load_weights`; call sites BS_brain.py:221,229,231,239,863,869,1254,1256).

Same constructor, attributes, dict-keyed inputs / outputs and History keys as the reference;
the arithmetic runs in the gfx950 kernels behind include/v2xgnn.h.  `Agent` (BS_brain.py:280-)
can use this class unchanged:  `self.brain = BS(num_d2d, 3, 1, num_feedback, num_neighbor, num_ch)`.
"""
import numpy as np

from .engine import GnnEngine
from .packing import PackedBatch, AdjacencyCache, feed_to_arrays, feed_to_packed, keras_list_shapes
from .spec import GnnSpec


class History(object):
    """What `Model.fit` returns; the caller reads history['D{k}_Decide_Output_loss'][0]
    (BS_brain.py:835-837)."""

    def __init__(self):
        self.history = {}
        self.epoch = []


def _glorot_list(spec, rng):
    """glorot_uniform kernels / zero biases like GNNLayer.build (BS_brain.py:26-41) and Dense."""
    out = []
    for shp in keras_list_shapes(spec):
        if len(shp) == 1:
            out.append(np.zeros(shp, np.float32))
        else:
            lim = np.sqrt(6.0 / (shp[0] + shp[1]))
            out.append(rng.uniform(-lim, lim, size=shp).astype(np.float32))
    return out


class GnnQModel(object):
    """Keras-`Model`-like object over one GnnEngine."""

    def __init__(self, spec: GnnSpec, device=0, seed=None, use_graph=False, validate_adjacency=True,
                 lr=1e-3, beta_1=0.5, beta_2=0.999, epsilon=1e-7, data_parallel=False, process_group=None, engine=None,
                 model_index=0, adjacency_cache=None):
        """data_parallel: every fit step shards the minibatch over the ranks of `process_group` (default group) and
        all-reduces the gradient (v2xgnn.dp); all ranks must call fit with the SAME full minibatch.
        engine: an object with GnnEngine's interface (the CPU tests inject one); default: the gfx950 engine,
        which raises without a GPU."""
        self.spec = spec
        self.model_index = int(model_index)        # how many models of this shape the session built before (layer names)
        self.engine = engine if engine is not None else GnnEngine(
            spec, device=device, use_graph=use_graph, lr=lr, beta_1=beta_1, beta_2=beta_2,
            epsilon=epsilon)                                 # Adam(lr=0.001, beta_1=0.5, beta_2=0.999) BS_brain.py:212
        self.trainer = None
        if data_parallel:
            from .dp import DataParallelTrainer
            import os
            self.trainer = DataParallelTrainer(self.engine, process_group=process_group,
                                               force=os.environ.get("V2X_FORCE_DP") == "1")
        self.validate_adjacency = validate_adjacency
        # which Adjacency_Matrix objects already passed the Kronecker check (shared by the two models of a BS: one
        # replay hands the same array to predict and to fit, BS_brain.py:603 -> :652, :716)
        self.adjacency_cache = adjacency_cache if adjacency_cache is not None else AdjacencyCache()
        N = spec.n_nodes
        self.input_names = []
        for k in range(1, N + 1):                            # order of Model(inputs=[...]) BS_brain.py:203-207
            self.input_names += ['D%d_Node_Input' % k, 'D%d_Edge_Input' % k, 'D%d_Neighbor_Input' % k]
        self.input_names.append('Adjacency_Matrix')
        self.output_names = ['D%d_Decide_Output' % k for k in range(1, N + 1)]       # :208
        self.engine.set_weights(_glorot_list(spec, np.random.default_rng(seed)))

    # ------------------------------------------------------------------ data plumbing
    def _named(self, x, names, what):
        if isinstance(x, dict):
            return x
        if isinstance(x, (list, tuple)):
            if len(x) != len(names):
                raise ValueError("Error when checking model %s: expected %d arrays but got %d" % (what, len(names), len(x)))
            return dict(zip(names, x))
        raise ValueError("Error when checking model %s: expected a dict or list of arrays" % what)

    def _pack(self, x):
        """numpy definition of the payload -> arrays step (minibatched fits; the tests' reference for _pack_batch)"""
        feed = self._named(x, self.input_names, "input")
        xs, es, nbr, adj = feed_to_arrays(self.spec, feed, self.validate_adjacency)
        return xs, es, nbr, adj

    def _pack_batch(self, x):
        """dict payload -> PackedBatch in one pass of compiled host code (v2x_pack_feed)"""
        feed = self._named(x, self.input_names, "input")
        return feed_to_packed(self.spec, feed, self.validate_adjacency, self.adjacency_cache)

    def _targets(self, y, B):
        yd = self._named(y, self.output_names, "target")
        cols = []
        for name in self.output_names:
            if name not in yd:
                raise ValueError('No data provided for "%s". Need data for each key' % name)
            a = np.asarray(yd[name])
            if a.shape != (B, self.spec.n_channels):
                raise ValueError('Error when checking target: expected %s to have shape (%d,) but got array with '
                                 'shape %r' % (name, self.spec.n_channels, a.shape[1:]))
            cols.append(a)
        return np.stack(cols, axis=1)        # [B, N, C]

    def _train_step(self, pb, y, presharded=False, n_global=None):
        """One optimizer step on the whole minibatch `pb` -> per-output losses (of the whole minibatch).
        presharded: `pb` already IS this rank's share of a minibatch of n_global graphs (sharded rollouts)."""
        if self.trainer is None:
            return self.engine.train_step(pb, y)
        if presharded:
            return self.trainer.train_step(pb, y, n_graphs_global=n_global)
        if pb.n_graphs % self.trainer.world:
            raise ValueError("data-parallel fit: batch of %d graphs is not divisible by %d ranks"
                             % (pb.n_graphs, self.trainer.world))
        sb, sy = self.trainer.shard(pb, y)
        return self.trainer.train_step(sb, sy, n_graphs_global=pb.n_graphs)

    # ------------------------------------------------------------------ Keras surface
    def predict(self, x, batch_size=None, verbose=0):
        """-> list of N fresh, writable float32 arrays [B, C] (the caller mutates them in place,
        BS_brain.py:684-692)."""
        pb = self._pack_batch(x)
        B, N, Cc = pb.n_graphs, self.spec.n_nodes, self.spec.n_channels
        q = self.engine.forward(pb).reshape(B, N, Cc)
        return [np.ascontiguousarray(q[:, k, :]) for k in range(N)]

    def fit(self, x, y, batch_size=None, epochs=1, verbose=0, shuffle=True):
        """Model.fit: `epochs` passes of minibatch Adam steps.  The reference always calls it with
        batch_size == len(x), i.e. exactly one step (BS_brain.py:218-223)."""
        N = self.spec.n_nodes
        batch_size = int(batch_size or 32)       # Keras default
        feed = self._named(x, self.input_names, "input")
        first = feed.get(self.input_names[0])
        one_batch = first is not None and np.ndim(first) == 2 and batch_size >= np.shape(first)[0]
        if one_batch:                            # the reference's call: the whole data set is one minibatch
            whole = self._pack_batch(feed)
            B = whole.n_graphs
        else:
            xs, es, nbr, adj = self._pack(feed)
            B = xs.shape[0]
        yt = self._targets(y, B)
        hist = History()
        keys = ['loss'] + [n + '_loss' for n in self.output_names]
        for k in keys:
            hist.history[k] = []
        for ep in range(int(epochs)):
            idx = np.arange(B)
            if shuffle and B > 1:
                np.random.shuffle(idx)           # Keras shuffles with the global numpy RNG
            tot = np.zeros(N, np.float64)
            for s in range(0, B, batch_size):
                sel = idx[s:s + batch_size]
                if one_batch:
                    # whole data set in one minibatch: sample order inside the batch only changes
                    # the fp32 summation order, so skip the gather
                    pb, by = whole, yt
                else:
                    bx, be, ba, by = xs[sel], es[sel], adj[sel], yt[sel]
                    bn = None if nbr is None else nbr[sel]
                    pb = PackedBatch.from_dense(bx, be, ba, bn)
                loss = self._train_step(pb, by.reshape(-1, self.spec.n_channels))
                tot += np.asarray(loss, np.float64) * len(sel)
            tot /= B
            hist.epoch.append(ep)
            hist.history['loss'].append(float(tot.sum()))              # sum of per-output losses (:214)
            for k, name in enumerate(self.output_names):
                hist.history[name + '_loss'].append(float(tot[k]))
        return hist

    # ------------------------------------------------------------------ compact surface
    # Same arithmetic as predict / fit without the dict-of-arrays and the dense kron(Adj, I_F)
    # adjacency (6.5 MB per graph at 20 links x 64 features); used by v2xgnn.rl.Agent.
    def predict_arrays(self, x, e, adj, nbr=None):
        """x [B, N, Dn], e [B, N, De], adj [B, N, N] (Adj[p, q] = 1 when p sends to q) -> Q [B, N, C]"""
        x = np.asarray(x)
        q = self.engine.forward(PackedBatch.from_dense(x, e, adj, nbr))
        return q.reshape(x.shape[0], self.spec.n_nodes, self.spec.n_channels)

    @staticmethod
    def consume_fit_shuffle(n_samples):
        """Keras `Model.fit(shuffle=True)` shuffles the sample indices with the GLOBAL numpy RNG on every call, even when
        the data is a single batch -- the same generator the epsilon-greedy policy and Memory.sample draw
        from (BS_brain.py:261,268,330,333).  The one-batch paths (fit_arrays, the device-resident replay) skip the
        gather but consume the identical draw, so a seeded run visits the same RNG stream on every path."""
        if n_samples > 1:
            np.random.shuffle(np.arange(n_samples))

    def fit_arrays(self, x, e, adj, y, nbr=None, presharded=False, n_global=None):
        """One Adam step on the whole batch (what the reference's fit call amounts to, BS_brain.py:218-223);
        y [B, N, C].  -> History with the same keys as fit.  presharded / n_global: see _train_step."""
        self.consume_fit_shuffle(np.shape(x)[0])
        loss = self._train_step(PackedBatch.from_dense(x, e, adj, nbr),
                                np.asarray(y, np.float32).reshape(-1, self.spec.n_channels), presharded, n_global)
        loss = np.asarray(loss, np.float64)
        hist = History()
        hist.epoch.append(0)
        hist.history['loss'] = [float(loss.sum())]
        for k, name in enumerate(self.output_names):
            hist.history[name + '_loss'] = [float(loss[k])]
        return hist

    def train_on_batch(self, x, y):
        h = self.fit(x, y, batch_size=len(next(iter(self._named(x, self.input_names, "input").values()))),
                     epochs=1, shuffle=False)
        return [h.history['loss'][0]] + [h.history[n + '_loss'][0] for n in self.output_names]

    def get_weights(self):
        return self.engine.get_weights()

    def set_weights(self, weights):
        self.engine.set_weights(weights)

    def keras_layer_table(self):
        """[(layer name, [weight names])] of the layers that own weights, in the order of get_weights() -- the names
        Keras gives the reference model's layers (BS_brain.py:121-200).  Auto-named layers get their class's snake-case
        name and a running index in CREATION order: the message-passing layers are created stage by stage, node by node
        (:154-164: gnn_layer_1..N, then N+1..2N), the hidden Dense layers node by node (:176-200: dense_1/2/3 are node
        1's 80/40/20, dense_4/5/6 node 2's, ...).  The running index is global to the Keras session, so the second
        model BS builds (the target network, :106) continues where the first stopped: `model_index` offsets it."""
        sp = self.spec
        slots = list(range(1, sp.n_nodes + 1)) if not sp.share_weights else [None]
        n_slots = len(slots)
        gnn_base = self.model_index * sp.n_mp_layers * n_slots
        dense_base = self.model_index * 3 * n_slots
        table = []
        for stage in range(sp.n_mp_layers + 1):
            for i, k in enumerate(slots):
                if stage == 0:
                    name = 'D%d_GNN' % k if k else 'GNN'
                else:
                    name = 'gnn_layer_%d' % (gnn_base + (stage - 1) * n_slots + i + 1)
                table.append((name, ['%s/%s:0' % (name, w) for w in ('W1', 'W2', 'W3', 'bias')]))
        for layer in range(4):
            for i, k in enumerate(slots):
                if layer == 3:
                    name = 'D%d_Decide_Output' % k if k else 'Decide_Output'
                else:
                    name = 'dense_%d' % (dense_base + 3 * i + layer + 1)
                table.append((name, ['%s/kernel:0' % name, '%s/bias:0' % name]))
        return table

    def save_weights(self, filepath, overwrite=True):
        """Weights only, like Keras `save_weights` (no optimizer state: BS_brain.py:853-870).  A '.h5' / '.hdf5' name
        is written as a Keras-layout HDF5 file through libhdf5 (h5weights.py) when that library can be loaded;
        otherwise (and for any other name) the container is NumPy's .npz under exactly the name the caller gives."""
        from . import h5weights
        weights = self.get_weights()
        if str(filepath).lower().endswith(('.h5', '.hdf5', '.keras.h5')) and h5weights.available():
            it = iter(weights)
            layers = [(ln, [(wn, next(it)) for wn in wns]) for ln, wns in self.keras_layer_table()]
            h5weights.save_keras_weights(filepath, layers)
            return
        arrays = {('w%03d' % i): w for i, w in enumerate(weights)}
        with open(filepath, 'wb') as f:
            np.savez(f, **arrays)

    def load_weights(self, filepath):
        """Keras-layout HDF5 (matched to the model by ORDER of the weighted layers, as Keras does) or the .npz
        container written by save_weights; the format is detected from the file signature."""
        from . import h5weights
        if h5weights.is_hdf5(filepath):
            layers = h5weights.load_keras_weights(filepath)
            self.set_weights([arr for _, ws in layers for _, arr in ws])
            return
        with np.load(filepath) as z:
            self.set_weights([z['w%03d' % i] for i in range(len(z.files))])


class BS(object):
    """Same constructor / attributes / methods as the reference class (BS_brain.py:90-239)."""

    def __init__(self, num_d2d, input_node_info, input_edge_info, num_d2d_feedback, num_d2d_neighbor, num_ch,
                 device=0, seed=None, n_mp_layers=2, share_weights=False, use_graph=False, data_parallel=False,
                 process_group=None, engine_factory=None):
        self.num_D2D = num_d2d
        self.num_Neighbor = num_d2d_neighbor
        self.num_CH = num_ch
        self.num_Feedback = num_d2d_feedback
        self.input_node_Info = input_node_info
        self.input_edge_Info = input_edge_info
        self.num_One_Node_Input = ((input_node_info - 1) * self.num_CH + 1) * self.num_Neighbor
        self.num_One_Edge_Input = input_edge_info * self.num_CH
        self.num_One_D2D_Input = self.num_One_Node_Input + self.num_One_Edge_Input
        self.num_D2D_Input = num_d2d * self.num_One_D2D_Input + self.num_D2D ** 2
        self._spec = GnnSpec(n_nodes=num_d2d, n_channels=num_ch, feat_dim=num_d2d_feedback,
                             n_mp_layers=n_mp_layers, share_weights=share_weights,
                             input_node_info=input_node_info, input_edge_info=input_edge_info,
                             n_neighbor=num_d2d_neighbor)
        self._device, self._use_graph = device, use_graph
        self._dp, self._group, self._engine_factory = data_parallel, process_group, engine_factory
        ss = np.random.SeedSequence(seed).spawn(2)
        self._seeds = [int(s.generate_state(1)[0]) for s in ss]
        self._n_models = 0
        self._adj_cache = AdjacencyCache()
        self.model = self._create_model()
        self.target_model = self._create_model()

    def _create_model(self):
        seed = self._seeds.pop(0) if self._seeds else None
        engine = self._engine_factory(self._spec) if self._engine_factory is not None else None
        index = self._n_models
        self._n_models += 1
        return GnnQModel(self._spec, device=self._device, seed=seed, use_graph=self._use_graph,
                         data_parallel=self._dp, process_group=self._group, engine=engine, model_index=index,
                         adjacency_cache=self._adj_cache)

    def train_dnn(self, data_train, labels, batch_size):
        epochs = 1
        Train_Result = self.model.fit(data_train, labels, batch_size=batch_size, epochs=epochs, verbose=0)
        return Train_Result

    def predict(self, data_test, target=False):
        if target:
            return self.target_model.predict(data_test)
        return self.model.predict(data_test)

    def predict_one_step(self, data_test, target=False):
        return self.predict(data_test, target=target)

    def update_target_model(self):
        # one device-to-device copy instead of get_weights()/set_weights() through the host
        self.target_model.engine.copy_weights_from(self.model.engine)
