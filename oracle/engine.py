"""ORACLE -- TEST INFRASTRUCTURE: the CPU oracle behind GnnEngine's interface, so that the host logic above the C ABI
(BS / GnnQModel / Agent / DataParallelTrainer) can be exercised on CPU, incl. world_size-2 gloo runs, and so that
bench.py's cpu_baseline leg can time the DQN loop of BASELINE configs[0] on the host.  The product never imports this;
its default engine is the gfx950 one and raises without a GPU."""
import numpy as np
import torch

from . import compact as oc
from .keras_semantics import KerasAdam
from .spec import GnnSpec as _OSpec


def ospec(spec):
    return _OSpec(n_nodes=spec.n_nodes, n_channels=spec.n_channels, feat_dim=spec.feat_dim,
                  n_mp_layers=spec.n_mp_layers, share_weights=spec.share_weights)


class OracleEngine(object):
    def __init__(self, spec, params=None, dtype=np.float64):
        self.spec, self.os, self.dtype = spec, ospec(spec), dtype
        if params is None:
            params = oc.init_params(self.os, np.random.default_rng(0), np.float64)
        self.params = oc.cast_params(params, dtype)
        self.opt = KerasAdam()
        self.n_params = sum(a.size for a in oc.param_arrays(self.params))
        self._g = torch.zeros(self.n_params, dtype=torch.float64)

    # ---- weights
    def set_weights(self, weights):
        self.params = oc.params_from_list(self.os, [np.asarray(w, self.dtype) for w in weights], self.dtype)

    def get_weights(self):
        return [np.asarray(w, np.float32) for w in oc.params_to_list(self.params)]

    def copy_weights_from(self, other):
        self.params = oc.cast_params(other.params, self.dtype)

    # ---- compute
    def _inputs(self, batch):
        goff = batch.graph_off if getattr(batch, "graph_off", None) is not None else \
            (np.arange(batch.n_graphs + 1) * batch.n_nodes).astype(np.int32)
        graph = (goff, batch.row_ptr, batch.col_idx)
        M = oc.csr_to_matrix(*graph, dtype=self.dtype)
        return batch.xe[:, :9].astype(self.dtype), batch.xe[:, 9:13].astype(self.dtype), M

    def forward(self, batch):
        x, e, M = self._inputs(batch)
        q, _ = oc.forward(self.os, self.params, x, e, M)
        return np.asarray(q, np.float32)

    def forward_backward(self, batch, y, n_global=None, want_loss=True):
        x, e, M = self._inputs(batch)
        q, cache = oc.forward(self.os, self.params, x, e, M)
        loss, dq = oc.huber_loss_and_grad(self.os, q, np.asarray(y, self.dtype), n_global)
        g = oc.backward(self.os, self.params, cache, dq)
        self._g.copy_(torch.from_numpy(np.concatenate([np.asarray(a, np.float64).ravel() for a in oc.param_arrays(g)])))
        return loss

    def grad_tensor(self):
        return self._g

    # ---- the phased step and the sharded optimizer step of v2xgnn.dp (same contract as GnnEngine's)
    phases_on_host = True

    def _layer_sizes(self):
        gnn = [sum(a.size for a in st.values()) for st in self.params['gnn']]
        dense = [sum(a.size for a in st.values()) for st in self.params['dense']]
        return gnn, dense

    def grad_buckets(self):
        """[(offset, count)] in the order the buckets become final; this engine's flat order is layer-major like the
        library's: graph layers 0..L, then the Dense layers."""
        gnn, dense = self._layer_sizes()
        offs = np.concatenate([[0], np.cumsum(gnn + dense)])
        L = len(gnn) - 1
        if self.spec.feat_dim < 128:
            return [(int(offs[L + 1]), int(sum(dense))), (0, int(offs[L + 1]))]
        return [(int(offs[L + 1]), int(sum(dense)))] + [(int(offs[s]), int(gnn[s])) for s in range(L, -1, -1)]

    def forward_backward_phase(self, batch, y, phase, n_global=None, want_loss=True):
        """Phase 0 computes the step; phase k RELEASES bucket k into grad_tensor() -- everything not yet released reads NaN,
        so a collective started too early poisons the result."""
        buckets = self.grad_buckets()
        if phase == 0:
            self._phase_loss = self.forward_backward(batch, y, n_global, want_loss)
            self._full = self._g.clone()
            self._g.fill_(float('nan'))
        off, n = buckets[phase]
        self._g[off:off + n] = self._full[off:off + n]
        return self._phase_loss if phase == len(buckets) - 1 else None

    def param_tensor(self):
        """flat float64 tensor of the parameters; params_changed() writes it back into the parameter arrays"""
        self._pflat = torch.from_numpy(np.concatenate([np.asarray(a, np.float64).ravel() for a in oc.param_arrays(self.params)]))
        return self._pflat

    def _sync_pflat(self, offset, count):
        """the updated range only: an all-gather into another range of the tensor may be in flight"""
        if getattr(self, "_pflat", None) is not None:
            flat = np.concatenate([np.asarray(a, np.float64).ravel() for a in oc.param_arrays(self.params)])
            self._pflat[offset:offset + count] = torch.from_numpy(flat[offset:offset + count])

    def params_changed(self):
        flat, pos = self._pflat.numpy(), 0
        for a in oc.param_arrays(self.params):
            a[...] = flat[pos:pos + a.size].reshape(a.shape).astype(self.dtype)
            pos += a.size

    def apply_gradients_range(self, offset, count, advance_iteration):
        """Keras Adam (keras_semantics.KerasAdam's expressions) on the flat range only; moments outside it are untouched."""
        if self.opt.m is None:
            self.opt.m = [np.zeros_like(p) for p in oc.param_arrays(self.params)]
            self.opt.v = [np.zeros_like(p) for p in oc.param_arrays(self.params)]
        if advance_iteration:
            self.opt.iterations += 1
        t = self.opt.iterations
        dt = self.dtype
        lr_t = dt(self.opt.lr * np.sqrt(1.0 - self.opt.b2 ** t) / (1.0 - self.opt.b1 ** t))
        b1, b2, eps, one = dt(self.opt.b1), dt(self.opt.b2), dt(self.opt.eps), dt(1.0)
        g_all, pos = self._g.numpy(), 0
        for p, m, v in zip(oc.param_arrays(self.params), self.opt.m, self.opt.v):
            lo, hi = max(offset, pos), min(offset + count, pos + p.size)
            if lo < hi:
                sl = slice(lo - pos, hi - pos)
                pf, mf, vf = p.reshape(-1), m.reshape(-1), v.reshape(-1)
                g = g_all[lo:hi].astype(dt)
                mf[sl] = b1 * mf[sl] + (one - b1) * g
                vf[sl] = b2 * vf[sl] + (one - b2) * (g * g)
                pf[sl] -= lr_t * mf[sl] / (np.sqrt(vf[sl]) + eps)
            pos += p.size
        self._sync_pflat(offset, count)

    def get_optimizer_state(self):
        z = lambda arrs: np.concatenate([np.asarray(a, np.float64).ravel() for a in arrs])
        if self.opt.m is None:
            return np.zeros(self.n_params), np.zeros(self.n_params), self.opt.iterations
        return z(self.opt.m), z(self.opt.v), self.opt.iterations

    def apply_gradients(self):
        flat = self._g.numpy()
        grads, pos = [], 0
        for a in oc.param_arrays(self.params):
            grads.append(flat[pos:pos + a.size].reshape(a.shape).astype(self.dtype))
            pos += a.size
        self.opt.step(oc.param_arrays(self.params), grads)

    def train_step(self, batch, y, n_global=None, want_loss=True):
        loss = self.forward_backward(batch, y, n_global, want_loss)
        self.apply_gradients()
        return loss
