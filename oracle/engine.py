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
