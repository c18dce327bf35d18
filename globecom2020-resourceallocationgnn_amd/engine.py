"""GnnEngine: one Q-network (weights + Adam state + workspaces) resident in the HBM of one
MI355X, driven through the C ABI (include/v2xgnn.h).  PyTorch is used only as plumbing:
device tensors for batches that stay resident, the current HIP stream, torch.distributed.
"""
import ctypes as C

import numpy as np

from . import lib as _lib
from .packing import PackedBatch, keras_list_to_flat, flat_to_keras_list
from .spec import GnnSpec


def _torch():
    import torch
    return torch


def current_stream_ptr(device_index):
    """hipStream_t of torch's current stream on the device (0 when torch has no GPU)."""
    try:
        torch = _torch()
        if torch.cuda.is_available():
            return int(torch.cuda.current_stream(device_index).cuda_stream)
    except ImportError:
        pass
    return 0


class DeviceBatch(object):
    """A PackedBatch whose arrays live in HBM (torch tensors own the memory)."""

    def __init__(self, host: PackedBatch, device):
        torch = _torch()
        self.n_graphs, self.n_nodes, self.n_rows, self.n_edges = host.n_graphs, host.n_nodes, host.n_rows, host.n_edges
        self.max_nodes, self.max_edges = host.max_nodes, host.max_edges
        self.device = torch.device(device)
        t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        self.xe, self.nbr = t(host.xe), t(host.nbr)
        self.row_ptr, self.col_idx, self.graph_off = t(host.row_ptr), t(host.col_idx), t(host.graph_off)
        if self.col_idx is not None and self.col_idx.numel() == 0:
            self.col_idx = torch.zeros(1, dtype=torch.int32, device=self.device)

    @classmethod
    def from_tensors(cls, n_graphs, n_nodes, xe, row_ptr, col_idx, max_edges, nbr=None, graph_off=None, max_nodes=None):
        """Wrap tensors that are ALREADY resident in HBM (float32 xe [R,16], int32 CSR); no copies."""
        self = cls.__new__(cls)
        self.n_graphs, self.n_nodes, self.n_rows = int(n_graphs), int(n_nodes), int(xe.shape[0])
        self.n_edges = int(col_idx.numel()) if max_edges > 0 else 0
        self.max_nodes, self.max_edges = int(max_nodes if max_nodes is not None else n_nodes), int(max_edges)
        self.device = xe.device
        self.xe, self.nbr, self.row_ptr, self.col_idx, self.graph_off = xe, nbr, row_ptr, col_idx, graph_off
        return self


def _batch_struct(b):
    s = _lib.Batch()
    s.n_graphs, s.n_rows, s.n_edges = b.n_graphs, b.n_rows, b.n_edges
    s.max_nodes, s.max_edges = b.max_nodes, b.max_edges
    if isinstance(b, DeviceBatch):
        s.on_device = 1
        p = lambda t: None if t is None else t.data_ptr()
    else:
        s.on_device = 0
        p = lambda a: None if a is None else a.ctypes.data
    s.xe, s.nbr_init, s.graph_off = p(b.xe), p(b.nbr), p(b.graph_off)
    s.row_ptr, s.col_idx = p(b.row_ptr), p(b.col_idx)
    return s


class GnnEngine(object):
    """use_graph: forward / fit steps are captured once per (batch pointers, sizes) as a hipGraph and replayed.
    Capture needs a NON-default stream: calls made while torch's current stream is the default one run eagerly
    (`with torch.cuda.stream(torch.cuda.Stream()): ...` enables the graphs).  Off by default: a replay saves host calls, not GPU time --
    on ROCm 7.2 one hipGraphLaunch costs 3.6-6 us per fit step MORE than the launches it replaces when the host runs ahead of the GPU
    (profiles/r06_launch_form.txt); it pays for host-bound loops."""

    def __init__(self, spec: GnnSpec, device=0, use_graph=False, lr=1e-3, beta_1=0.5, beta_2=0.999,
                 epsilon=1e-7):
        self.spec = spec
        self.device = int(device)
        self._lib = _lib.load_library()          # raises if the HIP extension is missing
        cfg = _lib.Config(spec.n_nodes, spec.n_channels, spec.feat_dim, spec.n_mp_layers,
                          int(spec.share_weights), int(spec.variable_graphs), self.device, int(use_graph),
                          lr, beta_1, beta_2, epsilon)
        h = C.c_void_p()
        rc = self._lib.v2x_create(C.byref(cfg), C.byref(h))
        _lib.check(self._lib, rc, None)
        self._h = h
        self.n_params = int(self._lib.v2x_param_count(self._h))
        if self.n_params != spec.n_params:
            raise _lib.V2XError("parameter count mismatch: library %d vs spec %d" % (self.n_params, spec.n_params))
        self.n_outputs = 1 if spec.variable_graphs else spec.n_nodes

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_h", None):
            self._lib.v2x_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return current_stream_ptr(self.device)

    def _check(self, rc):
        _lib.check(self._lib, rc, self._h)

    # ------------------------------------------------------------------ parameters
    def get_flat(self):
        out = np.empty(self.n_params, np.float32)
        self._check(self._lib.v2x_get_weights(self._h, out.ctypes.data, self._stream()))
        return out

    def set_flat(self, flat):
        flat = np.ascontiguousarray(flat, np.float32)
        if flat.size != self.n_params:
            raise ValueError("expected %d parameters, got %d" % (self.n_params, flat.size))
        self._check(self._lib.v2x_set_weights(self._h, flat.ctypes.data, self._stream()))

    def get_weights(self):
        return flat_to_keras_list(self.spec, self.get_flat())

    def set_weights(self, weights):
        self.set_flat(keras_list_to_flat(self.spec, weights))

    def copy_weights_from(self, other):
        """update_target_model (BS_brain.py:237-239) as ONE device-to-device copy."""
        self._check(self._lib.v2x_copy_weights(self._h, other._h, self._stream()))

    def get_optimizer_state(self):
        m = np.empty(self.n_params, np.float32)
        v = np.empty(self.n_params, np.float32)
        it = C.c_int64()
        self._check(self._lib.v2x_get_optimizer_state(self._h, m.ctypes.data, v.ctypes.data, C.byref(it), self._stream()))
        return m, v, int(it.value)

    def set_optimizer_state(self, m, v, iterations):
        m = np.ascontiguousarray(m, np.float32)
        v = np.ascontiguousarray(v, np.float32)
        self._check(self._lib.v2x_set_optimizer_state(self._h, m.ctypes.data, v.ctypes.data, int(iterations), self._stream()))

    def grad_tensor(self):
        """torch view (no copy) of the flat gradient buffer in HBM, for the RCCL all-reduce."""
        torch = _torch()
        ptr = int(self._lib.v2x_grad_ptr(self._h))

        class _Holder(object):
            pass
        hld = _Holder()
        hld.__cuda_array_interface__ = {"shape": (self.n_params,), "typestr": "<f4", "data": (ptr, False),
                                        "version": 2, "strides": None}
        t = torch.as_tensor(hld, device="cuda:%d" % self.device)
        t._v2x_owner = self
        return t

    def get_grad_flat(self):
        return self.grad_tensor().cpu().numpy()

    # ------------------------------------------------------------------ hot path
    def to_device(self, batch: PackedBatch):
        return DeviceBatch(batch, "cuda:%d" % self.device)

    def forward(self, batch, out=None):
        """Model.predict on a packed batch.  Returns q[R,C]: numpy for host batches, or fills /
        returns a torch tensor for device batches."""
        s = _batch_struct(batch)
        if isinstance(batch, DeviceBatch):
            torch = _torch()
            if out is None:
                out = torch.empty((batch.n_rows, self.spec.n_channels), dtype=torch.float32, device=batch.device)
            self._check(self._lib.v2x_forward(self._h, C.byref(s), out.data_ptr(), 1, self._stream()))
            return out
        q = np.empty((batch.n_rows, self.spec.n_channels), np.float32)
        self._check(self._lib.v2x_forward(self._h, C.byref(s), q.ctypes.data, 0, self._stream()))
        return q

    def forward_to_host(self, batch, q):
        """forward() of a device-addressable batch with the Q-values copied into the host array q [R, C] float32 and the stream
        synchronised inside the call (the library's own device-to-host copy: one call instead of forward + copy + synchronise).
        The batch's arrays may live in pinned host memory (device-mapped under unified addressing): the kernels then read the
        few kilobytes of a rollout batch straight over the bus and no copy launch is involved at all."""
        if not isinstance(batch, DeviceBatch):
            raise ValueError("forward_to_host takes a DeviceBatch")
        if not getattr(batch, "_addressable", False):        # once per batch object: its tensors do not change under it
            for name in ("xe", "nbr", "row_ptr", "col_idx", "graph_off"):
                t = getattr(batch, name)
                if t is not None and not t.is_cuda and self._lib.v2x_device_addressable(t.data_ptr()) != 1:
                    raise ValueError("forward_to_host: batch.%s is host memory the device cannot address (pageable, or mapped at "
                                     "another address): pin it (tensor.pin_memory()) or move it to the device" % name)
            batch._addressable = True
        if q.dtype != np.float32 or not q.flags.c_contiguous or q.size != batch.n_rows * self.spec.n_channels:
            raise ValueError("q must be a C-contiguous float32 array of %d x %d" % (batch.n_rows, self.spec.n_channels))
        s = _batch_struct(batch)
        self._check(self._lib.v2x_forward(self._h, C.byref(s), q.ctypes.data, 0, self._stream()))
        return q

    def _step(self, fn, batch, y, n_global, want_loss):
        s = _batch_struct(batch)
        n_global = int(n_global or 0)
        if isinstance(batch, DeviceBatch):
            torch = _torch()
            if not (hasattr(y, "is_cuda") and y.is_cuda):
                y = torch.as_tensor(np.ascontiguousarray(y, np.float32)).to(batch.device)
            loss = torch.empty(self.n_outputs, dtype=torch.float32, device=batch.device) if want_loss else None
            self._check(fn(self._h, C.byref(s), y.data_ptr(), 1, n_global,
                           None if loss is None else loss.data_ptr(), 1, self._stream()))
            return loss
        y = np.ascontiguousarray(y, np.float32)
        if y.size != batch.n_rows * self.spec.n_channels:
            raise ValueError("targets have %d entries, expected %d" % (y.size, batch.n_rows * self.spec.n_channels))
        loss = np.empty(self.n_outputs, np.float32) if want_loss else None
        self._check(fn(self._h, C.byref(s), y.ctypes.data, 0, n_global,
                       None if loss is None else loss.ctypes.data, 0, self._stream()))
        return loss

    def train_step(self, batch, y, n_global=None, want_loss=True):
        """One Model.fit step (forward + Huber + backward + Keras Adam).  y is [R,C]."""
        return self._step(self._lib.v2x_train_step, batch, y, n_global, want_loss)

    def forward_backward(self, batch, y, n_global=None, want_loss=True):
        """Forward + backward only: the local gradient is left in grad_tensor()."""
        return self._step(self._lib.v2x_forward_backward, batch, y, n_global, want_loss)

    def forward_backward_phase(self, batch, y, phase, n_global=None, want_loss=True):
        """forward_backward in len(grad_buckets()) calls (v2x_forward_backward_phase): after phase k bucket k of the gradient
        is final; the last phase also gives the losses.  Device batches only."""
        fn = lambda h, s, yp, yd, ng, lo, ld, st: self._lib.v2x_forward_backward_phase(h, s, yp, yd, ng, int(phase), lo, ld, st)
        last = int(self._lib.v2x_grad_bucket_count(self._h)) - 1
        return self._step(fn, batch, y, n_global, want_loss and phase == last)

    def param_tensor(self):
        """torch view (no copy) of the flat parameter buffer in HBM (the all-gather of a sharded optimizer step writes it).
        Taking it tells the library that the parameters may change behind its back (v2x_param_ptr)."""
        torch = _torch()
        ptr = int(self._lib.v2x_param_ptr(self._h))

        class _Holder(object):
            pass
        hld = _Holder()
        hld.__cuda_array_interface__ = {"shape": (self.n_params,), "typestr": "<f4", "data": (ptr, False),
                                        "version": 2, "strides": None}
        t = torch.as_tensor(hld, device="cuda:%d" % self.device)
        t._v2x_owner = self
        return t

    def apply_gradients_range(self, offset, count, advance_iteration):
        """Keras Adam on parameters [offset, offset + count) only (v2x_apply_gradients_range)."""
        self._check(self._lib.v2x_apply_gradients_range(self._h, int(offset), int(count), 1 if advance_iteration else 0,
                                                        self._stream()))

    def grad_buckets(self):
        """[(offset, count)] of the all-reduce buckets inside grad_tensor(), in the order they become final: phase k of
        forward_backward_phase completes bucket k."""
        out = []
        for b in range(int(self._lib.v2x_grad_bucket_count(self._h))):
            off = C.c_int64()
            n = int(self._lib.v2x_grad_bucket(self._h, b, C.byref(off)))
            out.append((int(off.value), n))
        return out

    def dqn_step(self, target, batch, batch_next, action, reward, gamma, y_out=None, n_global=None, want_loss=True):
        """One replay step on device-resident data (v2x_dqn_step): target forward on s', online forward on s, target
        rule, backward + Adam on this (online) engine.  action [B, N] int32, reward [B] float64 (device tensors)."""
        torch = _torch()
        if not isinstance(batch, DeviceBatch) or not isinstance(batch_next, DeviceBatch):
            raise ValueError("dqn_step takes device-resident batches")
        sb, sn = _batch_struct(batch), _batch_struct(batch_next)
        loss = torch.empty(self.n_outputs, dtype=torch.float32, device=batch.device) if want_loss else None
        self._check(self._lib.v2x_dqn_step(self._h, target._h, C.byref(sb), C.byref(sn), action.data_ptr(), reward.data_ptr(),
                                           float(gamma), int(n_global or 0), None if y_out is None else y_out.data_ptr(),
                                           None if loss is None else loss.data_ptr(), 1, self._stream()))
        return loss

    def validate(self, batch):
        """Check a batch against the layout contract of include/v2xgnn.h (sizes vs max_nodes / max_edges, sources
        inside their graph and strictly ascending); raises ValueError.  Host batches are checked on every call
        anyway; device batches only here."""
        s = _batch_struct(batch)
        self._check(self._lib.v2x_validate_batch(self._h, C.byref(s), self.spec.n_nodes, self._stream()))

    def check_errors(self):
        """Synchronise and raise if a kernel met a graph larger than the LDS tile the batch's max_nodes / max_edges
        promised (its outputs are then invalid)."""
        self._check(self._lib.v2x_check_errors(self._h, self._stream()))

    def reset_exchange(self):
        """Re-arm the one-launch predict's exchange buffer (v2x_reset_exchange); a predict that times out does it by itself."""
        self._check(self._lib.v2x_reset_exchange(self._h))

    def apply_gradients(self):
        self._check(self._lib.v2x_apply_gradients(self._h, self._stream()))

    # ------------------------------------------------------------------ measurement
    def path_info(self, batch):
        """{'graph_layers': 'fused', 'aggregation': 'complement' | 'edge-bitset-walk' (fused kernels) | 'edge-gather' (k_agg) | 'dense(complement-or-mfma-per-graph)', ...}: the kernels a fit step
        of this batch runs (v2x_path_info)."""
        buf = C.create_string_buffer(256)
        s = _batch_struct(batch)
        self._check(self._lib.v2x_path_info(self._h, C.byref(s), buf, 256))
        return dict(kv.split("=", 1) for kv in buf.value.decode().split())

    def profile(self, enable):
        self._check(self._lib.v2x_profile_enable(self._h, 1 if enable else 0))

    def profile_read(self):
        """{kernel name: (calls, total ms)} measured with HIP events around every launch."""
        cap, n = 4096, 64
        names = C.create_string_buffer(cap)
        ms = (C.c_double * n)()
        calls = (C.c_int64 * n)()
        k = self._lib.v2x_profile_read(self._h, names, cap, ms, calls, n)
        if k < 0:
            self._check(k)
        nm = names.value.decode().split("\n")
        return {nm[i]: (int(calls[i]), float(ms[i])) for i in range(k)}
