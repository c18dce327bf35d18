"""ctypes binding of libv2xgnn.so (the C ABI in include/v2xgnn.h).

The library is built in-tree (`csrc/Makefile`, or `__graft_entry__.build()`); if it is
missing this module raises -- there is deliberately no Python/CPU fallback for the hot path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

V2X_OK = 0


class V2XError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("n_channels", C.c_int32), ("feat_dim", C.c_int32),
                ("n_mp_layers", C.c_int32), ("share_weights", C.c_int32), ("variable_graphs", C.c_int32),
                ("device", C.c_int32), ("use_graph", C.c_int32),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float)]


class Batch(C.Structure):
    _fields_ = [("n_graphs", C.c_int32), ("n_rows", C.c_int32), ("n_edges", C.c_int32),
                ("max_nodes", C.c_int32), ("max_edges", C.c_int32), ("on_device", C.c_int32),
                ("xe", C.c_void_p), ("nbr_init", C.c_void_p), ("graph_off", C.c_void_p),
                ("row_ptr", C.c_void_p), ("col_idx", C.c_void_p)]


class ForwardClosure(C.Structure):
    """v2x_forward_closure of include/v2xgnn.h"""
    _fields_ = [("m", C.c_void_p), ("b", Batch), ("q_out", C.c_void_p), ("q_on_device", C.c_int32), ("pad_", C.c_int32),
                ("stream", C.c_void_p)]


class Feed(C.Structure):
    _fields_ = [("n_graphs", C.c_int32), ("n_nodes", C.c_int32), ("feat_dim", C.c_int32), ("node_in", C.c_int32),
                ("edge_in", C.c_int32), ("node", C.POINTER(C.c_void_p)), ("edge", C.POINTER(C.c_void_p)),
                ("nbr", C.POINTER(C.c_void_p)), ("is_f64", C.POINTER(C.c_uint8)), ("adjacency", C.c_void_p)]


# every symbol include/v2xgnn.h declares: (name, restype, argtypes)
_P, _I, _L, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float
SYMBOLS = [
    ("v2x_create", C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    ("v2x_destroy", None, [_P]),
    ("v2x_last_error", C.c_char_p, [_P]),
    ("v2x_version", C.c_char_p, []),
    ("v2x_param_count", _L, [_P]),
    ("v2x_get_weights", C.c_int, [_P, _P, _P]),
    ("v2x_set_weights", C.c_int, [_P, _P, _P]),
    ("v2x_copy_weights", C.c_int, [_P, _P, _P]),
    ("v2x_get_optimizer_state", C.c_int, [_P, _P, _P, C.POINTER(_L), _P]),
    ("v2x_set_optimizer_state", C.c_int, [_P, _P, _P, _L, _P]),
    ("v2x_param_ptr", _P, [_P]),
    ("v2x_grad_ptr", _P, [_P]),
    ("v2x_forward", C.c_int, [_P, C.POINTER(Batch), _P, C.c_int, _P]),
    ("v2x_forward_call", C.c_int, [_P]),
    ("v2x_train_step", C.c_int, [_P, C.POINTER(Batch), _P, C.c_int, _I, _P, C.c_int, _P]),
    ("v2x_forward_backward", C.c_int, [_P, C.POINTER(Batch), _P, C.c_int, _I, _P, C.c_int, _P]),
    ("v2x_apply_gradients", C.c_int, [_P, _P]),
    ("v2x_forward_backward_phase", C.c_int, [_P, C.POINTER(Batch), _P, C.c_int, _I, C.c_int, _P, C.c_int, _P]),
    ("v2x_grad_bucket", _L, [_P, C.c_int, C.POINTER(_L)]),
    ("v2x_grad_bucket_count", C.c_int, [_P]),
    ("v2x_apply_gradients_range", C.c_int, [_P, _L, _L, C.c_int, _P]),
    ("v2x_agg_fwd", C.c_int, [C.POINTER(Batch), _I, _I, _P, _P, _P]),
    ("v2x_agg_bwd", C.c_int, [C.POINTER(Batch), _I, _I, _P, _P, _P]),
    ("v2x_node_update_fwd", C.c_int, [_P, _I, _I, _P, _P, _P, _P, _P]),
    ("v2x_node_update_bwd", C.c_int, [_P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    ("v2x_mlp_fwd", C.c_int, [_P, _I, _P, _P, _P, _P, _P]),
    ("v2x_mlp_huber_bwd", C.c_int, [_P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    ("v2x_adam_step", C.c_int, [_P, _P, _P, _P, _L, _L, _F, _F, _F, _F, _P]),
    ("v2x_device_addressable", C.c_int, [_P]),
    ("v2x_gather_rows", C.c_int, [_P, _P, _P, _L, _L, _P]),
    ("v2x_gather_rows_multi", C.c_int, [_I, _P, _P, _P, _P, _L, _P]),
    ("v2x_q_stats", C.c_int, [_P, _I, _I, _I, _P, _P]),
    ("v2x_dqn_targets", C.c_int, [_P, _P, _P, _P, C.c_double, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    ("v2x_dqn_step", C.c_int, [_P, _P, _P, _P, _P, _P, C.c_double, C.c_int32, _P, _P, C.c_int, _P]),
    ("v2x_pack_feed", C.c_int, [C.POINTER(Feed), C.c_int, _P, _P, _P, _P, _P]),
    ("v2x_validate_batch", C.c_int, [_P, C.POINTER(Batch), _I, _P]),
    ("v2x_check_errors", C.c_int, [_P, _P]),
    ("v2x_reset_exchange", C.c_int, [_P]),
    ("v2x_debug_exchange_counters", _P, [_P]),
    ("v2x_debug_split_counters", _P, [_P, C.POINTER(C.c_int32)]),
    ("v2x_debug_phase_stamps", C.c_int, [_P, _P, C.c_int]),
    ("v2x_debug_ragged_plan", C.c_int, [_P, _P, C.c_int]),
    ("v2x_profile_enable", C.c_int, [_P, C.c_int]),
    ("v2x_path_info", C.c_int, [_P, _P, C.c_char_p, C.c_int]),
    ("v2x_profile_read", C.c_int, [_P, C.c_char_p, C.c_int, C.POINTER(C.c_double), C.POINTER(_L), C.c_int]),
]


def library_path():
    return os.environ.get("V2XGNN_LIB", os.path.join(_HERE, "libv2xgnn.so"))


def load_library():
    """dlopen libv2xgnn.so and bind every exported symbol.  Raises V2XError when the HIP
    extension has not been built (run `python -c "import __graft_entry__ as g; g.build()"`)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise V2XError("HIP extension %s is missing: build it with `make -C %s` (no CPU fallback exists)"
                       % (path, os.path.join(_HERE, "csrc")))
    # PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so).  Whichever copy of that soname is mapped
    # first serves the whole process, and torch does not find its GPUs through /opt/rocm's copy: load torch's first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        lib = C.CDLL(path)
    except OSError as exc:
        raise V2XError("cannot load %s: %s" % (path, exc))
    for name, restype, argtypes in SYMBOLS:
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise V2XError("%s does not export %s (stale build?)" % (path, name))
        fn.restype = restype
        fn.argtypes = argtypes
    _LIB = lib
    return lib


def check(lib, rc, handle=None):
    if rc != V2X_OK:
        msg = lib.v2x_last_error(handle)
        text = msg.decode() if msg else "unknown error"
        if rc == -1:
            raise ValueError(text)          # V2X_EINVAL: same exception class Keras raises on bad inputs
        raise V2XError("v2xgnn error %d: %s" % (rc, text))
