"""Keras-layout HDF5 weight files (`Model.save_weights` / `load_weights`, call sites BS_brain.py:863,869,1254,1256;
SURVEY.md 8 f3) through the HDF5 C library itself, bound with ctypes (h5py is not installed on the target image,
libhdf5 is).  Layout written and read (keras/engine/saving.py `save_weights_to_hdf5_group`, Keras 2.2.4):

    /                      attrs: layer_names [S..], backend, keras_version
    /<layer>/              attrs: weight_names [S..]           e.g.  D1_GNN/W1:0, dense_3/kernel:0
    /<layer>/<weight name> float32 dataset (the '/' inside the weight name makes a nested group, as h5py does)

Keras matches a file to a model by ORDER (layers that own weights, in `model.layers` order), not by name, so the
names only have to be well-formed; the order is the one `GnnQModel.get_weights()` uses.  If libhdf5 cannot be
loaded, `available()` is False and the callers fall back to the .npz container.
"""
import ctypes as C
import ctypes.util
import glob
import os

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_H = None

hid_t = C.c_int64
hsize_t = C.c_uint64
_P = C.c_void_p


class H5Error(RuntimeError):
    pass


def _candidates():
    env = os.environ.get("V2XGNN_HDF5_LIB")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5") or ctypes.util.find_library("hdf5_serial")
    if found:
        yield found
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so*",
                "/usr/lib/x86_64-linux-gnu/libhdf5.so*", "/usr/lib64/libhdf5.so*", "/usr/local/lib/libhdf5.so*"):
        for p in sorted(glob.glob(pat)):
            yield p


def _load():
    global _H
    if _H is not None:
        return _H or None
    for path in _candidates():
        try:
            lib = C.CDLL(path)
            lib.H5open.restype = C.c_int
            if lib.H5open() < 0:
                continue
        except (OSError, AttributeError):
            continue
        sig = {
            "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
            "H5Fclose": (C.c_int, [hid_t]), "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]),
            "H5Gopen2": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Gclose": (C.c_int, [hid_t]),
            "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), _P]), "H5Screate": (hid_t, [C.c_int]),
            "H5Sclose": (C.c_int, [hid_t]), "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
            "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), _P]),
            "H5Sget_simple_extent_npoints": (C.c_int64, [hid_t]),
            "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
            "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Dclose": (C.c_int, [hid_t]),
            "H5Dwrite": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, _P]),
            "H5Dread": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, _P]), "H5Dget_space": (hid_t, [hid_t]),
            "H5Acreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]),
            "H5Aopen": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Aclose": (C.c_int, [hid_t]),
            "H5Awrite": (C.c_int, [hid_t, hid_t, _P]), "H5Aread": (C.c_int, [hid_t, hid_t, _P]),
            "H5Aget_type": (hid_t, [hid_t]), "H5Aget_space": (hid_t, [hid_t]), "H5Aexists": (C.c_int, [hid_t, C.c_char_p]),
            "H5Tcopy": (hid_t, [hid_t]), "H5Tset_size": (C.c_int, [hid_t, C.c_size_t]), "H5Tget_size": (C.c_size_t, [hid_t]),
            "H5Tset_strpad": (C.c_int, [hid_t, C.c_int]), "H5Tis_variable_str": (C.c_int, [hid_t]),
            "H5Tclose": (C.c_int, [hid_t]), "H5Dvlen_reclaim": (C.c_int, [hid_t, hid_t, hid_t, _P]),
            "H5Eset_auto2": (C.c_int, [hid_t, _P, _P]),
        }
        try:
            for name, (res, args) in sig.items():
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
            lib.T_FLOAT = hid_t.in_dll(lib, "H5T_NATIVE_FLOAT_g").value
            lib.T_F32LE = hid_t.in_dll(lib, "H5T_IEEE_F32LE_g").value
            lib.T_C_S1 = hid_t.in_dll(lib, "H5T_C_S1_g").value
        except (AttributeError, ValueError):
            continue
        lib.H5Eset_auto2(0, None, None)            # errors come back as negative ids; no stderr stack dumps
        _H = lib
        return lib
    _H = False
    return None


def available():
    return _load() is not None


def is_hdf5(path):
    try:
        with open(path, "rb") as f:
            return f.read(8) == _SIG
    except OSError:
        return False


def _chk(v, what):
    if v < 0:
        raise H5Error("HDF5: %s failed" % what)
    return v


def _str_type(h, size):
    t = _chk(h.H5Tcopy(h.T_C_S1), "H5Tcopy")
    _chk(h.H5Tset_size(t, max(int(size), 1)), "H5Tset_size")
    _chk(h.H5Tset_strpad(t, 1), "H5Tset_strpad")          # H5T_STR_NULLPAD, what h5py writes for numpy 'S' arrays
    return t


def _write_str_attr(h, loc, name, values, scalar=False):
    vals = [v if isinstance(v, bytes) else str(v).encode("utf8") for v in values]
    width = max([len(v) for v in vals] + [1])
    t = _str_type(h, width)
    if scalar:
        space = _chk(h.H5Screate(0), "H5Screate")           # H5S_SCALAR
    else:
        dims = (hsize_t * 1)(len(vals))
        space = _chk(h.H5Screate_simple(1, dims, None), "H5Screate_simple")
    a = _chk(h.H5Acreate2(loc, name.encode(), t, space, 0, 0), "H5Acreate2 " + name)
    buf = C.create_string_buffer(b"".join(v.ljust(width, b"\0") for v in vals), max(width * len(vals), 1))
    if vals:
        _chk(h.H5Awrite(a, t, buf), "H5Awrite " + name)
    h.H5Aclose(a); h.H5Sclose(space); h.H5Tclose(t)


def _read_str_attr(h, loc, name):
    a = _chk(h.H5Aopen(loc, name.encode(), 0), "H5Aopen " + name)
    t, space = h.H5Aget_type(a), h.H5Aget_space(a)
    n = int(h.H5Sget_simple_extent_npoints(space))
    out = []
    if n > 0:
        if h.H5Tis_variable_str(t) > 0:
            ptrs = (C.c_char_p * n)()
            _chk(h.H5Aread(a, t, ptrs), "H5Aread " + name)
            out = [bytes(p) if p is not None else b"" for p in ptrs]
            h.H5Dvlen_reclaim(t, space, 0, ptrs)
        else:
            size = int(h.H5Tget_size(t))
            buf = C.create_string_buffer(size * n)
            _chk(h.H5Aread(a, t, buf), "H5Aread " + name)
            raw = buf.raw
            out = [raw[i * size:(i + 1) * size].split(b"\0", 1)[0] for i in range(n)]
    h.H5Tclose(t); h.H5Sclose(space); h.H5Aclose(a)
    return out


def save_keras_weights(path, layers, backend=b"tensorflow", keras_version=b"2.2.4"):
    """layers: [(layer_name, [(weight_name, ndarray), ...]), ...] in `model.layers` order."""
    h = _load()
    if h is None:
        raise H5Error("libhdf5 is not available")
    f = _chk(h.H5Fcreate(os.fsencode(path), 2, 0, 0), "H5Fcreate %s" % path)       # H5F_ACC_TRUNC
    try:
        _write_str_attr(h, f, "layer_names", [ln for ln, _ in layers])
        _write_str_attr(h, f, "backend", [backend], scalar=True)
        _write_str_attr(h, f, "keras_version", [keras_version], scalar=True)
        for lname, weights in layers:
            g = _chk(h.H5Gcreate2(f, lname.encode(), 0, 0, 0), "H5Gcreate2 " + lname)
            _write_str_attr(h, g, "weight_names", [wn for wn, _ in weights])
            made = set()
            for wname, arr in weights:
                arr = np.ascontiguousarray(arr, np.float32)
                parts = wname.split("/")
                for d in range(1, len(parts)):                                 # nested groups for 'layer/W1:0'
                    sub = "/".join(parts[:d])
                    if sub not in made:
                        sg = _chk(h.H5Gcreate2(g, sub.encode(), 0, 0, 0), "H5Gcreate2 " + sub)
                        h.H5Gclose(sg)
                        made.add(sub)
                dims = (hsize_t * max(arr.ndim, 1))(*arr.shape)
                space = _chk(h.H5Screate_simple(arr.ndim, dims, None), "H5Screate_simple")
                d = _chk(h.H5Dcreate2(g, wname.encode(), h.T_F32LE, space, 0, 0, 0), "H5Dcreate2 " + wname)
                _chk(h.H5Dwrite(d, h.T_FLOAT, 0, 0, 0, arr.ctypes.data_as(_P)), "H5Dwrite " + wname)
                h.H5Dclose(d); h.H5Sclose(space)
            h.H5Gclose(g)
    finally:
        h.H5Fclose(f)


def load_keras_weights(path):
    """-> [(layer_name, [(weight_name, float32 ndarray), ...]), ...] for the layers that own weights, file order."""
    h = _load()
    if h is None:
        raise H5Error("libhdf5 is not available: cannot read %s" % path)
    f = _chk(h.H5Fopen(os.fsencode(path), 0, 0), "H5Fopen %s" % path)                # H5F_ACC_RDONLY
    try:
        root = f
        if h.H5Aexists(f, b"layer_names") <= 0:                                      # full-model file: weights live under /model_weights
            root = _chk(h.H5Gopen2(f, b"model_weights", 0), "open /model_weights (no layer_names attribute)")
        out = []
        for lname in _read_str_attr(h, root, "layer_names"):
            g = _chk(h.H5Gopen2(root, lname, 0), "H5Gopen2 %r" % lname)
            weights = []
            for wname in _read_str_attr(h, g, "weight_names"):
                d = _chk(h.H5Dopen2(g, wname, 0), "H5Dopen2 %r" % wname)
                space = h.H5Dget_space(d)
                nd = h.H5Sget_simple_extent_ndims(space)
                dims = (hsize_t * max(nd, 1))()
                h.H5Sget_simple_extent_dims(space, dims, None)
                arr = np.empty(tuple(int(dims[i]) for i in range(nd)), np.float32)
                _chk(h.H5Dread(d, h.T_FLOAT, 0, 0, 0, arr.ctypes.data_as(_P)), "H5Dread %r" % wname)
                h.H5Sclose(space); h.H5Dclose(d)
                weights.append((wname.decode("utf8"), arr))
            h.H5Gclose(g)
            if weights:
                out.append((lname.decode("utf8"), weights))
        if root != f:
            h.H5Gclose(root)
        return out
    finally:
        h.H5Fclose(f)
