"""ctypes binding of libv2xsim.so (csrc/v2xsim.c): the array arithmetic of a batched simulator step in C on a pool of threads.

The numpy expressions of rl/batched_env.py stay the definition (and the fallback when the library is not built, or with
V2X_SIM_NATIVE=0); the library evaluates the same formulas in the same order, the environments spread over
the pool -- which numpy cannot do (its Python glue serialises on the GIL: rl/batched_env.py, `workers`).
Random streams stay in Python; the library is handed the uniforms of a step."""
import ctypes as C
import os

import numpy as np

_lib = None
_tried = False


def _load():
    global _lib, _tried
    if _tried:
        return _lib
    _tried = True
    if os.environ.get("V2X_SIM_NATIVE", "1") == "0":
        return None
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.environ.get("V2XSIM_LIB", os.path.join(here, "libv2xsim.so"))
    if not os.path.exists(path):
        return None
    lib = C.CDLL(path)
    if lib.v2xsim_abi() != 3:
        return None
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int64)
    lib.v2xsim_channels.argtypes = [C.c_int, C.c_int, C.c_int, dp, C.c_int, dp, dp, dp, dp, dp, dp, dp, dp, dp]
    lib.v2xsim_reward.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip, dp, dp, dp] + [C.c_double] * 7 + [dp] * 5
    lib.v2xsim_interference.argtypes = [C.c_int, C.c_int, C.c_int, ip, dp] + [C.c_double] * 4 + [dp]
    lib.v2xsim_observe.argtypes = [C.c_int, C.c_int, C.c_int, ip, dp, dp, C.c_double, dp, dp]
    lib.v2xsim_set_threads.argtypes = [C.c_int]
    lib.v2xsim_mt_uniforms.argtypes = [C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_int32), dp, C.c_int]
    u32p, i32p = C.POINTER(C.c_uint32), C.POINTER(C.c_int32)
    lib.v2xsim_reset_vehicles.argtypes = [C.c_int, C.c_int, u32p, i32p, C.c_int, dp, dp, dp, dp, C.c_int, C.c_int, dp,
                                          C.POINTER(C.c_int8), dp]
    lib.v2xsim_sample_dest.argtypes = [C.c_int, C.c_int, C.c_int, u32p, i32p, ip, ip]
    u8p, f32p, i8p = C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_int8)
    lib.v2xsim_observe_packed.argtypes = [C.c_int, C.c_int, C.c_int, ip, dp, dp, C.c_double, dp, dp, f32p, i32p, i32p, u8p]
    lib.v2xsim_positions.argtypes = [C.c_int, C.c_int, u32p, i32p, dp, i8p, dp, C.c_double, C.c_int, dp, dp, dp, dp, C.c_double, C.c_double]
    lib.v2xsim_advance.argtypes = [C.POINTER(AdvanceArgs)]
    lib.v2xsim_advance_start.argtypes = [C.POINTER(AdvanceArgs)]
    lib.v2xsim_advance_start.restype = C.c_int
    lib.v2xsim_advance_wait.argtypes = [C.c_int]
    lib.v2xsim_advance_wait.restype = C.c_int
    lib.v2xsim_np_choice_noreplace.argtypes = [u32p, i32p, C.c_int64, C.c_int64, i32p, u32p, ip]
    lib.v2xsim_np_choice_noreplace.restype = C.c_int
    lib.v2xsim_np_shuffle_skip.argtypes = [u32p, i32p, C.c_int64]
    lib.v2xsim_np_shuffle_skip.restype = C.c_int
    lib.v2xsim_rollout.argtypes = [C.POINTER(RolloutArgs)]
    lib.v2xsim_rollout.restype = C.c_int
    lib.v2xsim_np_policy_draws.argtypes = [u32p, i32p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double,
                                           C.c_int64, ip, u8p, dp]
    lib.v2xsim_np_policy_draws.restype = C.c_int
    for f in (lib.v2xsim_observe_packed, lib.v2xsim_positions, lib.v2xsim_advance):
        f.restype = None
    for f in (lib.v2xsim_channels, lib.v2xsim_reward, lib.v2xsim_interference, lib.v2xsim_observe, lib.v2xsim_set_threads,
              lib.v2xsim_mt_uniforms, lib.v2xsim_reset_vehicles, lib.v2xsim_sample_dest):
        f.restype = None
    _lib = lib
    import atexit
    atexit.register(lib.v2xsim_advance_wait, 0)                   # a look-ahead job still running must not outlive its arrays
    return lib


class AdvanceArgs(C.Structure):
    """v2xsim_advance_args of csrc/v2xsim.c (same order)"""
    _fields_ = ([("E", C.c_int32), ("n", C.c_int32), ("rb", C.c_int32), ("n_lanes", C.c_int32),
                 ("timestep", C.c_double), ("width", C.c_double), ("height", C.c_double)]
                + [(k, C.c_void_p) for k in ("up", "down", "left", "right", "vel", "dest")]
                + [(k, C.c_double) for k in ("p_v2v", "p_v2i", "veh_gain", "veh_nf", "sig2")]
                + [(k, C.c_void_p) for k in ("keys_in", "mtpos_in", "xy_in", "dirs_in", "v2i_shadow_in", "v2v_shadow_in",
                                             "keys", "mtpos", "xy", "dirs", "v2i_shadow", "v2v_shadow",
                                             "v2v_abs", "v2i_abs", "v2v_ff", "v2i_ff", "interf_db", "state", "adj",
                                             "xe", "mask", "col", "regular", "scratch")])


class RolloutArgs(C.Structure):
    """v2xsim_rollout_args of include/v2xsim.h (same order)"""
    _fields_ = ([(k, C.c_int32) for k in ("n", "rb", "n_lanes", "T", "n_actions", "batch_predict")]
                + [(k, C.c_double) for k in ("timestep", "width", "height")]
                + [(k, C.c_void_p) for k in ("up", "down", "left", "right", "vel", "dest")]
                + [(k, C.c_double) for k in ("p_v2v", "p_v2i", "veh_gain", "veh_nf", "sig2", "bs_gain", "bs_nf")]
                + [(k, C.c_void_p) for k in ("keys", "mtpos", "xy", "dirs", "v2i_shadow", "v2v_shadow", "v2v_abs", "v2i_abs", "v2v_ff",
                                             "v2i_ff", "interf_db", "state", "adj", "xe", "mask", "col", "regular",
                                             "interference", "v2i_interf", "v2v_interf", "np_key", "np_pos")]
                + [(k, C.c_double) for k in ("eps_max", "eps_min", "eps_per_step", "eps_steps")]
                + [("step_no0", C.c_int64)]
                + [(k, C.c_void_p) for k in ("predict", "predict_ctx", "xe_pin", "col_pin", "q_pin", "t_xe", "t_xe_next", "t_col", "t_mask",
                                             "t_regular", "t_action", "t_v2v_rate", "t_v2i_rate", "n_greedy")]
                + [("eps_last", C.c_double)])


def available():
    return _load() is not None


def rollout(args):
    """v2xsim_rollout (include/v2xsim.h): T sequential transitions of ONE simulator in one call; -> its return code"""
    return int(_load().v2xsim_rollout(C.byref(args)))


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _c(a, dtype=np.float64):
    return np.ascontiguousarray(a, dtype=dtype)


def set_threads(n):
    _load().v2xsim_set_threads(int(n))


def channels(u, vel, pos, v2i_shadow, v2v_shadow, rb):
    """One channel update of E environments from their uniforms u[E, n_u]; returns the six arrays of
    BatchedEnviron._channels_of (new shadowing states first)."""
    lib = _load()
    E, n = vel.shape
    u, vel, pos = _c(u), _c(vel), _c(pos)
    v2i_shadow, v2v_shadow = _c(v2i_shadow).copy(), _c(v2v_shadow).copy()
    v2v_abs, v2i_abs = np.empty((E, n, n)), np.empty((E, n))
    v2v_ff, v2i_ff = np.empty((E, n, n, rb)), np.empty((E, n, rb))
    scratch = np.empty_like(u)
    lib.v2xsim_channels(E, n, rb, _d(u), u.shape[1], _d(vel), _d(pos), _d(v2i_shadow), _d(v2v_shadow), _d(v2v_abs), _d(v2i_abs),
                        _d(v2v_ff), _d(v2i_ff), _d(scratch))
    return v2i_shadow, v2v_shadow, v2v_abs, v2i_abs, v2v_ff, v2i_ff


def reward(ch, dest, v2v_ff, v2i_ff, v2i_abs, p_v2v, p_v2i, veh_gain, bs_gain, bs_nf, veh_nf, sig2):
    lib = _load()
    E, n, _, rb = v2v_ff.shape
    m = min(rb, n)
    ch, dest = _c(ch, np.int64), _c(dest, np.int64)
    v2v_ff, v2i_ff, v2i_abs = _c(v2v_ff), _c(v2i_ff), _c(v2i_abs)
    v2v_rate, v2i_rate = np.empty((E, n, 1)), np.empty((E, m))
    interference, v2i_interf, v2v_interf = np.empty((E, rb)), np.empty((E, rb)), np.empty((E, n, 1))
    lib.v2xsim_reward(E, n, rb, _i(ch), _i(dest), _d(v2v_ff), _d(v2i_ff), _d(v2i_abs), p_v2v, p_v2i, veh_gain, bs_gain, bs_nf,
                      veh_nf, sig2, _d(v2v_rate), _d(v2i_rate), _d(interference), _d(v2i_interf), _d(v2v_interf))
    return v2v_rate, v2i_rate, interference, v2i_interf, v2v_interf


def interference_db(dest, v2v_ff, p_v2i, veh_gain, veh_nf, sig2):
    lib = _load()
    E, n, _, rb = v2v_ff.shape
    dest, v2v_ff = _c(dest, np.int64), _c(v2v_ff)
    out = np.empty((E, n, 1, rb))
    lib.v2xsim_interference(E, n, rb, _i(dest), _d(v2v_ff), p_v2i, veh_gain, veh_nf, sig2, _d(out))
    return out


def observe(dest, v2v_ff, v2i_ff, power, n_channels):
    lib = _load()
    E, n, _, rb = v2v_ff.shape
    dest, v2v_ff, v2i_ff = _c(dest, np.int64), _c(v2v_ff), _c(v2i_ff)
    state, adj = np.empty((E, n, 3 * n_channels + 1)), np.empty((E, n, n))
    lib.v2xsim_observe(E, n, n_channels, _i(dest), _d(v2v_ff), _d(v2i_ff), float(power), _d(state), _d(adj))
    return state, adj


def mt_uniforms(keys, pos, n_u):
    """The next n_u random.random() values of every stream: keys [E, 624] uint32 and pos [E] int32 are the MT19937 states
    (numpy RandomState layout) and are advanced in place."""
    lib = _load()
    E = keys.shape[0]
    assert keys.dtype == np.uint32 and keys.flags.c_contiguous and pos.dtype == np.int32 and pos.flags.c_contiguous
    out = np.empty((E, n_u))
    lib.v2xsim_mt_uniforms(E, keys.ctypes.data_as(C.POINTER(C.c_uint32)), pos.ctypes.data_as(C.POINTER(C.c_int32)), _d(out), n_u)
    return out


def _mt(keys, pos):
    assert keys.dtype == np.uint32 and keys.flags.c_contiguous and pos.dtype == np.int32 and pos.flags.c_contiguous
    return keys.ctypes.data_as(C.POINTER(C.c_uint32)), pos.ctypes.data_as(C.POINTER(C.c_int32))


def reset_vehicles(keys, pos, n, lanes, width, height):
    """The integer draws of add_new_vehicles_by_number for every stream (keys / pos: see mt_uniforms), in the stdlib
    generator's order; lanes = (down, up, left, right).  -> positions [E, n, 2], directions [E, n] int8, velocities [E, n]."""
    lib = _load()
    E = keys.shape[0]
    tabs = [_c(np.asarray(t, np.float64)) for t in lanes]
    xy, dirs, vel = np.empty((E, n, 2)), np.empty((E, n), np.int8), np.empty((E, n))
    k, p = _mt(keys, pos)
    lib.v2xsim_reset_vehicles(E, n, k, p, len(tabs[0]), _d(tabs[0]), _d(tabs[1]), _d(tabs[2]), _d(tabs[3]), int(width), int(height),
                              _d(xy), dirs.ctypes.data_as(C.POINTER(C.c_int8)), _d(vel))
    return xy, dirs, vel


def sample_dest(keys, pos, cand):
    """random.sample(cand[e][i], 1)[0] for every environment e and link i (cand [E, n, m], m <= 21), in link order."""
    lib = _load()
    cand = _c(cand, np.int64)
    E, n, m = cand.shape
    if m > 21 or m < 1:
        raise ValueError("sample_dest: populations of 1..21 candidates (CPython's pool method)")
    dest = np.empty((E, n), np.int64)
    k, p = _mt(keys, pos)
    lib.v2xsim_sample_dest(E, n, m, k, p, _i(cand), _i(dest))
    return dest


def observe_packed(dest, v2v_ff, v2i_ff, power, n_channels):
    """observe() plus the same observation in the engine's packed form: xe [E, n, 16] float32, source masks [E, n] int32, CSR
    sources [E, n (n-2)] int32 (zeros for an irregular graph) and the regular flags [E]."""
    lib = _load()
    E, n, _, rb = v2v_ff.shape
    dest, v2v_ff, v2i_ff = _c(dest, np.int64), _c(v2v_ff), _c(v2i_ff)
    state, adj = np.empty((E, n, 3 * n_channels + 1)), np.empty((E, n, n))
    xe, mask = np.empty((E, n, 16), np.float32), np.empty((E, n), np.int32)
    col, regular = np.empty((E, max(n * (n - 2), 1)), np.int32), np.empty(E, np.uint8)
    lib.v2xsim_observe_packed(E, n, n_channels, _i(dest), _d(v2v_ff), _d(v2i_ff), float(power), _d(state), _d(adj),
                              xe.ctypes.data_as(C.POINTER(C.c_float)), mask.ctypes.data_as(C.POINTER(C.c_int32)),
                              col.ctypes.data_as(C.POINTER(C.c_int32)), regular.ctypes.data_as(C.POINTER(C.c_uint8)))
    return state, adj, xe, mask, col, regular.astype(bool)


def positions(keys, pos, xy, dirs, vel, timestep, lanes, width, height):
    """renew_positions of every environment IN PLACE (xy [E, n, 2], dirs [E, n] int8; turn draws from the streams keys / pos);
    lanes = (up, down, left, right) tables."""
    lib = _load()
    E, n = dirs.shape
    assert xy.dtype == np.float64 and xy.flags.c_contiguous and dirs.dtype == np.int8 and dirs.flags.c_contiguous
    tabs = [_c(np.asarray(t, np.float64)) for t in lanes]
    k, p = _mt(keys, pos)
    vel = _c(vel)
    lib.v2xsim_positions(E, n, k, p, _d(xy), dirs.ctypes.data_as(C.POINTER(C.c_int8)), _d(vel), float(timestep), len(tabs[0]),
                         _d(tabs[0]), _d(tabs[1]), _d(tabs[2]), _d(tabs[3]), float(width), float(height))


def advance(args):
    _load().v2xsim_advance(C.byref(args))


def advance_start(args):
    """-> the job's ticket (> 0) when the pool took the job, 0 when it is still busy with another simulator's"""
    return max(0, _load().v2xsim_advance_start(C.byref(args)))


def advance_wait(ticket=0):
    """returns when job `ticket` is done (0: whatever is in flight)"""
    return _load().v2xsim_advance_wait(int(ticket)) == 0


def np_policy_draws(E, n, n_actions, eps_max, eps_min, eps_per_step, eps_steps, step_no0):
    """The epsilon-greedy draws of one iteration over E simulators on np.random's process-wide generator (v2xsim_np_policy_draws):
    -> (actions [E, n, 1] int64 -- rows of greedy simulators zero --, indices of the greedy simulators, the last epsilon).  Same
    draws, same values, same generator state afterwards as the loop of np.random.random() / np.random.randint(0, n_actions, (n, 1))."""
    lib = _load()
    name, key, pos, has_gauss, cached = np.random.get_state()
    if name != 'MT19937':
        raise RuntimeError("np.random is not on MT19937")
    key = np.ascontiguousarray(key, np.uint32).copy()
    p = np.array([pos], np.int32)
    actions, greedy, eps = np.zeros((E, n, 1), np.int64), np.zeros(E, np.uint8), np.zeros(1)
    rc = lib.v2xsim_np_policy_draws(key.ctypes.data_as(C.POINTER(C.c_uint32)), p.ctypes.data_as(C.POINTER(C.c_int32)), int(E), int(n),
                                    int(n_actions), float(eps_max), float(eps_min), float(eps_per_step), float(eps_steps), int(step_no0),
                                    actions.ctypes.data_as(C.POINTER(C.c_int64)), greedy.ctypes.data_as(C.POINTER(C.c_uint8)), _d(eps))
    if rc < 0:
        raise ValueError("np_policy_draws: bad argument")
    np.random.set_state((name, key, int(p[0]), has_gauss, cached))
    return actions, np.nonzero(greedy)[0], float(eps[0])


_choice_scratch = {}


def np_choice_noreplace(n, k):
    """np.random.choice(n, k, replace=False) on the process-wide legacy generator -- the same draws, the same indices, the same
    state afterwards -- with the permutation of the n positions done in the library (numpy: 15-28 ms at n = 1e6, here 3-5)."""
    lib = _load()
    name, key, pos, has_gauss, cached = np.random.get_state()
    if name != 'MT19937':
        raise RuntimeError("np.random is not on MT19937")
    key = np.ascontiguousarray(key, np.uint32).copy()
    p = np.array([pos], np.int32)
    buf = _choice_scratch.get("buf")
    if buf is None or buf[0].size < n:
        m = max(int(n), 2 * (buf[0].size if buf is not None else 0), 65536)
        buf = _choice_scratch["buf"] = (np.empty(m, np.int32), np.empty(m, np.uint32))
    out = np.empty(int(k), np.int64)
    rc = lib.v2xsim_np_choice_noreplace(key.ctypes.data_as(C.POINTER(C.c_uint32)), p.ctypes.data_as(C.POINTER(C.c_int32)), int(n), int(k),
                                        buf[0].ctypes.data_as(C.POINTER(C.c_int32)), buf[1].ctypes.data_as(C.POINTER(C.c_uint32)),
                                        out.ctypes.data_as(C.POINTER(C.c_int64)))
    if rc != 0:
        raise ValueError("np_choice_noreplace: need 1 <= k <= n < 2**31")
    np.random.set_state((name, key, int(p[0]), has_gauss, cached))
    return out


class ChoiceAhead(object):
    """np_choice_noreplace(n, k) running on a helper thread (the library call releases the GIL): started right after the caller's
    last draw from np.random, finished -- result(): the indices, and np.random moved on past the draws -- before its next one.
    Between the two the process-wide generator must not be used; the caller (Agent._packed_iteration inside Agent.train) knows
    that nothing draws there -- and result() CHECKS it: the generator's state (position, a CRC of the key, the cached gauss
    value) is compared with the snapshot the helper started from.  If anything drew from np.random in between (a callback, a
    logging hook, a later edit of the rollout), the helper's work is thrown away and the draw is redone synchronously from the
    CURRENT state -- exactly what the reference's `np.random.choice` at this point would have drawn (BS_brain.py:258-270);
    with V2X_RL_STRICT_RNG=1 (the tests) it raises instead.  `fallbacks` counts the redone draws of the process."""
    _pool = None
    fallbacks = 0

    @staticmethod
    def _fingerprint(state):
        import zlib
        name, key, pos, has_gauss, cached = state
        return (name, int(pos), zlib.crc32(np.ascontiguousarray(key, np.uint32).tobytes()), int(has_gauss), float(cached))

    def __init__(self, n, k):
        lib = _load()
        name, key, pos, has_gauss, cached = state = np.random.get_state()
        if name != 'MT19937':
            raise RuntimeError("np.random is not on MT19937")
        self._snapshot = self._fingerprint(state)
        self._n, self._k = int(n), int(k)
        self._rest = (name, has_gauss, cached)
        self._key = np.ascontiguousarray(key, np.uint32).copy()
        self._pos = np.array([pos], np.int32)
        self._buf = (np.empty(int(n), np.int32), np.empty(int(n), np.uint32))      # (not the shared scratch: another thread)
        self._out = np.empty(int(k), np.int64)
        if ChoiceAhead._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            ChoiceAhead._pool = ThreadPoolExecutor(max_workers=1)
        u32p, i32p = C.POINTER(C.c_uint32), C.POINTER(C.c_int32)
        self._fut = ChoiceAhead._pool.submit(lib.v2xsim_np_choice_noreplace, self._key.ctypes.data_as(u32p), self._pos.ctypes.data_as(i32p),
                                             int(n), int(k), self._buf[0].ctypes.data_as(i32p), self._buf[1].ctypes.data_as(u32p),
                                             self._out.ctypes.data_as(C.POINTER(C.c_int64)))

    def result(self):
        if self._fut.result() != 0:
            raise ValueError("np_choice_noreplace: need 1 <= k <= n < 2**31")
        if self._fingerprint(np.random.get_state()) != self._snapshot:
            if os.environ.get("V2X_RL_STRICT_RNG") == "1":
                raise RuntimeError("np.random was used between ChoiceAhead's start and result(): the helper's draws would be "
                                   "replayed over them")
            ChoiceAhead.fallbacks += 1
            return np_choice_noreplace(self._n, self._k)
        name, has_gauss, cached = self._rest
        np.random.set_state((name, self._key, int(self._pos[0]), has_gauss, cached))
        return self._out
