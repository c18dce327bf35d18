"""E independent V2X simulators stepped as arrays ("vectorised/batched Environ").

`BatchedEnviron` holds the state of E environments of N vehicles as stacked numpy arrays and advances all of them with
one set of array operations per simulator step: mobility (Environment.py:236-345), large-scale fading with correlated
shadowing (:378-393), Rayleigh fast fading (:395-406), the rate / interference computation (:408-493) and the agent's
observation (BS_brain.py:389-467).  Every environment owns an `MTStream` -- an MT19937 stream with the stdlib's draw
algorithms -- and consumes it in the reference's order, so environment e of a batch IS the single simulator seeded with
`seeds[e]`: same vehicles, bit-identical positions and directions, channels / rates to rounding
(tests/test_rl_batched_env.py checks both against `Environ` and against trajectories captured from the reference).
With `streams=None, n_envs=1` the one environment runs on the process-wide stdlib generator (borrowed per call), i.e.
it is a drop-in for `Environ` in seeded runs.

Only the 1-receiver configuration of the reference (n_Neighbor = 1) is batched.
"""
import os

import numpy as np

from . import native_sim
from .environment import Environ
from .mtstream import MTStream, gauss_uniforms, box_muller

_DIRS = 'udlr'        # direction codes 0..3


def _usable_cpus():
    """CPUs this process can keep busy: its affinity mask, cut to the cgroup's CPU quota when one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


class BatchedEnviron(object):
    def __init__(self, down_lane, up_lane, left_lane, right_lane, width, height, n_envs=1, seeds=None, workers=None,
                 native=None, lookahead=False):
        """seeds: one seed per environment (each environment then reproduces `random.seed(seed); Environ(...)`);
        None with n_envs == 1: the process-wide stdlib generator is used (borrowed and returned around every call).
        workers: threads that share the channel update of a step (3,360 Gaussian draws + their transcendentals per
        20-link environment and step: 3/4 of the simulator's time; numpy releases the GIL inside them).  None: one per 25
        environments, at most 4 -- measured on the 256-thread host of the MI355X box, 20 links: 100 environments 109 ->
        65 us per environment step with 4 threads, 78 with 8, 115 with 16 (the Python glue between the array
        operations serialises on the GIL); 10 environments are fastest on one thread.  Environments are independent,
        so the result does not depend on the thread count.
        native: evaluate the array arithmetic of a step (channel update, rates, interference, observation) in
        libv2xsim.so (csrc/v2xsim.c: the same formulas in C, the environments spread over a pool of threads -- real threads, no GIL).  None:
        whenever the library is built and the environments own their streams (V2X_SIM_NATIVE=0 switches it off); the numpy
        code below stays the definition, the two agree to the last bits of libm (tests: 1e-12).
        lookahead: compute the NEXT simulator step on a worker thread of the library while the caller is busy with the current
        observation (nothing in a step depends on the actions except the rates paid for them, see act()); the result is taken
        when act() is called next and dropped when anything else touches the simulator first, so every trajectory is the one
        without it.  Needs the native library and own streams; at most one simulator of a process looks ahead at a time."""
        self._proto = Environ.__new__(Environ)                 # constants + path-loss models of the single simulator
        p = self._proto
        p.timestep = 0.01
        p.down_lanes, p.up_lanes, p.left_lanes, p.right_lanes = list(down_lane), list(up_lane), list(left_lane), list(right_lane)
        p.width, p.height = width, height
        p.V2V_power_dB = p.V2I_power_dB = 23
        p.V2V_power_dB_List = [23, 10, 5]
        p.fixed_v2v_power_index = 1
        p.sig2_dB = -114
        p.bsAntGain, p.bsNoiseFigure, p.vehAntGain, p.vehNoiseFigure = 8, 5, 3, 9
        p.sig2 = 10 ** (p.sig2_dB / 10)
        for k in ('timestep', 'width', 'height', 'V2V_power_dB', 'V2I_power_dB', 'V2V_power_dB_List', 'fixed_v2v_power_index',
                  'sig2', 'bsAntGain', 'bsNoiseFigure', 'vehAntGain', 'vehNoiseFigure'):
            setattr(self, k, getattr(p, k))
        self.n_RB, self.n_Veh, self.n_Neighbor = 4, 4, 1
        self.E = int(n_envs)
        if seeds is None and self.E != 1:
            raise ValueError("n_envs > 1 needs one seed per environment")
        self._shared = seeds is None
        if workers is None:
            workers = min(4, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), max(1, self.E // 25))
        self.workers = 1 if self._shared else max(1, min(int(workers), self.E))
        self.native = (native_sim.available() and not self._shared) if native is None else bool(native)
        if self.native and not native_sim.available():
            raise RuntimeError("native=True but libv2xsim.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
        self._mt_keys = self._mt_pos = None
        if self.native:
            # threads of the library's pool: at most 16, and never more than the CPU time the process may use (cgroup cpu.max:
            # 16 CPUs on the MI355X boxes, whatever the 256 hardware threads suggest) less one for this thread.  The pool
            # sleeps between jobs; the OpenMP teams of rounds 3-4 kept spinning, ran the process into its CPU quota and got it
            # parked for 25-50 ms a few times per hundred steps (csrc/v2xsim.c).  V2X_SIM_THREADS overrides.
            cap = int(os.environ.get("V2X_SIM_THREADS", "0")) or min(16, max(1, _usable_cpus() - 1))
            # (ONE simulator: its steps are cut over a team of threads inside v2xsim_rollout -- the caller, a stream thread and
            #  up to 10 workers -- instead of environments over the pool)
            native_sim.set_threads(max(1, min(self.E if self.E > 1 else 12, cap)))
            if not self._shared:                               # the streams' MT19937 states, where the library can advance them
                self._mt_keys = np.empty((self.E, 624), np.uint32)
                self._mt_pos = np.zeros(self.E, np.int32)
        self._pool = None
        self.lookahead = bool(lookahead)
        self._ahead = None                 # a started v2xsim_advance job: its output arrays (inputs stay untouched until taken)
        self._obs = None                   # observation of the CURRENT channels: (v2v, v2i, state, adj, xe, mask, col, regular)
        self._scratch = None
        self._job_static = None
        self._step_pending = False         # act_deferred(): the rates are out, the simulator step is applied at the next call
        self.streams = None if self._shared else [MTStream(int(s)) for s in seeds]
        if not self._shared and len(self.streams) != self.E:
            raise ValueError("need %d seeds" % self.E)
        if self._mt_keys is not None:
            for e, st in enumerate(self.streams):
                st.attach(self._mt_keys[e], self._mt_pos, e)
                st.on_touch = self._drop_lookahead             # any draw outside the library invalidates a step computed ahead
        # lane tables of the mobility rule, in the reference's checking order (Environment.py:247-324):
        # direction -> (moving axis, sign, [(lanes, new direction, side-step sign, gap sign)])
        L, R, U, D = p.left_lanes, p.right_lanes, p.up_lanes, p.down_lanes
        self._moves = {0: (1, +1, [(L, 2, -1, -1), (R, 3, +1, +1)]), 1: (1, -1, [(L, 2, -1, -1), (R, 3, +1, +1)]),
                       3: (0, +1, [(U, 0, +1, -1), (D, 1, -1, -1)]), 2: (0, -1, [(U, 0, +1, -1), (D, 1, -1, -1)])}
        self._lanes_of = {0: np.array(L + R, float), 1: np.array(L + R, float), 3: np.array(U + D, float), 2: np.array(U + D, float)}
        with self._rng() as rs:                                # Environ.__init__ draws the initial shadowing (:208-209)
            for s in rs:
                s.gauss_array((self.n_Veh, self.n_Veh), Environ.V2V_SHADOW_STD)
                s.gauss_array((self.n_Veh,), Environ.V2I_SHADOW_STD)

    # ------------------------------------------------------------------ random streams
    class _Borrow(object):
        def __init__(self, env):
            self.env = env

        def __enter__(self):
            self.env._drop_lookahead()                         # the streams are about to be drawn from
            if self.env._shared:
                self.s = [MTStream.borrow_stdlib()]
                return self.s
            if self.env._mt_keys is not None:                  # streams whose state lives in the library's arrays: one
                self.live = []                                 # pull at a stream's first draw of this block, one push at the end
                for st in self.env.streams:
                    st.session = self.live
            return self.env.streams

        def __exit__(self, *exc):
            if self.env._shared:
                self.s[0].return_stdlib()
            elif self.env._mt_keys is not None:
                for st in self.live:
                    st._push(end_of_session=True)
                for st in self.env.streams:
                    st.session = None

    def _rng(self):
        return BatchedEnviron._Borrow(self)

    # ------------------------------------------------------------------ construction
    def new_random_game(self, n_Veh=0):
        """Environment.py:495-506 for every environment."""
        self._drop_lookahead()
        self._obs = None
        if n_Veh > 0:
            if n_Veh % 4:
                raise ValueError("n_Veh must be a multiple of 4; got %d" % n_Veh)
            self.n_Veh = n_Veh
        E, N, p = self.E, self.n_Veh, self._proto
        self.pos = np.zeros((E, N, 2))
        self.dirs = np.zeros((E, N), np.int8)
        self.vel = np.zeros((E, N))
        self.dest = np.zeros((E, N), np.int64)
        self._v2v_shadow = np.zeros((E, N, N))
        self._v2i_shadow = np.zeros((E, N))
        lanes = (p.down_lanes, p.up_lanes, p.left_lanes, p.right_lanes)
        in_c = self.native and self._mt_keys is not None
        import random as _stdlib_random
        fast = _stdlib_random.Random()
        if in_c:
            # the scalar integer draws of a reset (5 per vehicle) on the streams' states where the library holds them: one
            # call for all environments (the round trip stream -> stdlib generator -> stream below costs two 625-word state
            # conversions each way per environment: 8 of a reset's 15 ms with 50 environments)
            self.pos[:], self.dirs[:], self.vel[:] = native_sim.reset_vehicles(self._mt_keys, self._mt_pos, N, lanes, p.width, p.height)
        if in_c and not any(st.gauss_next is not None for st in self.streams):
            # ... and the reset's four Gaussian arrays of every environment out of ONE bulk draw: their counts are even, so the
            # 2 (N^2 + N) gauss() values are consecutive (cos, sin) pairs of consecutive uniforms, exactly as four
            # gauss_array calls per stream would pair them (V2V_Shadowing / V2I_Shadowing of the vehicles are drawn and never
            # used, :232-233; then the environment's V2V / V2I shadowing)
            n_g = 2 * (N * N + N)
            u = native_sim.mt_uniforms(self._mt_keys, self._mt_pos, n_g)
            x2pi = u[:, 0::2] * (2.0 * np.pi)
            g2rad = np.sqrt(-2.0 * np.log(1.0 - u[:, 1::2]))
            zg = np.stack([np.cos(x2pi) * g2rad, np.sin(x2pi) * g2rad], axis=2).reshape(E, n_g)
            o = N * N + N
            self._v2v_shadow = zg[:, o:o + N * N].reshape(E, N, N) * Environ.V2V_SHADOW_STD
            self._v2i_shadow = zg[:, o + N * N:] * Environ.V2I_SHADOW_STD
            gauss_done = True
        else:
            gauss_done = False
        with self._rng() as rs:
            for e, s in enumerate(rs):
                if gauss_done:
                    break
                if not in_c:
                    # ... at the stdlib generator's C speed: MTStream IS that generator draw for draw, so its state is lent
                    # out and taken back (one state copy each way per environment instead of a numpy round trip per draw)
                    fast.setstate(s._export())
                    k = 0
                    for _ in range(N // 4):                        # add_new_vehicles_by_number (:217-234)
                        ind = fast.randrange(0, len(p.down_lanes))
                        for code, lane_x, lane_y in ((1, p.down_lanes[ind], None), (0, p.up_lanes[ind], None),
                                                     (2, None, p.left_lanes[ind]), (3, None, p.right_lanes[ind])):
                            if lane_y is None:
                                self.pos[e, k] = (lane_x, fast.randint(0, p.height))
                            else:
                                self.pos[e, k] = (fast.randint(0, p.width), lane_y)
                            self.dirs[e, k] = code
                            self.vel[e, k] = fast.randint(10, 15)
                            k += 1
                    s._import(fast.getstate())
                s.gauss_array((N, N), 3)                        # V2V_Shadowing / V2I_Shadowing: drawn, never used (:232-233)
                s.gauss_array((N,), 8)
                self._v2v_shadow[e] = s.gauss_array((N, N), Environ.V2V_SHADOW_STD)
                self._v2i_shadow[e] = s.gauss_array((N,), Environ.V2I_SHADOW_STD)
        self.renew_channels_fastfading()
        # renew_neighbor (:360-376): link i's receiver is one of its N - 3 nearest other vehicles, random.sample(..., 1)
        z = self.pos[:, :, 0] + 1j * self.pos[:, :, 1]
        order = np.argsort(np.abs(z[:, :, None] - z[:, None, :]), axis=1)            # [e, rank, i]: vehicles by distance from i
        cand = np.ascontiguousarray(order[:, 1:N - 2, :].transpose(0, 2, 1))         # [e, i, N - 3]
        if in_c and 1 <= cand.shape[2] <= 21:
            self.dest[:] = native_sim.sample_dest(self._mt_keys, self._mt_pos, cand)
        else:
            with self._rng() as rs:
                for e, s in enumerate(rs):
                    ce = cand[e].tolist()                                            # (python ints: random.sample indexes with them)
                    fast.setstate(s._export())
                    for i in range(N):
                        self.dest[e, i] = fast.sample(ce[i], 1)[0]
                    s._import(fast.getstate())
        self.activate_links = np.ones((E, N, 1), dtype=bool)
        self._obs = None                                       # new receivers: the cached observation is of the old ones

    # ------------------------------------------------------------------ mobility
    def renew_positions(self):
        """One 10 ms step of every vehicle of every environment.  Vehicles that reach no crossing lane (almost all)
        move as arrays; the few that do are walked in the reference's order because each reached lane costs its
        environment one uniform draw (turn with probability 0.4)."""
        p = self._proto
        self._drop_lookahead()
        if self.native and self._mt_keys is not None:          # the same walk in C, on the streams' states (csrc/v2xsim.c)
            native_sim.positions(self._mt_keys, self._mt_pos, self.pos, self.dirs, self.vel, p.timestep,
                                 (p.up_lanes, p.down_lanes, p.left_lanes, p.right_lanes), p.width, p.height)
            return
        dd = self.vel * p.timestep
        axis = np.where(self.dirs < 2, 1, 0)
        sign = np.where((self.dirs == 0) | (self.dirs == 3), 1.0, -1.0)
        a = np.take_along_axis(self.pos, axis[..., None], axis=2)[..., 0]
        new_a = np.where(sign > 0, a + dd, a - dd)
        lo, hi = np.minimum(a, new_a), np.maximum(a, new_a)
        flagged = np.zeros(self.dirs.shape, bool)
        for d, lanes in self._lanes_of.items():
            m = self.dirs == d
            if m.any():
                flagged[m] = ((lanes[None, :] >= lo[m][:, None]) & (lanes[None, :] <= hi[m][:, None])).any(axis=1)
        straight = ~flagged
        e_idx, v_idx = np.nonzero(straight)
        self.pos[e_idx, v_idx, axis[straight]] = new_a[straight]
        if flagged.any():
            with self._rng() as rs:
                for e, v in zip(*np.nonzero(flagged)):          # row-major: vehicle order inside every environment
                    s = rs[e]
                    ax, sg, options = self._moves[int(self.dirs[e, v])]
                    av, ov, dv = self.pos[e, v, ax], self.pos[e, v, 1 - ax], dd[e, v]
                    turned = False
                    for lanes, new_dir, side_sign, gap_sign in options:
                        for lane in lanes:
                            reached = (av <= lane and av + dv >= lane) if sg > 0 else (av >= lane and av - dv <= lane)
                            if reached and s.uniform(0, 1) < 0.4:
                                gap = sg * (lane - av)
                                new_o = ov + side_sign * (dv + gap_sign * gap)
                                self.pos[e, v] = (new_o, lane) if ax == 1 else (lane, new_o)
                                self.dirs[e, v] = new_dir
                                turned = True
                                break
                        if turned:
                            break
                    if not turned:
                        self.pos[e, v, ax] = av + dv if sg > 0 else av - dv
        x, y = self.pos[..., 0], self.pos[..., 1]
        out = (x < 0) | (y < 0) | (x > p.width) | (y > p.height)
        if out.any():                                          # re-entry on the outermost lane (:326-345)
            d = self.dirs
            for code, new_dir, fix_axis, lane in ((0, 3, 1, p.right_lanes[-1]), (1, 2, 1, p.left_lanes[0]),
                                                  (2, 0, 0, p.up_lanes[0]), (3, 1, 0, p.down_lanes[-1])):
                m = out & (d == code)
                if m.any():
                    ee, vv = np.nonzero(m)
                    self.pos[ee, vv, fix_axis] = lane
                    self.dirs[ee, vv] = new_dir
                    out = out & ~m

    # ------------------------------------------------------------------ channels
    def renew_channels_fastfading(self):
        """renew_channel + fast fading (:378-406) for all environments; per environment ONE block of uniforms feeds the
        n + n^2 shadowing draws and the 2 n rb + 2 n^2 rb Rayleigh draws, in the reference's order."""
        E, n, rb = self.E, self.n_Veh, self.n_RB
        self._drop_lookahead()
        self._obs = None
        if (n + n * n + 2 * n * rb + 2 * n * n * rb) & 1:      # keep the odd value cached like random.gauss would
            raise NotImplementedError("odd number of draws per step")
        with self._rng() as rs:
            if self.native:
                n_draws = n + n * n + 2 * n * rb + 2 * n * n * rb
                if self._mt_keys is not None:                  # all streams advance in C, one thread per group of environments
                    if any(s.gauss_next is not None for s in rs):
                        raise RuntimeError("a stream holds a cached gauss value")
                    u = native_sim.mt_uniforms(self._mt_keys, self._mt_pos, n_draws)
                else:
                    u = gauss_uniforms(rs, n_draws)
                out = [native_sim.channels(u, self.vel, self.pos, self._v2i_shadow, self._v2v_shadow, rb)]
            elif self.workers <= 1 or E < 2:
                out = [self._channels_of(rs, 0, E)]
            else:
                if self._pool is None:
                    from concurrent.futures import ThreadPoolExecutor
                    self._pool = ThreadPoolExecutor(max_workers=self.workers)
                w = self.workers
                cuts = [E * i // w for i in range(w + 1)]
                out = list(self._pool.map(lambda i: self._channels_of(rs, cuts[i], cuts[i + 1]), range(w)))
        names = ('_v2i_shadow', '_v2v_shadow', 'V2V_channels_abs', 'V2I_channels_abs', 'V2V_channels_with_fastfading',
                 'V2I_channels_with_fastfading')
        for k, name in enumerate(names):
            setattr(self, name, out[0][k] if len(out) == 1 else np.concatenate([o[k] for o in out], axis=0))

    def _channels_of(self, rs, e0, e1):
        """The channel update of environments [e0, e1): pure function of their own state and streams."""
        n, rb, p = self.n_Veh, self.n_RB, self._proto
        E = e1 - e0
        n_sh, n_ff = n + n * n, 2 * n * rb + 2 * n * n * rb
        g = box_muller(gauss_uniforms(rs[e0:e1], n_sh + n_ff))[:, :n_sh + n_ff]
        dd = 0.002 * self.vel[e0:e1]
        v2i_shadow = (np.exp(-1 * (dd / Environ.V2I_DECORR)) * self._v2i_shadow[e0:e1]
                      + np.sqrt(1 - np.exp(-2 * (dd / Environ.V2I_DECORR))) * (g[:, :n] * Environ.V2I_SHADOW_STD))
        ddm = dd[:, :, None] + dd[:, None, :]
        v2v_shadow = (np.exp(-1 * (ddm / Environ.V2V_DECORR)) * self._v2v_shadow[e0:e1]
                      + np.sqrt(1 - np.exp(-2 * (ddm / Environ.V2V_DECORR))) * (g[:, n:n_sh].reshape(E, n, n) * Environ.V2V_SHADOW_STD))
        v2v_abs = p._v2v_pathloss(self.pos[e0:e1]) + v2v_shadow + 50 * np.identity(n)
        v2i_abs = p._v2i_pathloss(self.pos[e0:e1]) + v2i_shadow
        f = g[:, n_sh:]
        a, b = n * rb, n * n * rb
        re, im = f[:, :a].reshape(E, n, rb), f[:, a:2 * a].reshape(E, n, rb)
        v2i_ff = 20 * np.log10(np.abs(1 / np.sqrt(2) * (re + 1j * im)))
        re, im = f[:, 2 * a:2 * a + b].reshape(E, n, n, rb), f[:, 2 * a + b:].reshape(E, n, n, rb)
        v2v_ff = 20 * np.log10(np.abs(1 / np.sqrt(2) * (re + 1j * im)))
        return v2i_shadow, v2v_shadow, v2v_abs, v2i_abs, v2v_abs[..., None] - v2v_ff, v2i_abs[..., None] - v2i_ff

    # ------------------------------------------------------------------ reward
    def compute_reward_with_channel_selection(self, actions):
        """actions [E, N] or [E, N, 1] -> V2V rates [E, N, 1], V2I rates [E, min(rb, N)], interference at the base
        station [E, rb] (Environment.py:408-458, every link active, one receiver per link)."""
        self.finish_step()
        E, n, rb = self.E, self.n_Veh, self.n_RB
        ch = np.asarray(actions).reshape(E, n).astype(np.int64)
        if self.native:
            v2v_rate, v2i_rate, interference, self.V2I_Interference, self.V2V_Interference = native_sim.reward(
                ch, self.dest, self.V2V_channels_with_fastfading, self.V2I_channels_with_fastfading, self.V2I_channels_abs,
                self.V2V_power_dB_List[self.fixed_v2v_power_index], self.V2I_power_dB, self.vehAntGain, self.bsAntGain,
                self.bsNoiseFigure, self.vehNoiseFigure, self.sig2)
            return v2v_rate, v2i_rate, interference
        ei = np.arange(E)[:, None]
        ki = np.arange(n)[None, :]
        rx = self.dest
        p_v2v = self.V2V_power_dB_List[self.fixed_v2v_power_index]
        v2v, v2i = self.V2V_channels_with_fastfading, self.V2I_channels_with_fastfading
        onehot = ch[:, :, None] == np.arange(rb)[None, None, :]                      # [E, N, rb]
        at_bs = 10 ** ((p_v2v - v2i[ei, ki, ch] + self.vehAntGain + self.bsAntGain - self.bsNoiseFigure) / 10)
        interference = (at_bs[:, :, None] * onehot).sum(axis=1)
        self.V2I_Interference = interference + self.sig2
        gain = 2 * self.vehAntGain - self.vehNoiseFigure
        signal = 10 ** ((p_v2v - v2v[ei, ki, rx, ch] + gain) / 10)
        v2v_int = np.zeros((E, n))
        has_v2i = ch < n                                          # the V2I transmitter of that RB is vehicle number RB
        chc = np.minimum(ch, n - 1)
        v2v_int += np.where(has_v2i, 10 ** ((self.V2I_power_dB - v2v[ei, chc, rx, ch] + gain) / 10), 0.0)
        same = (ch[:, :, None] == ch[:, None, :]) & ~np.eye(n, dtype=bool)[None]       # [E, i, k]
        cross = 10 ** ((p_v2v - v2v[ei[:, :, None], np.arange(n)[None, None, :], rx[:, :, None], ch[:, :, None]] + gain) / 10)
        v2v_int += (cross * same).sum(axis=2)
        self.V2V_Interference = v2v_int[..., None] + self.sig2
        v2v_rate = np.log2(1 + np.divide(signal[..., None], self.V2V_Interference))
        m = min(rb, n)
        v2i_signals = self.V2I_power_dB - self.V2I_channels_abs[:, 0:m] + self.vehAntGain + self.bsAntGain - self.bsNoiseFigure
        v2i_rate = np.log2(1 + np.divide(10 ** (v2i_signals / 10), self.V2I_Interference[:, 0:m]))
        return v2v_rate, v2i_rate, interference

    def Compute_Interference(self, actions):
        """Environment.py:460-493 (observable part: noise + the co-channel V2I transmitter), [E, N, 1, rb] in dB."""
        self.finish_step()
        E, n, rb = self.E, self.n_Veh, self.n_RB
        if self.native and rb <= n:
            self.V2V_Interference_all = native_sim.interference_db(self.dest, self.V2V_channels_with_fastfading, self.V2I_power_dB,
                                                                   self.vehAntGain, self.vehNoiseFigure, self.sig2)
            return
        r = np.arange(rb)
        out = np.zeros((E, n, 1, rb)) + self.sig2
        v2v = self.V2V_channels_with_fastfading
        out += 10 ** ((self.V2I_power_dB - v2v[np.arange(E)[:, None, None], r[None, None, :], self.dest[:, :, None], r[None, None, :]][:, :, None, :]
                       + 2 * self.vehAntGain - self.vehNoiseFigure) / 10)
        self.V2V_Interference_all = 10 * np.log10(out)

    def act(self, actions):
        """Agent.act (BS_brain.py:366-376) for all environments: rates under `actions`, then one simulator step.
        Only the rates depend on the actions (they are computed on the channels BEFORE the step); mobility, channels, the
        observable interference and the next observation are one library call for all environments (v2xsim_advance), taken
        from the look-ahead worker when it was started after the previous step."""
        self.finish_step()
        rates = self._rates(actions)
        if self._one_call_step():
            self._advance()
            return rates
        self.renew_positions()
        self.renew_channels_fastfading()
        self.Compute_Interference(actions)
        return rates

    def act_deferred(self, actions):
        """act() that returns as soon as the rates are known; the simulator step itself (taking over the look-ahead result,
        starting the next one) is applied at the next call of any method of this object, or by finish_step() -- between the two
        the public arrays still show the state BEFORE the step.  For callers with better things to do first (the agent
        enqueues its replay); trajectories are those of act()."""
        self.finish_step()
        if not self._one_call_step():
            return self.act(actions)
        rates = self._rates(actions)
        self._step_pending = True
        return rates

    def finish_step(self, start_next=True):
        """apply the step act_deferred() left pending (start_next: look ahead again right away)"""
        if self._step_pending:
            self._step_pending = False
            self._advance(start_next)

    def next_packed_observation(self, n_channels=4):
        """observe_packed()[0] of the state AFTER the pending step, without applying it when the look-ahead has it ready"""
        self._foreign_job()
        job = self._ahead
        if self._step_pending and job is not None and job.get("done") and n_channels == self.n_RB:
            return job["out"]["xe"]
        self.finish_step()
        return self.observe_packed(n_channels)[0]

    def _rates(self, actions):
        self._foreign_job()
        if self._ahead is not None and not self._ahead.get("done"):
            native_sim.advance_wait(self._ahead["ticket"])     # (the pool is the library's only one: free it for the rates)
            self._ahead["done"] = True
        return self.compute_reward_with_channel_selection(actions)

    # ------------------------------------------------------------------ the step as one library call (+ look-ahead)
    def _one_call_step(self):
        n, rb = self.n_Veh, self.n_RB
        return (self.native and self._mt_keys is not None and 2 < n <= 31 and rb <= n and 3 * rb + 1 <= 16
                and not (n + n * n + 2 * n * rb + 2 * n * n * rb) & 1)

    def _start_job(self, ahead):
        """One v2xsim_advance from the current state into fresh arrays; ahead: on the library's worker thread.
        -> the job (its output arrays), or None when the worker is busy with another simulator's look-ahead."""
        if any(s.gauss_next is not None for s in self.streams):
            raise RuntimeError("a stream holds a cached gauss value")
        E, n, rb, p = self.E, self.n_Veh, self.n_RB, self._proto
        n_u = n + n * n + 2 * n * rb + 2 * n * n * rb
        if self._scratch is None or self._scratch.shape != (E, 2 * n_u):
            self._scratch = np.empty((E, 2 * n_u))
        f64, c = np.float64, np.ascontiguousarray
        out = {"keys": np.empty_like(self._mt_keys), "mtpos": np.empty_like(self._mt_pos), "xy": np.empty((E, n, 2)),
               "dirs": np.empty((E, n), np.int8), "v2i_shadow": np.empty((E, n)), "v2v_shadow": np.empty((E, n, n)),
               "v2v_abs": np.empty((E, n, n)), "v2i_abs": np.empty((E, n)), "v2v_ff": np.empty((E, n, n, rb)),
               "v2i_ff": np.empty((E, n, rb)), "interf_db": np.empty((E, n, 1, rb)), "state": np.empty((E, n, 3 * rb + 1)),
               "adj": np.empty((E, n, n)), "xe": np.empty((E, n, 16), np.float32), "mask": np.empty((E, n), np.int32),
               "col": np.empty((E, n * (n - 2)), np.int32), "regular": np.empty(E, np.uint8), "scratch": self._scratch}
        st = self._job_static
        if st is None or st[0] != (E, n, rb):                  # the constant half of the argument block, once per episode shape
            tabs = [c(np.asarray(t, f64)) for t in (p.up_lanes, p.down_lanes, p.left_lanes, p.right_lanes)]
            t = native_sim.AdvanceArgs()
            t.E, t.n, t.rb, t.n_lanes = E, n, rb, len(tabs[0])
            t.timestep, t.width, t.height = p.timestep, p.width, p.height
            t.p_v2v, t.p_v2i = self.V2V_power_dB_List[self.fixed_v2v_power_index], self.V2I_power_dB
            t.veh_gain, t.veh_nf, t.sig2 = self.vehAntGain, self.vehNoiseFigure, self.sig2
            t.up, t.down, t.left, t.right = (x.ctypes.data for x in tabs)
            st = self._job_static = ((E, n, rb), t, tabs)
        ins = {"tabs": st[2], "vel": c(self.vel, f64), "dest": c(self.dest, np.int64),
               "keys_in": self._mt_keys, "mtpos_in": self._mt_pos, "xy_in": c(self.pos, f64), "dirs_in": c(self.dirs, np.int8),
               "v2i_shadow_in": c(self._v2i_shadow, f64), "v2v_shadow_in": c(self._v2v_shadow, f64)}
        a = native_sim.AdvanceArgs.from_buffer_copy(st[1])
        for k, v in ins.items():
            if k != "tabs":
                setattr(a, k, v.ctypes.data)
        for k, v in out.items():
            setattr(a, k, v.ctypes.data)
        job = {"out": out, "ins": ins, "args": a}            # (the arrays the library reads and writes stay alive with the job)
        if ahead:
            job["pid"] = os.getpid()                           # (a forked child has the job dict but not the threads working on it)
            job["ticket"] = native_sim.advance_start(a)
            return job if job["ticket"] else None
        native_sim.advance(a)
        return job

    def __del__(self):
        try:
            self._drop_lookahead()                             # the worker writes into arrays this object keeps alive
        except Exception:
            pass

    def _drop_lookahead(self):
        """A started look-ahead step is abandoned (its inputs were never written): wait for the worker and forget the result."""
        if self._step_pending:                                 # ... but a pending step is applied first: its draws come before
            self.finish_step(start_next=False)                 # whatever the caller is about to do with the streams
        if self._ahead is not None:
            if not self._ahead.get("done") and not self._foreign_job():
                native_sim.advance_wait(self._ahead["ticket"])
            self._ahead = None

    def _foreign_job(self):
        """The look-ahead job was started by ANOTHER process (this one is a fork taken while it was in flight): the pool threads
        that were writing its arrays do not exist here and the library's bookkeeping was reset (pool_after_fork), so waiting
        returns at once on half-written outputs.  Its inputs were never touched: drop it, the step is recomputed synchronously."""
        job = self._ahead
        if job is not None and job.get("pid", os.getpid()) != os.getpid():
            self._ahead = None
            return True
        return False

    def _advance(self, start_next=True):
        self._foreign_job()
        job = self._ahead
        if job is not None:
            if not job.get("done"):
                native_sim.advance_wait(job["ticket"])
            self._ahead = None
        else:
            job = self._start_job(False)
        o = job["out"]
        self._mt_keys[:] = o["keys"]                           # in place: the MTStream objects are attached to these rows
        self._mt_pos[:] = o["mtpos"]
        self.pos[:] = o["xy"]
        self.dirs[:] = o["dirs"]
        self._v2i_shadow, self._v2v_shadow = o["v2i_shadow"], o["v2v_shadow"]
        self.V2V_channels_abs, self.V2I_channels_abs = o["v2v_abs"], o["v2i_abs"]
        self.V2V_channels_with_fastfading, self.V2I_channels_with_fastfading = o["v2v_ff"], o["v2i_ff"]
        self.V2V_Interference_all = o["interf_db"]
        self._obs = (o["v2v_ff"], o["v2i_ff"], o["state"], o["adj"], o["xe"], o["mask"], o["col"], o["regular"].astype(bool))
        if self.lookahead and start_next:
            self._ahead = self._start_job(True)

    def native_rollout(self, T, n_actions, policy):
        """T sequential transitions of this ONE simulator (E == 1) in one library call (v2xsim_rollout, include/v2xsim.h): the
        reference's own loop shape, Agent.generate_d2d_transition with 50 transitions before every replay (BS_brain.py:409-553,
        :818-832).  policy: eps_max, eps_min, eps_per_step, eps_steps, step_no0 (the epsilon schedule as the agent evaluates it),
        predict / predict_ctx (addresses of a C callback `int (*)(void*)` and its closure: scores the graph in xe_pin / col_pin
        into q_pin), xe_pin [n, 16] float32, col_pin [n (n-2)] int32, q_pin [n, n_actions] float32; batch_predict: the callback
        scores all T graphs in ONE call (the buffers hold T graphs; see include/v2xsim.h).  Epsilon draws and random
        actions are taken from numpy's process-wide generator exactly as np.random.random() / np.random.randint would.
        -> dict(done, rc, xe, xe_next, col, mask, regular, action, v2v_rate [done, n, 1], v2i_rate [done, m], eps_last, n_greedy);
        the simulator stands at the state after `done` transitions (done < T: transition `done` needs the caller's general path --
        a link that is its own receiver -- or its predict failed, rc says which)."""
        if self.E != 1 or not self._one_call_step() or not self.packed_ok(self.n_RB):
            raise RuntimeError("native_rollout: one simulator of 3..31 links on the native library")
        self._drop_lookahead()
        n, rb, p = self.n_Veh, self.n_RB, self._proto
        m, ne, W = min(rb, n), n * (n - 2), 3 * rb + 1
        self.observe_packed(rb)
        ob = self._observation(rb)
        if ob is None:
            ob = native_sim.observe_packed(self.dest, np.ascontiguousarray(self.V2V_channels_with_fastfading),
                                           np.ascontiguousarray(self.V2I_channels_with_fastfading),
                                           self.V2V_power_dB_List[self.fixed_v2v_power_index], rb)
        f64, c = np.float64, np.ascontiguousarray
        st = {"xy": c(self.pos, f64).copy(), "dirs": c(self.dirs, np.int8).copy(),
              "v2i_shadow": c(self._v2i_shadow, f64).copy(), "v2v_shadow": c(self._v2v_shadow, f64).copy(),
              "v2v_abs": c(self.V2V_channels_abs, f64).copy(), "v2i_abs": c(self.V2I_channels_abs, f64).copy(),
              "v2v_ff": c(self.V2V_channels_with_fastfading, f64).copy(), "v2i_ff": c(self.V2I_channels_with_fastfading, f64).copy(),
              "interf_db": np.zeros((1, n, 1, rb)),             # (output only: Compute_Interference of the final state)
              "state": c(ob[0], f64).copy(), "adj": c(ob[1], f64).copy(), "xe": c(ob[2], np.float32).copy(),
              "mask": c(ob[3], np.int32).copy(), "col": c(ob[4], np.int32).copy(), "regular": np.asarray(ob[5]).astype(np.uint8).copy(),
              "interference": np.zeros(rb), "v2i_interf": np.zeros(rb), "v2v_interf": np.zeros(n)}
        out = {"xe": np.empty((T, n, 16), np.float32), "xe_next": np.empty((T, n, 16), np.float32),
               "col": np.empty((T, max(ne, 1)), np.int32), "mask": np.empty((T, n), np.int32), "regular": np.zeros(T, np.uint8),
               "action": np.zeros((T, n), np.int64), "v2v_rate": np.zeros((T, n, 1)), "v2i_rate": np.zeros((T, m))}
        name, key, pos, has_gauss, cached = np.random.get_state()
        if name != 'MT19937':
            raise RuntimeError("np.random is not on MT19937")
        key, npos = np.ascontiguousarray(key, np.uint32).copy(), np.array([pos], np.int32)
        n_greedy = np.zeros(1, np.int32)
        tabs = [c(np.asarray(t, f64)) for t in (p.up_lanes, p.down_lanes, p.left_lanes, p.right_lanes)]
        vel, dest = c(self.vel, f64), c(self.dest, np.int64)
        a = native_sim.RolloutArgs()
        a.n, a.rb, a.n_lanes, a.T, a.n_actions = n, rb, len(tabs[0]), int(T), int(n_actions)
        a.timestep, a.width, a.height = p.timestep, p.width, p.height
        a.up, a.down, a.left, a.right = (t.ctypes.data for t in tabs)
        a.vel, a.dest = vel.ctypes.data, dest.ctypes.data
        a.p_v2v, a.p_v2i = self.V2V_power_dB_List[self.fixed_v2v_power_index], self.V2I_power_dB
        a.veh_gain, a.veh_nf, a.sig2, a.bs_gain, a.bs_nf = self.vehAntGain, self.vehNoiseFigure, self.sig2, self.bsAntGain, self.bsNoiseFigure
        a.keys, a.mtpos = self._mt_keys.ctypes.data, self._mt_pos.ctypes.data
        for k_, v_ in st.items():
            setattr(a, k_, v_.ctypes.data)
        a.np_key, a.np_pos = key.ctypes.data, npos.ctypes.data
        a.eps_max, a.eps_min, a.eps_per_step = float(policy["eps_max"]), float(policy["eps_min"]), float(policy["eps_per_step"])
        a.eps_steps, a.step_no0 = float(policy["eps_steps"]), int(policy["step_no0"])
        a.predict, a.predict_ctx = policy.get("predict"), policy.get("predict_ctx")
        # batch_predict: ONE predict for all T observations (they do not depend on the actions: include/v2xsim.h); the pinned
        # buffers then hold T graphs.  Not for a graph with a link that is its own receiver (every greedy transition would end the call).
        a.batch_predict = 1 if (policy.get("batch_predict") and bool(np.all(ob[5]))) else 0
        g_ = int(T) if a.batch_predict else 1
        pins = [policy.get(k_) for k_ in ("xe_pin", "col_pin", "q_pin")]
        if a.predict:
            if (pins[0].dtype != np.float32 or pins[0].size < g_ * n * 16 or pins[1].dtype != np.int32 or pins[1].size < g_ * max(ne, 1)
                    or pins[2].dtype != np.float32 or pins[2].size < g_ * n * n_actions):
                raise ValueError("native_rollout: xe_pin / col_pin / q_pin have the wrong type or size")
            a.xe_pin, a.col_pin, a.q_pin = (t.ctypes.data for t in pins)
        a.t_xe, a.t_xe_next, a.t_col, a.t_mask = (out[k_].ctypes.data for k_ in ("xe", "xe_next", "col", "mask"))
        a.t_regular, a.t_action = out["regular"].ctypes.data, out["action"].ctypes.data
        a.t_v2v_rate, a.t_v2i_rate = out["v2v_rate"].ctypes.data, out["v2i_rate"].ctypes.data
        a.n_greedy = n_greedy.ctypes.data
        if any(s_.gauss_next is not None for s_ in self.streams):
            raise RuntimeError("a stream holds a cached gauss value")
        rc = native_sim.rollout(a)
        np.random.set_state((name, key, int(npos[0]), has_gauss, cached))
        if rc in (-1, -2, -3):
            raise RuntimeError("v2xsim_rollout failed (%d)" % rc)
        done = T if rc == T else (-1000 - rc if rc <= -1000 else -4 - rc)
        if done > 0:                                           # the simulator's public arrays: the state after `done` transitions
            self.pos[:] = st["xy"]
            self.dirs[:] = st["dirs"]
            self._v2i_shadow, self._v2v_shadow = st["v2i_shadow"], st["v2v_shadow"]
            self.V2V_channels_abs, self.V2I_channels_abs = st["v2v_abs"], st["v2i_abs"]
            self.V2V_channels_with_fastfading, self.V2I_channels_with_fastfading = st["v2v_ff"], st["v2i_ff"]
            self.V2V_Interference_all = st["interf_db"]
            self._obs = (st["v2v_ff"], st["v2i_ff"], st["state"], st["adj"], st["xe"], st["mask"], st["col"], st["regular"].astype(bool))
            self.V2I_Interference, self.V2V_Interference = st["v2i_interf"].reshape(1, rb), st["v2v_interf"].reshape(1, n, 1)
        res = {k_: v_[:done] for k_, v_ in out.items()}
        res["regular"] = res["regular"].astype(bool)
        res.update(done=done, rc=rc, eps_last=float(a.eps_last), n_greedy=int(n_greedy[0]))
        return res

    def _observation(self, n_channels):
        """the cached observation of the current channels (state, adj, xe, mask, col, regular), or None"""
        ob = self._obs
        if (ob is not None and n_channels == self.n_RB and ob[0] is self.V2V_channels_with_fastfading
                and ob[1] is self.V2I_channels_with_fastfading):
            return ob[2:]
        return None

    def observe_packed(self, n_channels=4):
        """observe() in the engine's packed form (rl/replay.py, include/v2xgnn.h): xe [E, N, 16] float32 = the observation rows
        cast like Keras casts the feed, source masks [E, N] int32, CSR sources by destination [E, N (N-2)] int32 (zeros
        for a graph where some link is its own receiver) and the regular flags [E].  Same arrays every call until the
        simulator moves: read-only for the caller.  Needs the native library (packed_ok())."""
        self.finish_step()
        ob = self._observation(n_channels)
        if ob is None:
            if not self.packed_ok(n_channels):
                raise RuntimeError("observe_packed needs libv2xsim.so, 3..31 links and n_channels == n_RB")
            v2v, v2i = self.V2V_channels_with_fastfading, self.V2I_channels_with_fastfading
            ob = native_sim.observe_packed(self.dest, v2v, v2i, self.V2V_power_dB_List[self.fixed_v2v_power_index], n_channels)
            if v2v.flags.c_contiguous and v2i.flags.c_contiguous:
                self._obs = (v2v, v2i) + tuple(ob)
        return ob[2], ob[3], ob[4], ob[5]

    def packed_ok(self, n_channels=4):
        return bool(self.native and n_channels == self.n_RB and 2 < self.n_Veh <= 31 and 3 * n_channels + 1 <= 16)

    # ------------------------------------------------------------------ the agent's view
    def observe(self, n_channels=4):
        """Agent.observe for all environments -> D2D_State [E, N, 2C+1+C] = [V2V gain | V2I gain | power | edge gain]
        (BS_brain.py:389-407, :458-467) and the adjacency [E, N, N] (Adj[p, q] = 0 for p == q and for the receiver p of
        link q, :441-445)."""
        self.finish_step()
        E, n, C = self.E, self.n_Veh, n_channels
        ob = self._observation(C)
        if ob is not None:                                     # computed with the step (same arrays until the simulator moves)
            return ob[0], ob[1]
        if self.native and C == self.n_RB and n > 2:
            return native_sim.observe(self.dest, self.V2V_channels_with_fastfading, self.V2I_channels_with_fastfading,
                                      self.V2V_power_dB_List[self.fixed_v2v_power_index], C)
        A, Bc = 80, 60
        v2v, v2i = self.V2V_channels_with_fastfading, self.V2I_channels_with_fastfading
        ei = np.arange(E)[:, None]
        k = np.arange(n)[None, :]
        dst = self.dest
        ch = (v2v[ei, k, dst, :] - A) / Bc
        towards = v2v[ei[:, :, None], np.arange(n)[None, None, :], dst[:, :, None], :]             # [E, k, p, C]: p -> receiver of k
        edge = (((np.sum(towards, axis=2) - v2v[ei, dst, dst, :]) - (n - 1) * A) / Bc - ch) / (n - 2)
        state = np.zeros((E, n, 2 * C + 1 + C))
        state[:, :, 0:C] = ch
        state[:, :, C:2 * C] = (v2i - A) / Bc
        state[:, :, 2 * C] = self.V2V_power_dB_List[self.fixed_v2v_power_index]
        state[:, :, 2 * C + 1:] = edge
        adj = np.ones((E, n, n)) - np.eye(n)[None]
        adj[ei, dst, k] = 0
        return state, adj

    # ------------------------------------------------------------------ single-environment views (tests, tools)
    def directions(self, e=0):
        return [_DIRS[c] for c in self.dirs[e]]
