"""One MT19937 stream with the draw algorithms of CPython's `random` module, on a numpy generator.

The reference simulator draws everything from the process-wide stdlib generator (Environment.py:14-42 gauss, :219-233
randint, :251.. uniform, :375 sample).  A batched simulator needs one independent stream PER environment, and it needs
them cheap: `MTStream(seed)` reproduces `random.Random(seed)` draw for draw (same MT19937 state, same 53-bit doubles,
same getrandbits-based integers, same Box-Muller pairing of gauss() including the cached second value), but bulk draws
are one numpy call instead of one Python call per number.  `MTStream.borrow_stdlib()` / `.return_stdlib()` lend the
process-wide stdlib state to a stream and hand it back, which is how the single-environment path stays in lock-step
with code that seeds / uses `random` directly.
"""
import random as _random

import numpy as np

_TWOPI = 2.0 * np.pi


class MTStream(object):
    def __init__(self, seed=None, _state=None):
        self._rs = np.random.RandomState(0)
        self._ext = None               # attach(): the MT19937 state lives in caller-owned arrays (bulk draws in C)
        self.session = None            # a list while the owner is inside one `with env._rng()` block
        self._live = False             # pulled in during the current session: no per-draw synchronisation until it ends
        self.on_touch = None           # called before the external state is read (the owner drops work computed ahead from it)
        self.gauss_next = None
        if _state is None:
            _state = _random.Random(seed).getstate()       # CPython seeds through init_by_array: reuse it verbatim
        self._import(_state)

    # ------------------------------------------------------------------ externally held state (rl/native_sim.py)
    def attach(self, keys_row, pos_arr, idx):
        """From now on the authoritative MT19937 state is keys_row[624] / pos_arr[idx] (rows of arrays the batched simulator
        hands to libv2xsim.so for its bulk draws); every numpy-side draw of this object pulls the state in first and pushes
        it back afterwards, so the two views never diverge.  Inside a session (`self.session` is a list: the owner's
        `with _rng()` block) the first draw pulls the state in and the owner pushes it back once at the end."""
        _, keys, pos = self._rs.get_state()[:3]
        keys_row[:] = keys
        pos_arr[idx] = pos
        self._ext = (keys_row, pos_arr, idx)

    def _pull(self):
        if self._ext is None or self._live:
            return
        if self.on_touch is not None:
            self.on_touch()
        keys_row, pos_arr, idx = self._ext
        self._rs.set_state(('MT19937', keys_row, int(pos_arr[idx])))
        if self.session is not None:
            self._live = True
            self.session.append(self)

    def _push(self, end_of_session=False):
        if self._ext is None or (self._live and not end_of_session):
            return
        keys_row, pos_arr, idx = self._ext
        _, keys, pos = self._rs.get_state()[:3]
        keys_row[:] = keys
        pos_arr[idx] = pos
        if end_of_session:
            self._live = False

    # ------------------------------------------------------------------ state exchange with the stdlib generator
    def _import(self, state):
        _, st, gauss_next = state
        self._rs.set_state(('MT19937', np.array(st[:-1], dtype=np.uint32), st[-1]))
        self.gauss_next = gauss_next
        self._push()

    def _export(self):
        self._pull()
        _, keys, pos = self._rs.get_state()[:3]
        return (3, tuple(keys.tolist()) + (int(pos),), self.gauss_next)

    @classmethod
    def borrow_stdlib(cls, inst=None):
        """A stream that continues the stdlib generator `inst` (default: the module-level one)."""
        return cls(_state=(inst or _random._inst).getstate())

    def return_stdlib(self, inst=None):
        (inst or _random._inst).setstate(self._export())

    # ------------------------------------------------------------------ random.Random's algorithms
    def random(self):
        self._pull()
        v = float(self._rs.random_sample())
        self._push()
        return v

    def random_array(self, k):
        self._pull()
        v = self._rs.random_sample(k)
        self._push()
        return v

    def uniform(self, a, b):
        return a + (b - a) * self.random()

    def getrandbits(self, k):
        """k <= 32: one 32-bit output, top bits (random.getrandbits)."""
        if not 0 < k <= 32:
            raise ValueError("getrandbits: 1..32 bits")
        self._pull()
        v = int.from_bytes(self._rs.bytes(4), 'little') >> (32 - k)
        self._push()
        return v

    def _randbelow(self, n):
        k = int(n).bit_length()
        r = self.getrandbits(k)
        while r >= n:
            r = self.getrandbits(k)
        return r

    def randrange(self, start, stop=None):
        if stop is None:
            start, stop = 0, start
        width = stop - start
        if width <= 0:
            raise ValueError("empty range for randrange()")
        return start + self._randbelow(width)

    def randint(self, a, b):
        return self.randrange(a, b + 1)

    def sample(self, population, k):
        """random.sample (CPython 3.10: pool method for small populations, rejection set otherwise)."""
        population = list(population)
        n = len(population)
        if not 0 <= k <= n:
            raise ValueError("Sample larger than population or is negative")
        result = [None] * k
        setsize = 21
        if k > 5:
            setsize += 4 ** int(np.ceil(np.log(k * 3) / np.log(4)))
        if n <= setsize:
            pool = list(population)
            for i in range(k):
                j = self._randbelow(n - i)
                result[i] = pool[j]
                pool[j] = pool[n - i - 1]
        else:
            selected = set()
            for i in range(k):
                j = self._randbelow(n)
                while j in selected:
                    j = self._randbelow(n)
                selected.add(j)
                result[i] = population[j]
        return result

    def gauss_array(self, shape, sigma):
        """Row-major array of gauss(0, sigma) draws: cos value now, sin value cached for the next draw, exactly like
        random.gauss (so consecutive calls continue each other's pairing)."""
        n = int(np.prod(shape))
        out = np.empty(n, dtype=np.float64)
        i = 0
        if n and self.gauss_next is not None:
            out[0] = self.gauss_next * sigma
            self.gauss_next = None
            i = 1
        pairs = (n - i + 1) // 2
        if pairs:
            u = self.random_array(2 * pairs)
            x2pi = u[0::2] * _TWOPI
            g2rad = np.sqrt(-2.0 * np.log(1.0 - u[1::2]))
            z = np.stack([np.cos(x2pi) * g2rad, np.sin(x2pi) * g2rad], axis=1).reshape(-1)
            out[i:] = z[:n - i] * sigma
            if (n - i) & 1:
                self.gauss_next = float(z[-1])
        return out.reshape(shape)


def gauss_uniforms(streams, n):
    """The uniforms behind n gauss() draws of every stream, [E, 2 * ceil(n/2)]; only valid while no stream holds a cached
    value (checked) -- the batched simulator's draw counts per step are even."""
    pairs = (n + 1) // 2
    u = np.empty((len(streams), 2 * pairs))
    for e, s in enumerate(streams):
        if s.gauss_next is not None:
            raise RuntimeError("stream %d holds a cached gauss value" % e)
        u[e] = s.random_array(2 * pairs)
    return u


def box_muller(u, sigma=1.0):
    """[E, 2p] uniforms -> [E, 2p] gauss draws in random.gauss order (cos, sin, cos, sin, ...)."""
    x2pi = u[:, 0::2] * _TWOPI
    g2rad = np.sqrt(-2.0 * np.log(1.0 - u[:, 1::2]))
    return np.stack([np.cos(x2pi) * g2rad, np.sin(x2pi) * g2rad], axis=2).reshape(u.shape[0], -1) * sigma
