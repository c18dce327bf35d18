"""V2X simulator counterpart of the reference `Environ` (/root/reference/Environment.py:179-506), needed to run
the DQN loop of BASELINE configs[0] / configs[2] where the reference's Python does not travel.

Same public surface as the reference object the agent talks to -- `new_random_game`, `renew_positions`,
`renew_channels_fastfading`, `renew_neighbor`, `compute_reward_with_channel_selection`, `Compute_Interference`;
attributes `vehicles[i].position/direction/velocity/destinations`, `V2V_channels_with_fastfading`,
`V2I_channels_with_fastfading`, `V2V_power_dB_List`, `fixed_v2v_power_index`, `n_Veh`, `n_RB`, `n_Neighbor`,
`activate_links` -- but written from the behaviour, not from the text: channel models are vectorised numpy, the
mobility rules are one data table instead of four hand-unrolled branches.  Random numbers are drawn from the
stdlib `random` module in the reference's order (Environment.py:14-42, :219-233, :251.., :375), so a seeded run
reproduces the reference trajectory (tests/golden/golden_env_*.npz, captured from the reference itself).
"""
import random

import numpy as np


_TWOPI = 2.0 * np.pi
_mt = np.random.RandomState(0)        # scratch generator: only ever runs on a state borrowed from the stdlib one


def _uniforms(k):
    """k draws of stdlib random.random(), produced in bulk: CPython's generator and numpy's legacy RandomState are the
    same MT19937 with the same 53-bit double construction, so the stdlib state is lent to numpy and handed back."""
    ver, st, gauss_next = random.getstate()
    _mt.set_state(('MT19937', np.array(st[:-1], dtype=np.uint32), st[-1]))
    u = _mt.random_sample(k)
    _, keys, pos = _mt.get_state()[:3]
    random.setstate((ver, tuple(keys.tolist()) + (int(pos),), gauss_next))
    return u


def _gauss(shape, sigma):
    """row-major array of random.gauss(0, sigma) draws == the nested loops of RandomGenerate (Environment.py:14-42).
    Same stream and same pairing as CPython's gauss() (cos value now, sin value cached for the next call), computed
    with array math instead of one Python call per element (1.9 M calls per 500 simulator steps at 20 links)."""
    n = int(np.prod(shape))
    out = np.empty(n, dtype=np.float64)
    inst = random._inst
    i = 0
    if n and inst.gauss_next is not None:
        out[0] = inst.gauss_next * sigma
        inst.gauss_next = None
        i = 1
    pairs = (n - i + 1) // 2
    if pairs:
        u = _uniforms(2 * pairs)
        x2pi = u[0::2] * _TWOPI
        g2rad = np.sqrt(-2.0 * np.log(1.0 - u[1::2]))
        z = np.stack([np.cos(x2pi) * g2rad, np.sin(x2pi) * g2rad], axis=1).reshape(-1)
        out[i:] = z[:n - i] * sigma
        if (n - i) & 1:
            random._inst.gauss_next = float(z[-1])
    return out.reshape(shape)


class Vehicle(object):
    """Environment.py:168-176"""

    def __init__(self, position, direction, velocity):
        self.position = position
        self.direction = direction
        self.velocity = velocity
        self.neighbors = []
        self.destinations = []


class Environ(object):
    # radio constants: Environment.py:183-212 (Environ), :48-57 (V2V), :127-133 (V2I)
    V2V_H = 1.5
    FC = 2
    V2V_DECORR = 10.0
    V2V_SHADOW_STD = 3
    V2I_H_BS, V2I_H_MS = 25, 1.5
    V2I_DECORR = 50.0
    V2I_SHADOW_STD = 8
    BS_POSITION = (750 / 2, 1299 / 2)

    def __init__(self, down_lane, up_lane, left_lane, right_lane, width, height):
        self.timestep = 0.01
        self.down_lanes, self.up_lanes = list(down_lane), list(up_lane)
        self.left_lanes, self.right_lanes = list(left_lane), list(right_lane)
        self.width, self.height = width, height
        self.vehicles = []
        self.V2V_power_dB = 23
        self.V2I_power_dB = 23
        self.V2V_power_dB_List = [23, 10, 5]
        self.fixed_v2v_power_index = 1
        self.sig2_dB = -114
        self.bsAntGain, self.bsNoiseFigure = 8, 5
        self.vehAntGain, self.vehNoiseFigure = 3, 9
        self.sig2 = 10 ** (self.sig2_dB / 10)
        self.n_RB, self.n_Veh, self.n_Neighbor = 4, 4, 1
        self.n_step = 0
        # the reference constructs its channel objects here already (Environment.py:208-209), which draws their
        # initial shadowing: keep the draws so that a seeded run consumes the RNG identically
        self._v2v_shadow = _gauss((self.n_Veh, self.n_Veh), self.V2V_SHADOW_STD)
        self._v2i_shadow = _gauss((self.n_Veh,), self.V2I_SHADOW_STD)
        # (direction) -> moving axis, sign, [(crossing lanes, new direction, sign of the side step, sign of the gap term)]
        # encodes Environment.py:247-324 including its asymmetry (turning onto a right lane ADDS the gap)
        self._moves = {
            'u': (1, +1, [(self.left_lanes, 'l', -1, -1), (self.right_lanes, 'r', +1, +1)]),
            'd': (1, -1, [(self.left_lanes, 'l', -1, -1), (self.right_lanes, 'r', +1, +1)]),
            'r': (0, +1, [(self.up_lanes, 'u', +1, -1), (self.down_lanes, 'd', -1, -1)]),
            'l': (0, -1, [(self.up_lanes, 'u', +1, -1), (self.down_lanes, 'd', -1, -1)]),
        }

    # ------------------------------------------------------------------ construction
    def add_new_vehicles_by_number(self, n):
        """Environment.py:217-234: n groups of 4 vehicles (down, up, left, right) on one random lane index."""
        for _ in range(n):
            ind = random.randrange(0, len(self.down_lanes))
            self.vehicles.append(Vehicle([self.down_lanes[ind], random.randint(0, self.height)], 'd', random.randint(10, 15)))
            self.vehicles.append(Vehicle([self.up_lanes[ind], random.randint(0, self.height)], 'u', random.randint(10, 15)))
            self.vehicles.append(Vehicle([random.randint(0, self.width), self.left_lanes[ind]], 'l', random.randint(10, 15)))
            self.vehicles.append(Vehicle([random.randint(0, self.width), self.right_lanes[ind]], 'r', random.randint(10, 15)))
        nv = len(self.vehicles)
        self.V2V_Shadowing = _gauss((nv, nv), 3)          # drawn (and never used) by the reference as well
        self.V2I_Shadowing = _gauss((nv,), 8)
        self.delta_distance = np.asarray([c.velocity for c in self.vehicles])

    def new_random_game(self, n_Veh=0):
        """Environment.py:495-506"""
        self.n_step = 0
        self.vehicles = []
        if n_Veh > 0:
            if n_Veh % 4:
                raise ValueError("n_Veh must be a multiple of 4 (vehicles are added four at a time, one per direction: "
                                 "Environment.py:217-231); got %d" % n_Veh)
            self.n_Veh = n_Veh
        self.add_new_vehicles_by_number(int(self.n_Veh / 4))
        self._v2v_shadow = _gauss((self.n_Veh, self.n_Veh), self.V2V_SHADOW_STD)     # V2Vchannels.__init__ (:59)
        self._v2i_shadow = _gauss((self.n_Veh,), self.V2I_SHADOW_STD)                # V2Ichannels.__init__ (:136)
        self.renew_channels_fastfading()
        self.renew_neighbor()
        self.activate_links = np.ones((self.n_Veh, self.n_Neighbor), dtype='bool')

    # ------------------------------------------------------------------ mobility
    def renew_positions(self):
        """One 10 ms mobility step (Environment.py:236-345): straight motion, a 0.4-probability turn at every crossing
        lane reached during the step, and re-entry on the outermost lane when the map is left."""
        for v in self.vehicles:
            dd = v.velocity * self.timestep
            axis, sign, options = self._moves[v.direction]
            a, o = v.position[axis], v.position[1 - axis]
            turned = False
            for lanes, new_dir, side_sign, gap_sign in options:
                for lane in lanes:
                    reached = (a <= lane and a + dd >= lane) if sign > 0 else (a >= lane and a - dd <= lane)
                    if reached and random.uniform(0, 1) < 0.4:
                        gap = sign * (lane - a)
                        new_o = o + side_sign * (dd + gap_sign * gap)
                        v.position = [new_o, lane] if axis == 1 else [lane, new_o]
                        v.direction = new_dir
                        turned = True
                        break
                if turned:
                    break
            if not turned:
                v.position[axis] = a + dd if sign > 0 else a - dd
            x, y = v.position
            if x < 0 or y < 0 or x > self.width or y > self.height:
                if v.direction == 'u':
                    v.direction, v.position = 'r', [x, self.right_lanes[-1]]
                elif v.direction == 'd':
                    v.direction, v.position = 'l', [x, self.left_lanes[0]]
                elif v.direction == 'l':
                    v.direction, v.position = 'u', [self.up_lanes[0], y]
                elif v.direction == 'r':
                    v.direction, v.position = 'd', [self.down_lanes[-1], y]

    def renew_neighbor(self):
        """Environment.py:360-376: the receiver of link i is drawn among its nearest vehicles, excluding itself
        and the two farthest ones."""
        z = np.array([[complex(c.position[0], c.position[1]) for c in self.vehicles]])
        dist = abs(z.T - z)
        for i, v in enumerate(self.vehicles):
            order = np.argsort(dist[:, i])
            v.neighbors = [order[j + 1] for j in range(self.n_Neighbor)]
            v.actions = []
            v.destinations = random.sample(list(order[1:(len(order) - 2)]), self.n_Neighbor)

    # ------------------------------------------------------------------ channels
    def _v2v_pathloss(self, pos):
        """WINNER-style urban V2V path loss, LOS when the vehicles share a street (Environment.py:94-122)."""
        d1 = np.abs(pos[..., :, None, 0] - pos[..., None, :, 0])       # (leading batch axes allowed: batched_env.py)
        d2 = np.abs(pos[..., :, None, 1] - pos[..., None, :, 1])
        d = np.hypot(d1, d2) + 0.001
        fc, h = self.FC, self.V2V_H
        d_bp = 4 * (h - 1) * (h - 1) * fc * (10 ** 9) / (3 * 10 ** 8)
        off = 41 + 20 * np.log10(fc / 5)

        def los(x):
            x = np.maximum(x, 1e-300)
            near = 22.7 * np.log10(3) + off
            mid = 22.7 * np.log10(x) + off
            far = 40.0 * np.log10(x) + 9.45 - 17.3 * np.log10(h) - 17.3 * np.log10(h) + 2.7 * np.log10(fc / 5)
            return np.where(x <= 3, near, np.where(x < d_bp, mid, far))

        def nlos(da, db):
            db = np.maximum(db, 1e-300)
            nj = np.maximum(2.8 - 0.0024 * db, 1.84)
            return los(da) + 20 - 12.5 * nj + 10 * nj * np.log10(db) + 3 * np.log10(fc / 5)

        return np.where(np.minimum(d1, d2) < 7, los(d), np.minimum(nlos(d1, d2), nlos(d2, d1)))

    def _v2i_pathloss(self, pos):
        """Environment.py:140-146"""
        dist = np.hypot(np.abs(pos[..., 0] - self.BS_POSITION[0]), np.abs(pos[..., 1] - self.BS_POSITION[1]))
        return 128.1 + 37.6 * np.log10(np.sqrt(dist ** 2 + (self.V2I_H_BS - self.V2I_H_MS) ** 2) / 1000)

    def renew_channel(self):
        """Large-scale fading: path loss + spatially correlated log-normal shadowing (Environment.py:378-393)."""
        pos = np.array([c.position for c in self.vehicles], dtype=np.float64)
        vel = np.asarray([c.velocity for c in self.vehicles])
        dd = 0.002 * vel
        n = self.n_Veh
        g = _gauss((n + n * n,), 1.0)     # ONE bulk draw, consumed in the reference's order: V2I shadow, then V2V shadow
        self._v2i_shadow = (np.exp(-1 * (dd / self.V2I_DECORR)) * self._v2i_shadow
                            + np.sqrt(1 - np.exp(-2 * (dd / self.V2I_DECORR))) * (g[:n] * self.V2I_SHADOW_STD))
        ddm = dd[:, None] + dd[None, :]
        self._v2v_shadow = (np.exp(-1 * (ddm / self.V2V_DECORR)) * self._v2v_shadow
                            + np.sqrt(1 - np.exp(-2 * (ddm / self.V2V_DECORR))) * (g[n:].reshape(n, n) * self.V2V_SHADOW_STD))
        self.V2V_channels_abs = self._v2v_pathloss(pos) + self._v2v_shadow + 50 * np.identity(len(self.vehicles))
        self.V2I_channels_abs = self._v2i_pathloss(pos) + self._v2i_shadow

    def renew_channels_fastfading(self):
        """Large-scale update + Rayleigh fast fading per resource block (Environment.py:395-406, :88-92, :160-165)."""
        self.renew_channel()
        n, rb = self.n_Veh, self.n_RB
        g = _gauss((2 * n * rb + 2 * n * n * rb,), 1)       # one bulk draw: V2I re, V2I im, V2V re, V2V im
        a, b = n * rb, n * n * rb
        re, im = g[:a].reshape(n, rb), g[a:2 * a].reshape(n, rb)
        v2i_ff = 20 * np.log10(np.abs(1 / np.sqrt(2) * (re + 1j * im)))
        re, im = g[2 * a:2 * a + b].reshape(n, n, rb), g[2 * a + b:].reshape(n, n, rb)
        v2v_ff = 20 * np.log10(np.abs(1 / np.sqrt(2) * (re + 1j * im)))
        self.V2V_channels_with_fastfading = self.V2V_channels_abs[:, :, None] - v2v_ff
        self.V2I_channels_with_fastfading = self.V2I_channels_abs[:, None] - v2i_ff

    # ------------------------------------------------------------------ reward
    def compute_reward_with_channel_selection(self, actions_ch_sel):
        """Shannon rates of every V2V link and of the V2I links under the chosen channels
        (Environment.py:408-458).  Link (i, j) transmits on RB actions[i, j] to vehicle destinations[j];
        V2I link r occupies RB r (vehicle r is its transmitter)."""
        actions = np.asarray(actions_ch_sel).astype(int)
        n, nn, rb = len(self.vehicles), self.n_Neighbor, self.n_RB
        p_v2v = self.V2V_power_dB_List[self.fixed_v2v_power_index]
        act = np.where(self.activate_links, actions, -1)
        tx, nb = np.nonzero(act >= 0)
        ch = act[tx, nb]
        rx = np.array([self.vehicles[i].destinations[j] for i, j in zip(tx, nb)], dtype=int)
        # interference the V2V transmitters cause at the base station, per RB
        interference = np.zeros(rb)
        np.add.at(interference, ch, 10 ** ((p_v2v - self.V2I_channels_with_fastfading[tx, ch]
                                            + self.vehAntGain + self.bsAntGain - self.bsNoiseFigure) / 10))
        self.V2I_Interference = interference + self.sig2
        gain = 2 * self.vehAntGain - self.vehNoiseFigure
        signal = np.zeros((n, nn))
        v2v_int = np.zeros((n, nn))
        signal[tx, nb] = 10 ** ((p_v2v - self.V2V_channels_with_fastfading[tx, rx, ch] + gain) / 10)
        # the V2I transmitter on the same RB (vehicle index == RB index), only for RBs that have one
        has_v2i = ch < n
        v2v_int[tx[has_v2i], nb[has_v2i]] += 10 ** ((self.V2I_power_dB - self.V2V_channels_with_fastfading[
            ch[has_v2i], rx[has_v2i], ch[has_v2i]] + gain) / 10)
        # every other V2V transmitter on the same RB
        same = (ch[:, None] == ch[None, :]) & ~np.eye(len(ch), dtype=bool)
        cross = 10 ** ((p_v2v - self.V2V_channels_with_fastfading[tx[None, :], rx[:, None], ch[:, None]] + gain) / 10)
        v2v_int[tx, nb] += (cross * same).sum(axis=1)
        self.V2V_Interference = v2v_int + self.sig2
        v2v_rate = np.log2(1 + np.divide(signal, self.V2V_Interference))
        m = min(rb, n)
        v2i_signals = self.V2I_power_dB - self.V2I_channels_abs[0:m] + self.vehAntGain + self.bsAntGain - self.bsNoiseFigure
        v2i_rate = np.log2(1 + np.divide(10 ** (v2i_signals / 10), self.V2I_Interference[0:m]))
        return v2v_rate, v2i_rate, interference

    def Compute_Interference(self, actions):
        """Environment.py:460-493.  In the reference the V2V->V2V part is unreachable (its `continue` fires for every
        valid channel, :486), so the observable result is noise + the co-channel V2I transmitter; kept that way."""
        n, nn, rb = len(self.vehicles), self.n_Neighbor, self.n_RB
        out = np.zeros((n, nn, rb)) + self.sig2
        if np.asarray(actions).ndim == 2:
            dest = np.array([[self.vehicles[k].destinations[m] for m in range(nn)] for k in range(n)], dtype=int)
            r = np.arange(rb)
            out += 10 ** ((self.V2I_power_dB - self.V2V_channels_with_fastfading[r[None, None, :], dest[:, :, None], r[None, None, :]]
                           + 2 * self.vehAntGain - self.vehNoiseFigure) / 10)
        self.V2V_Interference_all = 10 * np.log10(out)
