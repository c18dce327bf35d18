#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the reference's OWN Python code.

Runs ONLY in the build container (needs /root/reference); nothing here ships to the GPU box
except the .npz files it writes.  Usage:  python tests/golden/make_golden.py

The reference's arithmetic lives in Keras 2.2.4 / TF 1.14 (not installable here), so the
reference modules are imported behind a small *numpy-eager* stand-in for the Keras/TF
LIBRARY entry points the hot path calls (K.dot, K.bias_add, K.concatenate, K.batch_dot,
Input, Dense, Layer, concatenate, Model, Adam, tf.losses.huber_loss), each implementing the
documented semantics listed in SURVEY.md Appendix B.  Everything that is the reference's own
code runs unmodified:
  * `BS._create_model` wiring, `GNNLayer.build/call`, `AggLayer.call`  (BS_brain.py:17-216)
      -> golden_forward_n4.npz : inputs, every weight in creation order, the 4 outputs
  * `Agent.generate_d2d_initial_states`, `Agent.generate_d2d_transition`, `Agent.replay`
    on the real `Environment.Environ`                                  (BS_brain.py:389-748)
      -> golden_agent_n4.npz   : state dicts, replay samples, the x / y payload handed to fit

The fixtures hold DATA only (arrays); no reference source text is stored.
"""
import os
import sys
import types
import random

import numpy as np

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------- numpy-eager stub
class _State:
    feed = {}          # name -> ndarray returned by Input(name=...)
    weights = []       # (layer_name, weight_name, ndarray) in creation order
    rng = None
    counters = {}
    nonzero_bias = True


def _auto_name(cls):
    n = _State.counters.get(cls, 0) + 1
    _State.counters[cls] = n
    return '%s_%d' % (cls, n)


def _act(name):
    if name is None or name == 'linear':
        return lambda t: t
    if name == 'relu':
        return lambda t: np.maximum(t, 0)
    raise ValueError(name)


class Layer(object):
    def __init__(self, name=None, **kwargs):
        self.name = name or _auto_name(type(self).__name__.lower())
        self.built = False

    def add_weight(self, name, shape, initializer, trainable=True):
        shape = tuple(int(s) for s in shape)
        if initializer == 'glorot_uniform':
            lim = np.sqrt(6.0 / (shape[-2] + shape[-1]))
            w = _State.rng.uniform(-lim, lim, size=shape)
        elif initializer == 'zeros':
            # non-zero on purpose: parity is defined on injected weights, and a zero bias
            # would hide bias-path mistakes
            w = (_State.rng.uniform(-0.1, 0.1, size=shape) if _State.nonzero_bias
                 else np.zeros(shape))
        else:
            raise ValueError(initializer)
        _State.weights.append((self.name, name, w))
        return w

    def build(self, input_shape):
        self.built = True

    def __call__(self, x):
        if not self.built:
            if isinstance(x, list):
                shape = [(None,) + tuple(t.shape[1:]) for t in x]
            else:
                shape = (None,) + tuple(x.shape[1:])
            self.build(shape)
            self.built = True
        return self.call(x)


class Dense(Layer):
    def __init__(self, units, activation=None, name=None, **kw):
        super(Dense, self).__init__(name=name)
        self.units = units
        self.activation = _act(activation)

    def build(self, input_shape):
        self.kernel = self.add_weight('kernel', (input_shape[-1], self.units), 'glorot_uniform')
        self.bias = self.add_weight('bias', (self.units,), 'zeros')

    def call(self, x):
        return self.activation(np.dot(x, self.kernel) + self.bias)


def Input(shape, name=None):
    v = _State.feed[name]
    assert tuple(v.shape[1:]) == tuple(shape), (name, v.shape, shape)
    return v


def _concatenate(tensors, axis=-1, name=None):
    return np.concatenate(tensors, axis=axis)


class Model(object):
    def __init__(self, inputs, outputs):
        self.inputs, self.outputs = inputs, outputs

    def compile(self, optimizer=None, loss=None):
        self.loss = loss


def _batch_dot(x, y, axes):
    assert list(axes) == [1, 1] and x.ndim == 2 and y.ndim == 3
    return np.einsum('bi,bij->bj', x, y)


def _huber(labels, predictions, delta=1.0):
    err = predictions - labels
    a = np.abs(err)
    quad = np.minimum(a, delta)
    return np.mean(0.5 * quad ** 2 + delta * (a - quad))


def install_stubs():
    keras = types.ModuleType('keras')
    layers = types.ModuleType('keras.layers')
    models = types.ModuleType('keras.models')
    backend = types.ModuleType('keras.backend')
    optimizers = types.ModuleType('keras.optimizers')
    activations = types.SimpleNamespace(get=_act)
    layers.Input, layers.Dense, layers.Layer = Input, Dense, Layer
    layers.activations = activations
    layers.Lambda = lambda *a, **k: None
    layers.add = lambda ts: sum(ts)
    layers.concatenate = _concatenate
    models.Model = Model
    backend.dot = np.dot
    backend.bias_add = lambda x, b, data_format=None: x + b
    backend.concatenate = lambda ts, axis=-1: np.concatenate(ts, axis=axis)
    backend.batch_dot = _batch_dot
    optimizers.Adam = lambda **kw: ('adam', kw)
    keras.layers, keras.models, keras.backend, keras.optimizers = layers, models, backend, optimizers
    tf = types.ModuleType('tensorflow')
    tf.losses = types.SimpleNamespace(huber_loss=_huber)
    tf.set_random_seed = lambda s: None
    for n, m in [('keras', keras), ('keras.layers', layers), ('keras.models', models),
                 ('keras.backend', backend), ('keras.optimizers', optimizers), ('tensorflow', tf)]:
        sys.modules[n] = m
    if not hasattr(np, 'int'):
        np.int = int      # BS_brain.py:352,364 use the removed alias


# --------------------------------------------------------------------------- capture
def make_env(Environment):
    # lane constants: RL_Train_main.py:82-88
    up = [3.5 / 2, 3.5 / 2 + 3.5, 250 + 3.5 / 2, 250 + 3.5 + 3.5 / 2, 500 + 3.5 / 2, 500 + 3.5 + 3.5 / 2]
    down = [250 - 3.5 - 3.5 / 2, 250 - 3.5 / 2, 500 - 3.5 - 3.5 / 2, 500 - 3.5 / 2, 750 - 3.5 - 3.5 / 2, 750 - 3.5 / 2]
    left = [3.5 / 2, 3.5 / 2 + 3.5, 433 + 3.5 / 2, 433 + 3.5 + 3.5 / 2, 866 + 3.5 / 2, 866 + 3.5 + 3.5 / 2]
    right = [433 - 3.5 - 3.5 / 2, 433 - 3.5 / 2, 866 - 3.5 - 3.5 / 2, 866 - 3.5 / 2, 1299 - 3.5 - 3.5 / 2, 1299 - 3.5 / 2]
    env = Environment.Environ(down, up, left, right, 750, 1299)
    env.new_random_game(env.n_Veh)
    return env


class FakeBS(object):
    """Stands in for BS while capturing the CALLER's payloads:
    exposes the size attributes of BS_brain.py:95-104 and records predict / train_dnn I/O."""
    log = None

    def __init__(self, num_d2d, input_node_info, input_edge_info, num_d2d_feedback, num_d2d_neighbor, num_ch):
        self.num_D2D, self.num_Neighbor, self.num_CH = num_d2d, num_d2d_neighbor, num_ch
        self.num_Feedback = num_d2d_feedback
        self.input_node_Info, self.input_edge_Info = input_node_info, input_edge_info
        self.num_One_Node_Input = ((input_node_info - 1) * num_ch + 1) * num_d2d_neighbor
        self.num_One_Edge_Input = input_edge_info * num_ch
        self.num_One_D2D_Input = self.num_One_Node_Input + self.num_One_Edge_Input
        self.num_D2D_Input = num_d2d * self.num_One_D2D_Input + num_d2d ** 2
        self.prng = np.random.RandomState(77)   # private: must not touch the global RNG stream
        FakeBS.log = {'predict': [], 'fit': []}

    def predict(self, data, target=False):
        B = data['D1_Node_Input'].shape[0]
        out = [self.prng.normal(2.0, 1.0, size=(B, self.num_CH)).astype(np.float32)
               for _ in range(self.num_D2D)]
        FakeBS.log['predict'].append(({k: v.copy() for k, v in data.items()}, bool(target),
                                      [o.copy() for o in out]))
        return out

    predict_one_step = predict

    def train_dnn(self, x, y, batch_size):
        FakeBS.log['fit'].append(({k: v.copy() for k, v in x.items()},
                                  {k: v.copy() for k, v in y.items()}, batch_size))
        hist = types.SimpleNamespace(history={('D%d_Decide_Output_loss' % (k + 1)): [0.0]
                                              for k in range(self.num_D2D)})
        return hist


def main():
    install_stubs()
    sys.path.insert(0, REF)
    import Environment
    import Sim_Config
    import BS_brain

    RealBS = BS_brain.BS

    # ---- (1) caller payloads: Agent on the real Environ with a recording fake brain
    seed = 1001                                              # RL_Train_main.py:44
    random.seed(seed)
    np.random.seed(seed)
    cfg = Sim_Config.RL_Config()
    cfg.set_train_value(16, 0.5, 32, 1, 0.1)                 # FB=16, gamma=.5 (RL_Train_main.py:29-36); small batch
    env = make_env(Environment)
    BS_brain.BS = FakeBS
    BS_brain.Memory.samples = []                             # class attribute (BS_brain.py:246)
    agent = BS_brain.Agent(env.n_Veh, env.n_RB, env.n_Neighbor, cfg.Num_Feedback, env, cfg)
    agent.num_Episodes, agent.num_Train_Step, agent.num_transition = 10, 20, 50
    dest0 = np.array([v.destinations[0] for v in env.vehicles])
    init_states = agent.generate_d2d_initial_states()
    agent.num_step = 10 ** 9                                 # epsilon -> MIN_EPSILON: mostly greedy => predict path
    rewards = agent.generate_d2d_transition(24)              # 24 < batch 32 => with-replacement sampling branch
    n_pred_rollout = len(FakeBS.log['predict'])
    replay_out = agent.replay()
    samples = BS_brain.Memory.samples
    (states, tgt0, p), (states_, tgt1, p_) = FakeBS.log['predict'][n_pred_rollout:n_pred_rollout + 2]
    assert (tgt0, tgt1) == (False, True)
    fit_x, fit_y, fit_bs = FakeBS.log['fit'][0]

    agent_fx = {'seed': seed, 'gamma': cfg.Gamma, 'batch_size': cfg.Batch_Size,
                'destinations': dest0, 'rollout_rewards': rewards,
                'mem_states': np.stack([s[0][0] for s in samples]),
                'mem_actions': np.stack([s[1][0] for s in samples]),
                'mem_rewards': np.array([s[2] for s in samples]),
                'mem_states_next': np.stack([s[3][0] for s in samples]),
                'q_mean': replay_out[1], 'q_max_mean': replay_out[2]}
    for k, v in init_states.items():
        agent_fx['init/' + k] = v
    for k, v in states.items():
        agent_fx['replay_s/' + k] = v
    for k, v in states_.items():
        agent_fx['replay_s_next/' + k] = v
    for i in range(4):
        agent_fx['replay_p/%d' % i] = p[i]         # online prediction BEFORE the caller mutates it
        agent_fx['replay_p_next/%d' % i] = p_[i]
    for k, v in fit_x.items():
        agent_fx['fit_x/' + k] = v
    for k, v in fit_y.items():
        agent_fx['fit_y/' + k] = v
    np.savez_compressed(os.path.join(OUT, 'golden_agent_n4.npz'), **agent_fx)

    # ---- (2) forward goldens: the reference's own _create_model / GNNLayer / AggLayer code
    BS_brain.BS = RealBS
    fwd = {}
    cases = {
        'env_b1': init_states,                     # real simulator state, B=1 (predict_one_step path)
        'replay_b32': fit_x,                       # real replay minibatch, B=32
    }
    # synthetic case with NON-ZERO neighbor inputs and a generic (non-reference) adjacency
    r = np.random.RandomState(5)
    Bs, N, F = 6, 4, 16
    syn = {}
    for k in range(N):
        syn['D%d_Node_Input' % (k + 1)] = r.normal(0.7, 0.4, size=(Bs, 9))
        syn['D%d_Edge_Input' % (k + 1)] = r.normal(0.9, 0.2, size=(Bs, 4))
        syn['D%d_Neighbor_Input' % (k + 1)] = r.normal(0.0, 0.5, size=(Bs, F))
    adj = (r.uniform(size=(Bs, N, N)) < 0.5).astype(np.float64)
    syn['Adjacency_Matrix'] = np.kron(adj, np.eye(F))
    cases['synthetic_b6'] = syn

    for cname, feed in cases.items():
        _State.feed = feed
        _State.weights = []
        _State.counters = {}
        _State.rng = np.random.RandomState({'env_b1': 11, 'replay_b32': 12, 'synthetic_b6': 13}[cname])
        brain = RealBS(4, 3, 1, 16, 1, 4)           # builds online + target model (BS_brain.py:105-106)
        per_model = len(_State.weights) // 2
        for k, v in feed.items():
            fwd['%s/in/%s' % (cname, k)] = np.asarray(v, np.float64)
        for mi, (model, tag) in enumerate([(brain.model, 'online'), (brain.target_model, 'target')]):
            pre = '%s/%s/' % (cname, tag)
            ws = _State.weights[mi * per_model:(mi + 1) * per_model]
            fwd[pre + 'weight_names'] = np.array(['%s:%s' % (ln, wn) for ln, wn, _ in ws])
            for i, (_, _, w) in enumerate(ws):
                fwd[pre + 'w/%03d' % i] = w
            for k, o in enumerate(model.outputs):
                fwd[pre + 'out/%d' % k] = o
    np.savez_compressed(os.path.join(OUT, 'golden_forward_n4.npz'), **fwd)

    # ---- (3) simulator trajectories: the reference Environ stepped with seeded random actions
    #          (Environment.py:179-506; the call order per step is Agent.act, BS_brain.py:366-376)
    for n_veh, steps in ((4, 40), (20, 12)):
        random.seed(2020 + n_veh)
        np.random.seed(2020 + n_veh)
        env = make_env(Environment)
        if n_veh != env.n_Veh:
            env.new_random_game(n_veh)
        tr = {'n_veh': n_veh, 'steps': steps,
              'init_pos': np.array([v.position for v in env.vehicles], float),
              'init_dir': np.array([v.direction for v in env.vehicles]),
              'velocity': np.array([v.velocity for v in env.vehicles], float),
              'init_dest': np.array([v.destinations[0] for v in env.vehicles]),
              'init_v2v': env.V2V_channels_with_fastfading.copy(),
              'init_v2i': env.V2I_channels_with_fastfading.copy()}
        pos, dirs, v2v, v2i, acts, r_v2v, r_v2i, intf, v2v_int_all = [], [], [], [], [], [], [], [], []
        for t in range(steps):
            a = np.random.randint(0, env.n_RB, size=(n_veh, 1))
            acts.append(a.copy())
            V2V_Rate, V2I_Rate, Interference = env.compute_reward_with_channel_selection(a.copy())
            r_v2v.append(V2V_Rate.copy()); r_v2i.append(V2I_Rate.copy()); intf.append(Interference.copy())
            env.renew_positions()
            env.renew_channels_fastfading()
            env.Compute_Interference(a.copy())
            pos.append(np.array([v.position for v in env.vehicles], float))
            dirs.append(np.array([v.direction for v in env.vehicles]))
            v2v.append(env.V2V_channels_with_fastfading.copy())
            v2i.append(env.V2I_channels_with_fastfading.copy())
            v2v_int_all.append(env.V2V_Interference_all.copy())
        tr.update(pos=np.stack(pos), dirs=np.stack(dirs), v2v=np.stack(v2v), v2i=np.stack(v2i), actions=np.stack(acts),
                  v2v_rate=np.stack(r_v2v), v2i_rate=np.stack(r_v2i), interference=np.stack(intf),
                  v2v_interference_all=np.stack(v2v_int_all))
        np.savez_compressed(os.path.join(OUT, 'golden_env_n%d.npz' % n_veh), **tr)

    # ---- (4) mobility at crossings and exits: vehicles are re-placed just before a crossing lane / the map
    #          border (seeded), then the reference renew_positions runs; pins the turn logic and its RNG use
    random.seed(77)
    env = make_env(Environment)
    pr = np.random.RandomState(9)
    rounds, per = 60, 3
    place_pos, place_dir, out_pos, out_dir = [], [], [], []
    for rd in range(rounds):
        pp, pd = [], []
        for v in env.vehicles:
            d = ['u', 'd', 'l', 'r'][pr.randint(4)]
            if pr.uniform() < 0.25:                     # near the border: exercises the exit handling
                if d == 'u': pos = [env.up_lanes[pr.randint(6)], env.height - 0.05]
                elif d == 'd': pos = [env.down_lanes[pr.randint(6)], 0.05]
                elif d == 'l': pos = [0.05, env.left_lanes[pr.randint(6)]]
                else: pos = [env.width - 0.05, env.right_lanes[pr.randint(6)]]
            else:                                        # just before a crossing lane
                if d in ('u', 'd'):
                    lanes = env.left_lanes + env.right_lanes
                    y = lanes[pr.randint(len(lanes))] + (-0.06 if d == 'u' else 0.06)
                    pos = [(env.up_lanes if d == 'u' else env.down_lanes)[pr.randint(6)], y]
                else:
                    lanes = env.up_lanes + env.down_lanes
                    x = lanes[pr.randint(len(lanes))] + (0.06 if d == 'l' else -0.06)
                    pos = [x, (env.left_lanes if d == 'l' else env.right_lanes)[pr.randint(6)]]
            v.position, v.direction = list(pos), d
            pp.append(list(pos)); pd.append(d)
        place_pos.append(pp); place_dir.append(pd)
        for _ in range(per):
            env.renew_positions()
            out_pos.append([list(v.position) for v in env.vehicles])
            out_dir.append([v.direction for v in env.vehicles])
    np.savez_compressed(os.path.join(OUT, 'golden_env_cross.npz'), velocity=np.array([v.velocity for v in env.vehicles], float),
                        place_pos=np.array(place_pos, float), place_dir=np.array(place_dir), per=per,
                        out_pos=np.array(out_pos, float), out_dir=np.array(out_dir))
    print('wrote', sorted(os.listdir(OUT)))
    print('weights per model:', per_model)


if __name__ == '__main__':
    main()
