"""DQN agent glue around the Q-network: the counterpart of the reference `Memory` (BS_brain.py:245-270) and
`Agent` (BS_brain.py:280-910) for any number of V2V links (the reference hard-codes D1..D4), SURVEY.md 8f-1.

Same method names, argument meaning, return values and numpy-RNG consumption as the reference (including the
np.random.shuffle draw Keras' Model.fit makes per call, which the compact and device-resident paths reproduce), so a
seeded run with the same brain reproduces the reference's replay memory and the exact x / y payload it hands to `fit`
(pinned by tests/golden/golden_agent_n4.npz, captured from the reference agent on the real simulator).  What is
different: no per-sample Python loops (state packing, batch assembly and target construction are vectorised), and
when the brain exposes the compact entry points (`predict_arrays` / `fit_arrays`, v2xgnn.GnnQModel) the dense
`kron(Adj, I_F)` adjacency of the dict payload -- 6.5 MB per graph at 20 links x 64 features -- is never built.
"""
import datetime
import os

import numpy as np

from . import native_sim

MEMORY_CAPACITY = 1000000          # BS_brain.py:274
UPDATE_TARGET_FREQUENCY = 500      # :275
MAX_EPSILON = 1                    # :276
MIN_EPSILON = 0.01                 # :277
NATIVE_SAMPLER_MIN = 16384         # stored transitions from which Memory.sample's draw runs in libv2xsim.so (below: numpy is as fast)



def _random_channels(n, nn, n_ch):
    """The reference draws `np.random.choice(range(0, n_ch), nn)` once per link (BS_brain.py:322-324, :1015-1017): with
    replace=True and no probabilities that IS `randint(0, n_ch, nn)`, element by element on the process-wide numpy
    stream, so one call for all links consumes the stream identically (tests/test_rl_agent.py checks it) at 1/40 of the
    cost (172 -> 4.6 us for 20 links)."""
    return np.random.randint(0, n_ch, size=(n, nn))


class Memory(object):
    """FIFO replay memory of (s, a, r, s_) (BS_brain.py:245-270); per-instance storage (the reference's list is a
    class attribute shared by all instances, :246)."""

    def __init__(self, capacity):
        self.capacity = int(capacity)
        self.samples = []

    def add(self, sample):
        self.samples.append(sample)
        if len(self.samples) > self.capacity:
            self.samples.pop(0)

    def sample_indices(self, n, length=None):
        """n distinct positions when enough are stored, otherwise n draws with replacement -- the same global
        numpy RNG calls as the reference (:261, :268).  length: draw for a memory of that many samples (default: now)."""
        length = len(self.samples) if length is None else int(length)
        if length >= n:
            # numpy permutes the WHOLE memory for this (permutation(length)[:n]): 15-75 ms at the reference's capacity of 1e6
            # transitions.  The library runs the same shuffle on the same generator state -- same draws, same positions, same
            # state afterwards (tests/test_rl_agent.py) -- in 3-7 ms; below ~16k stored transitions numpy's call is as fast.
            if length >= NATIVE_SAMPLER_MIN and os.environ.get("V2X_RL_NATIVE_SAMPLER", "1") != "0":
                from . import native_sim
                if native_sim.available():
                    return native_sim.np_choice_noreplace(length, n)
            return np.random.choice(length, n, replace=False)
        # (one vectorised call consumes the legacy generator exactly like n scalar randint calls; checked in the tests)
        return np.random.randint(0, length, size=n)

    def sample(self, n):
        return [self.samples[i] for i in self.sample_indices(n)]


class Agent(object):
    def __init__(self, num_d2d, num_ch, num_neighbor, num_d2d_feedback, environment, curr_rl_config, brain=None,
                 device_replay='auto', rollouts='replicated', **brain_kwargs):
        """device_replay: keep the replay memory in HBM (rl/replay.py) and run replay() without the minibatch
        visiting the host; 'auto' = whenever the brain runs on the gfx950 engine.

        rollouts (data-parallel runs only; north_star: "RL rollouts from Environment.py are the natural shard"):
          'replicated'  every rank steps the SAME seeded simulator and draws the same minibatch, of which it fits its
                        contiguous share: bit-identical to the single-process run, but G GPUs do G identical rollouts;
          'sharded'     every rank owns a differently seeded simulator and its own replay memory, contributes
                        ceil(50 / G) of the transitions of a train step and samples its B / G graphs of the minibatch
                        from its OWN memory (stratified sampling: the minibatch is the union of the ranks' draws); the
                        Huber mean is over all B graphs and the gradients are all-reduced, so the replicas stay
                        bit-identical while the simulator work per rank drops by G.
                        This is NOT the reference's schedule transition for transition: when G does not divide 50 the job
                        collects ceil(50 / G) * G transitions per train step (56 at G = 8); the epsilon schedule and the
                        500-transition target sync count the JOB's transitions, but MEMORY_CAPACITY is per rank (the
                        job's replay memory is G times the reference's), and a rank only ever replays its own
                        simulator's trajectories.  'replicated' is the mode that reproduces the single-process run."""
        self.epsilon = MAX_EPSILON
        self.num_step = 0
        self.num_CH = num_ch
        self.num_D2D = num_d2d
        self.num_Neighbor = num_neighbor
        self.num_Feedback = num_d2d_feedback
        self.memory = Memory(MEMORY_CAPACITY)
        self.input_Node_Info = 3       # BS_brain.py:294
        self.input_Edge_Info = 1       # :295
        self.env = environment
        if brain is None:
            from ..bs_brain import BS
            brain = BS(self.num_D2D, self.input_Node_Info, self.input_Edge_Info, self.num_Feedback,
                       self.num_Neighbor, self.num_CH, **brain_kwargs)
        self.brain = brain
        self.num_States = self.brain.num_D2D_Input
        self.num_Actions = self.num_CH * self.num_Neighbor
        self.batch_size = curr_rl_config.Batch_Size
        self.gamma = curr_rl_config.Gamma
        self.v2v_weight = curr_rl_config.v2v_weight
        self.v2i_weight = curr_rl_config.v2i_weight
        self.num_Episodes, self.num_Train_Step, self.num_transition = 1, 1, 50
        if rollouts not in ('replicated', 'sharded'):
            raise ValueError("rollouts must be 'replicated' or 'sharded'")
        self.rollouts = rollouts
        self._sync_mark = 0
        engine = getattr(getattr(self.brain, 'model', None), 'engine', None)
        on_gpu = hasattr(engine, '_h') and self.num_Neighbor == 1
        self.device_replay = None
        # (the HBM-resident memory keeps a transition's adjacency as one 32-bit source mask per link: at most
        #  DeviceReplay.MAX_LINKS links; larger scenarios keep the reference's host-side Memory under 'auto')
        from .replay import DeviceReplay
        if device_replay is True and self.num_D2D > DeviceReplay.MAX_LINKS:
            raise ValueError("device_replay=True: at most %d links (got %d); use device_replay='auto' or False"
                             % (DeviceReplay.MAX_LINKS, self.num_D2D))
        if device_replay is True or (device_replay == 'auto' and on_gpu and self.num_D2D <= DeviceReplay.MAX_LINKS):
            if not on_gpu:
                raise ValueError("device_replay needs a brain on the gfx950 engine")
            self.device_replay = DeviceReplay(MEMORY_CAPACITY, self.num_D2D, device=engine.device)

    def _trainer(self):
        return getattr(getattr(self.brain, 'model', None), 'trainer', None)

    def _shard_world(self):
        """number of ranks the rollouts are sharded over (1 unless rollouts='sharded' under data parallelism)"""
        tr = self._trainer()
        return tr.world if (self.rollouts == 'sharded' and tr is not None) else 1

    # ------------------------------------------------------------------ observation
    def get_state(self, idx):
        """Normalised observation of link idx[0] towards its idx[1]-th receiver (BS_brain.py:389-407)."""
        Constant_A, Constant_B = 80, 60
        v2v = self.env.V2V_channels_with_fastfading
        dst = self.env.vehicles[idx[0]].destinations[idx[1]]
        V2V_channel = (v2v[idx[0], dst, :] - Constant_A) / Constant_B
        V2I_channel = (self.env.V2I_channels_with_fastfading[idx[0], :] - Constant_A) / Constant_B
        V2V_edge = (((np.sum(v2v[:, dst, :], axis=0) - v2v[dst, dst, :]) - (self.num_D2D - 1) * Constant_A) / Constant_B
                    - V2V_channel) / (self.num_D2D - 2)
        return V2V_channel, V2I_channel, V2V_edge

    def adjacency(self):
        """Adj[p, q] = 1 unless p == q or p is the receiver of link q (BS_brain.py:441-445)."""
        n = self.num_D2D
        adj = np.ones((n, n)) - np.eye(n)
        for q in range(n):
            adj[self.env.vehicles[q].destinations[0], q] = 0
        return adj

    def _batched(self):
        """the simulator is a BatchedEnviron (rl/batched_env.py): E environments stepped as arrays"""
        return hasattr(self.env, 'E')

    def _one_env(self):
        if self.env.E != 1:
            raise ValueError("this entry point steps ONE environment; the simulator holds %d" % self.env.E)

    def observe(self):
        """-> D2D_State [N, Dn+De] = [V2V gain x C | V2I gain x C | power | edge gain x C] (BS_brain.py:458-467)
        and the adjacency."""
        if self._batched():
            self._one_env()
            state, adj = self.env.observe(self.num_CH)
            return state[0], adj[0]
        n, C, nn = self.num_D2D, self.num_CH, self.num_Neighbor
        power = self.env.V2V_power_dB_List[self.env.fixed_v2v_power_index]
        state = np.zeros((n, self.brain.num_One_D2D_Input))
        if nn == 1:                                   # all links at once (same arithmetic as get_state)
            A, Bc = 80, 60
            v2v, v2i = self.env.V2V_channels_with_fastfading, self.env.V2I_channels_with_fastfading
            k = np.arange(n)
            dst = np.array([self.env.vehicles[i].destinations[0] for i in range(n)])
            ch = (v2v[k, dst, :] - A) / Bc
            edge = (((np.sum(v2v[:, dst, :], axis=0) - v2v[dst, dst, :]) - (n - 1) * A) / Bc - ch) / (n - 2)
            state[:, 0:C] = ch
            state[:, C:2 * C] = (v2i - A) / Bc
            state[:, 2 * C] = power
            state[:, 2 * C + 1:] = edge
            return state, self.adjacency()
        for k in range(n):
            ch, v2i, edge = np.zeros((nn, C)), None, np.zeros((nn, C))
            for m in range(nn):
                ch[m], v2i, edge[m] = self.get_state([k, m])
            state[k, 0:nn * C] = ch.reshape(-1)
            state[k, nn * C:2 * nn * C] = v2i
            state[k, 2 * nn * C:2 * nn * C + nn] = power
            state[k, 2 * nn * C + nn:] = edge.reshape(-1)
        return state, self.adjacency()

    # ------------------------------------------------------------------ brain I/O
    def _compact(self):
        return hasattr(getattr(self.brain, 'model', None), 'predict_arrays')

    def _feed(self, states, adj):
        """states [B, N, Dn+De], adj [B, N, N] -> the reference's dict payload (BS_brain.py:495-504, :642-651)."""
        B, n = states.shape[0], self.num_D2D
        dn, F = self.brain.num_One_Node_Input, self.brain.num_Feedback
        feed = {}
        for k in range(n):
            feed['D%d_Node_Input' % (k + 1)] = np.array(states[:, k, :dn])
            feed['D%d_Edge_Input' % (k + 1)] = np.array(states[:, k, dn:])
            feed['D%d_Neighbor_Input' % (k + 1)] = np.zeros((B, F))
        feed['Adjacency_Matrix'] = np.kron(adj, np.eye(F))
        return feed

    def _predict(self, states, adj, target=False):
        """-> Q [N, B, C]"""
        dn = self.brain.num_One_Node_Input
        if self._compact():
            model = self.brain.target_model if target else self.brain.model
            q = model.predict_arrays(states[:, :, :dn], states[:, :, dn:], adj)          # [B, N, C]
            return np.ascontiguousarray(np.transpose(q, (1, 0, 2)))
        return np.stack(self.brain.predict(self._feed(states, adj), target=target))

    # ------------------------------------------------------------------ acting
    def select_action_while_training(self, state):
        """epsilon-greedy (BS_brain.py:308-352); `state` = (D2D_State [N, .], Adj [N, N])."""
        n, nn = self.num_D2D, self.num_Neighbor
        steps = self.num_Episodes * 0.8 * self.num_Train_Step * self.num_transition
        per_step = (MAX_EPSILON - MIN_EPSILON) / steps
        self.epsilon = MAX_EPSILON - per_step * self.num_step if self.num_step < steps else MIN_EPSILON
        if np.random.random() < self.epsilon:
            return _random_channels(n, nn, self.num_CH)
        d2d_state, adj = state
        q = self._predict(d2d_state[None], adj[None])[:, 0, :]                           # [N, C]
        return np.argmax(q, axis=1).reshape(n, nn).astype(int)   # first maximiser, as np.where(...)[0][0] (:342-344)

    def select_action_random(self, state):
        return _random_channels(self.num_D2D, self.num_Neighbor, self.num_CH)

    def act(self, actions):
        """BS_brain.py:366-376"""
        self.num_step += 1
        if self._batched():
            self._one_env()
            v2v, v2i, intf = self.env.act(np.asarray(actions)[None])
            return v2v[0], v2i[0], intf[0]
        rates = self.env.compute_reward_with_channel_selection(actions)
        self.env.renew_positions()
        self.env.renew_channels_fastfading()
        self.env.Compute_Interference(actions)
        return rates

    def dump_act(self, actions):
        if self._batched():
            self._one_env()
            v2v, v2i, intf = self.env.compute_reward_with_channel_selection(np.asarray(actions)[None])
            return v2v[0], v2i[0], intf[0]
        return self.env.compute_reward_with_channel_selection(actions)

    def train_observe(self, sample):
        self.memory.add(sample)

    def generate_d2d_transition(self, num_transitions):
        """num_transitions environment steps under the epsilon-greedy policy into the replay memory
        (BS_brain.py:409-553).  A sample is [States(1, N*13+N*N), Actions(1, N), reward, States_]."""
        if self._batched():
            return self._generate_batched(num_transitions)
        rewards = np.zeros(num_transitions)
        for self.train_step in range(num_transitions):
            d2d_state, adj = self.observe()
            states = np.concatenate((d2d_state.reshape(1, -1), adj.reshape(1, -1)), axis=-1)
            action = self.select_action_while_training((d2d_state, adj))
            v2v_rate, v2i_rate, _ = self.act(action)
            reward = self.v2v_weight * np.sum(np.sum(v2v_rate, axis=1)) + self.v2i_weight * np.sum(v2i_rate)
            rewards[self.train_step] = reward
            next_state, _ = self.observe()
            states_ = np.concatenate((next_state.reshape(1, -1), adj.reshape(1, -1)), axis=-1)   # same adjacency (:547)
            if self.device_replay is not None:
                dn = self.brain.num_One_Node_Input
                self.device_replay.add(d2d_state[:, :dn], d2d_state[:, dn:], adj, action.reshape(-1), reward,
                                       next_state[:, :dn], next_state[:, dn:])
                self.train_observe(None)               # the host list only keeps the FIFO bookkeeping
            else:
                self.train_observe([states, action.reshape(1, -1), reward, states_])
        return rewards

    def _generate_batched(self, num_transitions):
        """generate_d2d_transition on a BatchedEnviron: every iteration observes all E environments, picks the E joint
        actions (epsilon-greedy per environment: one uniform draw each, in order; the greedy environments share ONE
        forward pass), steps the E simulators as arrays and stores E transitions.  ceil(num / E) iterations; with E = 1
        the numpy-RNG consumption and the stored transitions are those of the single-simulator loop."""
        E, n, nn, C = self.env.E, self.num_D2D, self.num_Neighbor, self.num_CH
        dn = self.brain.num_One_Node_Input
        n_iter = -(-num_transitions // E)
        rewards = np.zeros(n_iter * E)
        # observations straight in the engine's packed layout (no float64 state / dense adjacency on the way to the GPU or to
        # the replay memory) when the simulator offers them and the memory lives in HBM; V2X_RL_PACKED=0: the array path
        packed = self._packed_rollouts()
        if packed and E == 1 and self._native_rollout_ok():
            return self._rollout_one_simulator(num_transitions)
        for it in range(n_iter):
            if packed:
                rewards[it * E:(it + 1) * E] = self._packed_iteration(last=it == n_iter - 1)
                continue
            states, adj = self.env.observe(C)
            steps = self.num_Episodes * 0.8 * self.num_Train_Step * self.num_transition
            per_step = (MAX_EPSILON - MIN_EPSILON) / steps
            actions = np.zeros((E, n, nn), int)
            greedy = []
            for e in range(E):
                step_no = self.num_step + e
                self.epsilon = MAX_EPSILON - per_step * step_no if step_no < steps else MIN_EPSILON
                if np.random.random() < self.epsilon:
                    actions[e] = _random_channels(n, nn, C)
                else:
                    greedy.append(e)
            if greedy:
                q = self._predict(states[greedy], adj[greedy])                            # [N, len(greedy), C]
                actions[greedy] = np.transpose(np.argmax(q, axis=2))[:, :, None]
            v2v, v2i, _ = self.env.act(actions)
            self.num_step += E
            reward = self.v2v_weight * v2v.sum(axis=(1, 2)) + self.v2i_weight * v2i.sum(axis=1)
            rewards[it * E:(it + 1) * E] = reward
            nxt, _ = self.env.observe(C)
            if self.device_replay is not None:
                self.device_replay.add_many(states[:, :, :dn], states[:, :, dn:], adj, actions.reshape(E, -1), reward,
                                            nxt[:, :, :dn], nxt[:, :, dn:])
            for e in range(E):
                if self.device_replay is not None:
                    self.train_observe(None)
                else:
                    self.train_observe([np.concatenate((states[e].reshape(1, -1), adj[e].reshape(1, -1)), axis=-1),
                                        actions[e].reshape(1, -1), reward[e],
                                        np.concatenate((nxt[e].reshape(1, -1), adj[e].reshape(1, -1)), axis=-1)])
        return rewards[:num_transitions] if n_iter * E == num_transitions else rewards

    def _packed_iteration(self, last=False, force_greedy=False):
        """One iteration of _generate_batched on packed observations: same epsilon draws in the same order, same Q-values
        (the same kernels on the same float32 rows and CSR), same stored transitions.
        While the GPU is still busy with the previous fit (the predict below has to wait for it anyway) the host does what
        does not depend on the actions: the observation half of the transitions goes to its replay slots, and -- inside
        Agent.train, where the replay is known to follow this rollout -- the replay's minibatch is drawn and its slots
        uploaded: the numpy stream sees epsilon draws, minibatch draw, shuffle draw in the reference's order either way."""
        env, rep = self.env, self.device_replay
        E, n, C = env.E, self.num_D2D, self.num_CH
        xe, mask, col, regular = env.observe_packed(C)
        steps = self.num_Episodes * 0.8 * self.num_Train_Step * self.num_transition
        per_step = (MAX_EPSILON - MIN_EPSILON) / steps
        if (not force_greedy and E >= 4 and os.environ.get("V2X_RL_NATIVE_POLICY", "0") == "1" and native_sim.available()):
            # the E epsilon draws and the exploring simulators' randint draws in one library call on numpy's own generator state
            # (draw for draw the loop below).  OFF by default: measured at 50 simulators, batch 4096 -- 0.87-0.93 ms per train step
            # with it, 0.88 without: this part of the rollout runs while the GPU is still fitting and is not on the critical path
            actions, greedy, self.epsilon = native_sim.np_policy_draws(E, n, C, MAX_EPSILON, MIN_EPSILON, per_step, steps, self.num_step)
            greedy = list(greedy)
        else:
            actions = np.zeros((E, n, 1), int)
            greedy = []
            draw, base = np.random.random, self.num_step
            for e in range(E):
                step_no = base + e
                self.epsilon = MAX_EPSILON - per_step * step_no if step_no < steps else MIN_EPSILON
                if force_greedy:                       # (the native rollout took this transition's epsilon draw and found it greedy)
                    greedy.append(e)
                elif draw() < self.epsilon:
                    actions[e] = _random_channels(n, 1, C)
                else:
                    greedy.append(e)
        rep.stage_early(xe, col, mask)
        if last and getattr(self, '_predraw_ok', False):
            mem_len = min(self.memory.capacity, len(self.memory.samples) + E)
            n_draw = self.batch_size // (self.brain.model.trainer.world if self._shard_world() > 1 else 1)
            if (mem_len >= NATIVE_SAMPLER_MIN and mem_len >= n_draw and native_sim.available()
                    and os.environ.get("V2X_RL_NATIVE_SAMPLER", "1") != "0"):
                # a large memory: the permutation behind the draw takes 0.2-7 ms -- on a helper thread while this one scores,
                # acts and stores (nothing below draws from np.random; _replay_on_device collects the result first)
                self._predrawn = ("ahead", native_sim.ChoiceAhead(mem_len, n_draw), mem_len)
            else:
                idx = self._draw_replay_indices(mem_len)
                self._predrawn = (idx, rep.prefetch_indices(idx, E))
        if greedy:
            if regular.all():
                # ALL environments are scored, the exploring ones' rows are dropped: one batch shape for the whole run (one
                # captured launch instead of one per distinct number of greedy environments), and a graph's Q-values do not
                # depend on what else is in the batch (every kernel path gives the same bits, tests/test_gpu_shapes.py)
                q = self._predict_packed(xe, col)                                         # [E, N, C]
                actions[greedy] = np.argmax(q[greedy], axis=2)[:, :, None]
            else:                                      # a link that is its own receiver (rare): the general CSR builder
                states, adj = env.observe(C)
                q = self._predict(states[greedy], adj[greedy])
                actions[greedy] = np.transpose(np.argmax(q, axis=2))[:, :, None]
        # inside Agent.train the simulator step itself waits until the replay is enqueued (the next observe applies it): only the
        # rates and the next observation -- both ready -- are needed to store the transitions
        defer_step = getattr(self, '_predraw_ok', False) and hasattr(env, 'act_deferred')
        v2v, v2i, _ = env.act_deferred(actions) if defer_step else env.act(actions)
        self.num_step += E
        reward = self.v2v_weight * v2v.sum(axis=(1, 2)) + self.v2i_weight * v2i.sum(axis=1)
        xe_next = env.next_packed_observation(C) if defer_step else env.observe_packed(C)[0]
        rep.add_many_packed(xe, xe_next, col, mask, regular, actions.reshape(E, n), reward)
        samples = self.memory.samples                  # the host list only keeps the FIFO bookkeeping (train_observe(None) x E)
        samples.extend([None] * E)
        if len(samples) > self.memory.capacity:
            del samples[:len(samples) - self.memory.capacity]
        return reward

    def _native_rollout_ok(self):
        """ONE simulator on the native library and the gfx950 engine: the whole rollout is one library call (v2xsim_rollout)"""
        return (os.environ.get("V2X_RL_NATIVE_ROLLOUT", "1") != "0" and native_sim.available() and hasattr(self.env, 'native_rollout')
                and self.env._one_call_step() and self._rollout_closure(1) is not None)

    def _rollout_closure(self, graphs=1):
        """The predict of the native rollout for `graphs` graphs at once: page-locked buffers (the kernels read the observations over
        the bus, the library copies Q back: GnnEngine.forward_to_host's path) behind a v2x_forward_closure, and the address of
        v2x_forward_call.  A brain without the C ABI (tests) may provide `rollout_predict_callback(xe, col, q)` instead."""
        cache = self.__dict__.setdefault('_native_io', {})
        io = cache.get(graphs)
        if io is not None or graphs in cache:
            return io
        n, C = self.num_D2D, self.num_CH
        ne = n * (n - 2)
        custom = getattr(self.brain, 'rollout_predict_callback', None)
        if custom is not None:
            import ctypes
            xe, col, q = np.zeros((graphs * n, 16), np.float32), np.zeros(graphs * max(ne, 1), np.int32), np.zeros((graphs * n, C), np.float32)
            cb = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p)(lambda _ctx: int(custom(xe, col, q) or 0))
            io = cache[graphs] = {"xe": xe, "col": col, "q": q, "cb": cb, "fn": ctypes.cast(cb, ctypes.c_void_p).value, "ctx": None}
            return io
        engine = getattr(getattr(self.brain, 'model', None), 'engine', None)
        rep = self.device_replay
        cache[graphs] = None
        if engine is None or rep is None or not hasattr(engine, '_lib') or not hasattr(engine._lib, 'v2x_forward_call'):
            return None
        import ctypes
        from .. import lib as _lib
        torch = rep.torch
        pins = {"xe": torch.empty((graphs * n, 16), dtype=torch.float32).pin_memory(),
                "col": torch.empty(graphs * max(ne, 1), dtype=torch.int32).pin_memory(),
                "q": torch.empty((graphs * n, C), dtype=torch.float32).pin_memory()}
        if os.environ.get("V2X_RL_ZERO_COPY", "1") == "0" or any(engine._lib.v2x_device_addressable(pins[k_].data_ptr()) != 1 for k_ in ("xe", "col")):
            return None                                        # no unified addressing: the per-transition path copies
        if rep.n_edges is None:
            rep.n_edges = ne
        rp = rep.row_ptr(graphs)
        cl = _lib.ForwardClosure()
        cl.m = engine._h
        cl.b.n_graphs, cl.b.n_rows, cl.b.n_edges, cl.b.max_nodes, cl.b.max_edges, cl.b.on_device = graphs, graphs * n, graphs * ne, n, ne, 1
        cl.b.xe, cl.b.row_ptr, cl.b.col_idx = pins["xe"].data_ptr(), rp.data_ptr(), pins["col"].data_ptr()
        cl.q_out, cl.q_on_device = pins["q"].data_ptr(), 0
        io = cache[graphs] = {"pins": pins, "rp": rp, "closure": cl, "engine": engine,
                              "xe": pins["xe"].numpy(), "col": pins["col"].numpy(), "q": pins["q"].numpy(),
                              "fn": ctypes.cast(engine._lib.v2x_forward_call, ctypes.c_void_p).value, "ctx": ctypes.addressof(cl)}
        return io

    def _rollout_one_simulator(self, num_transitions):
        """generate_d2d_transition on ONE batched simulator as one library call: the reference's loop shape (50 sequential
        transitions, a B = 1 predict for every greedy one, BS_brain.py:409-553) with the simulator step cut over the library's
        threads and computed while the predict is in flight.  Same epsilon draws, same random actions, same Q-values, same
        stored transitions as _packed_iteration called num_transitions times (tests/test_rl_batched_env.py)."""
        env, rep = self.env, self.device_replay
        # ONE predict for all transitions of the rollout (their observations do not depend on the actions, the network does not
        # change before the replay): V2X_RL_ROLLOUT_BATCH_PREDICT=0 scores transition by transition, a B = 1 predict each
        batch = os.environ.get("V2X_RL_ROLLOUT_BATCH_PREDICT", "1") != "0"
        n = self.num_D2D
        rewards = np.zeros(num_transitions)
        steps = self.num_Episodes * 0.8 * self.num_Train_Step * self.num_transition
        per_step = (MAX_EPSILON - MIN_EPSILON) / steps
        done = 0
        while done < num_transitions:
            todo = num_transitions - done
            io = self._rollout_closure(todo) if batch else None
            use_batch = io is not None
            if io is None:
                io = self._rollout_closure(1)
            if "closure" in io:
                io["closure"].stream = io["engine"]._stream()
            out = env.native_rollout(todo, self.num_CH,
                                     dict(eps_max=MAX_EPSILON, eps_min=MIN_EPSILON, eps_per_step=per_step, eps_steps=steps, step_no0=self.num_step,
                                          predict=io["fn"], predict_ctx=io["ctx"], xe_pin=io["xe"], col_pin=io["col"], q_pin=io["q"],
                                          batch_predict=use_batch))
            k = out["done"]
            if k > 0:
                self.epsilon = out["eps_last"]
                self.num_step += k
                reward = self.v2v_weight * out["v2v_rate"].sum(axis=(1, 2)) + self.v2i_weight * out["v2i_rate"].sum(axis=1)
                rewards[done:done + k] = reward
                rep.add_many_packed(out["xe"], out["xe_next"], out["col"], out["mask"], out["regular"], out["action"], reward)
                samples = self.memory.samples
                samples.extend([None] * k)
                if len(samples) > self.memory.capacity:
                    del samples[:len(samples) - self.memory.capacity]
                done += k
            if done < num_transitions:
                if out["rc"] <= -1000:
                    raise RuntimeError("native rollout: the predict of transition %d failed" % done)
                # a graph with a link that is its own receiver needs the general CSR builder: this transition the slow way.
                # (its epsilon draw is already taken from np.random: the greedy branch, without drawing again)
                rewards[done] = self._packed_iteration(force_greedy=True)[0]
                done += 1
        return rewards

    def _packed_rollouts(self):
        """the batched rollout takes observations in the engine's packed layout (see _generate_batched)"""
        C = self.num_CH
        return (self.device_replay is not None and self.num_Neighbor == 1 and hasattr(self.env, 'observe_packed')
                and self.env.packed_ok(C) and self._compact()
                and self.brain.num_One_Node_Input + self.brain.num_One_Edge_Input == 3 * C + 1
                and os.environ.get("V2X_RL_PACKED", "1") != "0")

    def _warm_rollout_predict(self):
        """One throw-away predict of the current observation before a training run: the rollout predict's buffers, workspaces
        and captured launch exist before the first greedy step needs them (early steps explore, so a warm-up episode never
        scores anything).  Consumes no random draw and changes no state."""
        if self._batched() and self._packed_rollouts() and getattr(self.env, 'pos', None) is not None:
            xe, _, col, regular = self.env.observe_packed(self.num_CH)
            if regular.all():
                self._predict_packed(xe, col)

    def _predict_packed(self, xe, col):
        """Q-values [E, N, C] (float32, a view of a pinned buffer: valid until the next call) of all environments from their
        packed observations: two asynchronous copies in, one launch, one copy out, one synchronisation."""
        from ..engine import DeviceBatch
        rep, engine = self.device_replay, self.brain.model.engine
        torch = rep.torch
        E, n, C = xe.shape[0], self.num_D2D, self.num_CH
        ne = col.shape[1]
        io = getattr(self, '_rollout_io', None)
        if io is None or io["E"] != E:
            dev = rep.device
            if rep.n_edges is None:
                rep.n_edges = n * (n - 2)
            io = self._rollout_io = {
                "E": E, "zero_copy": os.environ.get("V2X_RL_ZERO_COPY", "1") != "0",
                "xe_pin": torch.empty((E * n, 16), dtype=torch.float32).pin_memory(), "col_pin": torch.empty(E * ne, dtype=torch.int32).pin_memory(),
                "q_pin": torch.empty((E * n, C), dtype=torch.float32).pin_memory(),
                "q_dev": torch.empty((E * n, C), dtype=torch.float32, device=dev)}
            if io["zero_copy"] and any(engine._lib.v2x_device_addressable(io[k_].data_ptr()) != 1 for k_ in ("xe_pin", "col_pin")):
                io["zero_copy"] = False                        # no unified addressing here: the copy path
            if io["zero_copy"]:
                # the batch descriptor points INTO the pinned buffers: the predict's kernels read the 136 KB of observations over
                # the bus themselves, the library copies Q back and synchronises -- no copy launches of ours, one call
                io["db"] = DeviceBatch.from_tensors(E, n, io["xe_pin"], rep.row_ptr(E), io["col_pin"], ne)
            else:
                io["db"] = DeviceBatch.from_tensors(E, n, torch.empty((E * n, 16), dtype=torch.float32, device=dev), rep.row_ptr(E),
                                                    torch.empty(E * ne, dtype=torch.int32, device=dev), ne)
            io["xe_np"] = io["xe_pin"].numpy().reshape(E, n, 16)
            io["col_np"] = io["col_pin"].numpy().reshape(E, ne)
            io["q_np"] = io["q_pin"].numpy().reshape(E, n, C)
        db = io["db"]
        np.copyto(io["xe_np"], xe)
        np.copyto(io["col_np"], col)
        if io["zero_copy"]:
            engine.forward_to_host(db, io["q_np"].reshape(E * n, C))
            return io["q_np"]
        db.xe.copy_(io["xe_pin"], non_blocking=True)
        db.col_idx.copy_(io["col_pin"], non_blocking=True)
        engine.forward(db, out=io["q_dev"])
        io["q_pin"].copy_(io["q_dev"], non_blocking=True)
        torch.cuda.current_stream(rep.device).synchronize()      # (polling an event instead: no difference, measured)
        return io["q_np"]

    # ------------------------------------------------------------------ learning
    def replay(self):
        """One DQN update on a replay minibatch (BS_brain.py:555-748): y = Q(s) with the taken action's entry
        replaced by r + gamma * max_a' Q_target(s', a')."""
        if self.device_replay is not None:
            return self._replay_on_device()
        n, d = self.num_D2D, self.brain.num_One_D2D_Input
        world = self._shard_world()
        if self.batch_size % world:
            raise ValueError("batch %d not divisible by %d ranks" % (self.batch_size, world))
        batch = self.memory.sample(self.batch_size // world)         # sharded rollouts: this rank's share, from its own memory
        B = len(batch)
        s = np.stack([b[0][0] for b in batch])
        s_ = np.stack([b[3][0] for b in batch])
        a = np.stack([b[1][0] for b in batch]).astype(int)                                # [B, N]
        r = np.array([b[2] for b in batch], dtype=np.float64)
        states, adj = s[:, :n * d].reshape(B, n, d), s[:, n * d:].reshape(B, n, n)
        states_ = s_[:, :n * d].reshape(B, n, d)
        p = self._predict(states, adj)                                                    # online   [N, B, C]
        p_ = self._predict(states_, adj, target=True)                                     # target   (adjacency reused, :583)
        # r + GAMMA * max(p_) scalar by scalar (BS_brain.py:690): under the reference's numpy-1.x stack python-float x
        # np.float32 is a float64 product, rounded to float32 once when stored into p -- evaluated in float64 here and in
        # k_dqn_targets alike (numpy >= 2 / NEP 50 would round the product to float32 first: <= 1 ulp apart)
        target = r[None, :] + self.gamma * np.max(p_, axis=2).astype(np.float64)          # [N, B]
        np.put_along_axis(p, np.transpose(a)[:, :, None], target[:, :, None].astype(p.dtype), axis=2)
        y = p.astype(np.float64)
        if self._compact():
            dn = self.brain.num_One_Node_Input
            kw = dict(presharded=True, n_global=self.batch_size) if world > 1 else {}
            result = self.brain.model.fit_arrays(states[:, :, :dn], states[:, :, dn:], adj, np.transpose(y, (1, 0, 2)), **kw)
        else:
            y_train = {'D%d_Decide_Output' % (k + 1): y[k] for k in range(n)}
            result = self.brain.train_dnn(self._feed(states, adj), y_train, self.batch_size)
        q_mean = np.sum(np.sum(y, axis=2) / self.num_Actions, axis=1) / self.batch_size
        q_max_mean = np.sum(np.max(y, axis=2), axis=1) / self.batch_size
        if world > 1:
            q_mean, q_max_mean = self._trainer().all_reduce_numpy(np.stack([q_mean, q_max_mean]))
        # the reference reads "original" Q statistics from p AFTER it was overwritten in place (:684-690, :743-746)
        return result, q_mean, q_max_mean, q_mean.copy(), q_max_mean.copy()

    def _draw_replay_indices(self, mem_len=None, ahead=None):
        """The replay's draws from the process-wide numpy stream, in the reference's order: the minibatch positions
        (Memory.sample, BS_brain.py:261/:268), then Model.fit's shuffle of the sample order.
        mem_len: the memory's length at the time of the replay when that is not now (the rollout draws ahead).
        ahead: a native_sim.ChoiceAhead started for exactly this draw (its result replaces Memory.sample_indices)."""
        B, model = self.batch_size, self.brain.model
        trainer = model.trainer
        if self._shard_world() > 1:                                # own memory, own draws: B / G graphs of the minibatch
            idx = ahead.result() if ahead is not None else self.memory.sample_indices(B // trainer.world, mem_len)
            model.consume_fit_shuffle(len(idx))
        else:
            idx = ahead.result() if ahead is not None else self.memory.sample_indices(B, mem_len)
            model.consume_fit_shuffle(B)       # same RNG stream as fit()
            if trainer is not None and trainer.world > 1:
                per = B // trainer.world
                idx = idx[trainer.rank * per:(trainer.rank + 1) * per]
        return idx

    def _replay_on_device(self, defer=False):
        """replay() with the minibatch gathered, scored, labelled and fitted in HBM.  Under data parallelism every
        rank draws the same indices and processes only its own contiguous share of them.
        defer: return (per-output losses, [Q mean; Q max mean]) as DEVICE tensors instead of the reference's tuple -- nothing
        in the step then waits for the GPU (train() reads a whole episode's statistics back in one copy)."""
        from ..bs_brain import History
        n, B = self.num_D2D, self.batch_size
        model, target = self.brain.model, self.brain.target_model
        trainer = model.trainer
        if trainer is not None and trainer.world > 1 and B % trainer.world:
            raise ValueError("batch %d not divisible by %d ranks" % (B, trainer.world))
        rep = self.device_replay
        pre, self._predrawn = getattr(self, '_predrawn', None), None
        if pre is not None and isinstance(pre[0], str):            # started by the rollout on a helper thread: collect it
            idx, pre = self._draw_replay_indices(pre[2], ahead=pre[1]), None
        elif pre is not None:                                      # drawn by the rollout right after its epsilon draws
            idx, pre = pre
        else:
            idx = self._draw_replay_indices()
        sb, sb_next, action, reward = rep.sample(idx, pre=pre)
        if trainer is None and hasattr(model.engine, 'dqn_step'):
            # single GPU: the whole step in one call, the online graph layers run once (v2x_dqn_step)
            y = rep.target_buffer(len(idx), self.num_CH)
            loss = model.engine.dqn_step(target.engine, sb, sb_next, action, reward, self.gamma, y_out=y)
        else:
            q = model.engine.forward(sb)                              # online  [k*n, C]
            q_next = target.engine.forward(sb_next)                   # target  (adjacency reused, :583)
            y = rep.dqn_targets(q, q_next, action, reward, self.gamma)
            loss = trainer.train_step(sb, y, n_graphs_global=B) if trainer is not None else model.engine.train_step(sb, y)
        # Q statistics of the fitted targets (BS_brain.py:743-746), summed in float64 on the device: sum of all entries and sum of
        # the per-sample maxima, per link.  Deferred on one GPU: the two divisions (by the number of actions, by the batch) are
        # the same IEEE operations on the host at the end of the episode -- three launches per step instead of seven.
        torch = rep.torch
        if hasattr(rep, 'q_stats') and y.is_cuda:
            # (one launch of the library instead of three reductions + elementwise launches of the host framework)
            st = rep.q_stats(y, len(idx), self.num_CH)
            s_all, s_max = st[0], st[1]
        else:
            y3 = y.view(len(idx), n, -1)
            s_all = y3.sum(dim=(0, 2), dtype=torch.float64)
            s_max = y3.amax(dim=2).sum(dim=0, dtype=torch.float64)
        multi = trainer is not None and trainer.world > 1
        if defer and hasattr(loss, 'cpu') and not multi:
            return loss, (s_all, s_max)
        stats = torch.stack([s_all / self.num_Actions, s_max])
        if multi:
            trainer.dist.all_reduce(stats, group=trainer.group)
        if defer and hasattr(loss, 'cpu'):
            return loss, stats / B
        stats = (stats / B).cpu().numpy()
        loss = np.asarray(loss.cpu().numpy() if hasattr(loss, 'cpu') else loss, np.float64)
        result = History()
        result.epoch.append(0)
        result.history['loss'] = [float(loss.sum())]
        for k, name in enumerate(model.output_names):
            result.history[name + '_loss'] = [float(loss[k])]
        return result, stats[0], stats[1], stats[0].copy(), stats[1].copy()

    def train(self, num_episodes, num_train_steps, save_dir=None, save_interval=5, verbose=False):
        """BS_brain.py:750-910 without the plotting / pickling: episodes x train steps x (50 transitions + 1 replay),
        target sync whenever num_step is a multiple of 500, weights saved every `save_interval` episodes."""
        self.num_Episodes, self.num_Train_Step = num_episodes, num_train_steps
        # Everything alive now (the imported frameworks, the engines, the simulator) goes to the collector's permanent
        # generation: a full collection inside a train step walked ~1e6 such objects -- 40-70 ms, once or twice per hundred
        # steps of a 3 ms loop (measured, tools/prof_rl_sections.py).  gc.freeze() alone is a list splice; a gc.collect() in front
        # of it (to keep garbage out of the permanent generation) is that same 40-70 ms walk and was 0.4-0.7 ms per step of a
        # 100-step run -- whatever garbage exists now simply waits for the unfreeze.  The freeze is a process-wide change of the
        # collector, so it ends with this call (gc.unfreeze in the finally below: cycles among the frozen objects become
        # collectable again, and repeated train() calls do not pile up frozen generations -- ADVICE r04); V2X_RL_GC_FREEZE=0
        # leaves the collector alone.
        frozen = os.environ.get("V2X_RL_GC_FREEZE", "1") != "0"
        if frozen:
            import gc
            gc.freeze()
        try:
            return self._train_loop(num_episodes, num_train_steps, save_dir, save_interval, verbose)
        finally:
            pre = getattr(self, '_predrawn', None)
            if pre is not None and isinstance(pre[0], str):        # a draw still running on the helper thread: np.random gets its state back
                pre[1].result()
            self._predraw_ok, self._predrawn = False, None
            if hasattr(self.env, 'finish_step'):
                self.env.finish_step()
            if frozen:
                gc.unfreeze()

    def _train_loop(self, num_episodes, num_train_steps, save_dir, save_interval, verbose):
        n = self.num_D2D
        world = self._shard_world()
        self.num_transition = -(-50 // world)          # sharded rollouts: this rank's share of the 50 transitions per step
        if self._batched():                            # batched simulator: whole iterations of E transitions
            self.num_transition = -(-self.num_transition // self.env.E) * self.env.E
        self._sync_mark = 0
        loss = np.ones((n, num_episodes, num_train_steps))
        q_mean, q_max = np.zeros_like(loss), np.zeros_like(loss)
        self.num_step = 0
        reward_step = np.zeros((num_episodes, num_train_steps, self.num_transition))
        reward_episode = np.zeros(num_episodes)
        # HBM-resident replay: the per-step losses and Q statistics stay on the device and are read back once per episode
        # (the reference reads them after every fit, BS_brain.py:835-845 -- two host synchronisations per train step here)
        defer = self.device_replay is not None and os.environ.get("V2X_RL_DEFER_STATS", "1") != "0"
        self._warm_rollout_predict()
        self._predraw_ok = defer                   # every rollout of this loop is followed by its replay (see _packed_iteration)
        for ep in range(num_episodes):
            self.env.new_random_game(self.num_D2D)
            pending = []
            for it in range(num_train_steps):
                reward_step[ep, it, :] = self.generate_d2d_transition(self.num_transition)
                out = self._replay_on_device(defer=True) if defer else self.replay()
                if defer and len(out) == 2:
                    pending.append(out)
                else:
                    # (a device replay that came back with host-side results -- no device tensors to defer -- IS this step's
                    #  replay: use it, do not run a second one; later steps go through replay() directly)
                    result, qm, qx, _, _ = out
                    defer = False
                    for k in range(n):
                        loss[k, ep, it] = result.history['D%d_Decide_Output_loss' % (k + 1)][0]
                    q_mean[:, ep, it], q_max[:, ep, it] = qm, qx
                # target sync whenever the job has collected another 500 transitions (BS_brain.py:846-847 tests
                # num_step % 500 once per train step; num_step advances by 50 per step there, so this is the same rule)
                mark = (self.num_step * world) // UPDATE_TARGET_FREQUENCY
                if mark > self._sync_mark:
                    self._sync_mark = mark
                    self.brain.update_target_model()
            if pending:
                torch = self.device_replay.torch
                lo = torch.stack([p[0].double() for p in pending]).cpu().numpy()            # [steps, n]
                loss[:, ep, :len(pending)] = lo.T
                if isinstance(pending[0][1], tuple):                                        # raw sums: divide here (see _replay_on_device)
                    s_all = torch.stack([p[1][0] for p in pending]).cpu().numpy()           # [steps, n]
                    s_max = torch.stack([p[1][1] for p in pending]).cpu().numpy()
                    q_mean[:, ep, :len(pending)] = (s_all / self.num_Actions / self.batch_size).T
                    q_max[:, ep, :len(pending)] = (s_max / self.batch_size).T
                else:
                    st = torch.stack([p[1] for p in pending]).cpu().numpy()                 # [steps, 2, n]
                    q_mean[:, ep, :len(pending)], q_max[:, ep, :len(pending)] = st[:, 0].T, st[:, 1].T
            reward_episode[ep] = np.sum(reward_step[ep])
            if verbose:
                print(datetime.datetime.now().strftime('%H:%M:%S'), 'episode', ep + 1, 'reward %.3f' % reward_episode[ep],
                      'loss', np.round(loss[:, ep].mean(axis=1), 4))
            if save_dir is not None and (ep + 1) % save_interval == 0:
                os.makedirs(save_dir, exist_ok=True)
                tag = '-Episode-%d-Step-%d-Batch-%d.h5' % (ep + 1, num_train_steps, self.batch_size)     # names of :859-868
                self.brain.model.save_weights(os.path.join(save_dir, 'Q-Network_model_weights' + tag))
                self.brain.target_model.save_weights(os.path.join(save_dir, 'Target-Network_model_weights' + tag))
        return loss, reward_step, reward_episode, q_mean, q_max, q_mean.copy(), q_max.copy()

    # ------------------------------------------------------------------ evaluation
    def generate_d2d_initial_states(self):
        """BS_brain.py:912-984: the dict payload of the current simulator state (B = 1)."""
        d2d_state, adj = self.observe()
        return self._feed(d2d_state[None], adj[None])

    def _joint_actions(self):
        """All C^N joint channel assignments, link 0 most significant (the digit extraction of BS_brain.py:1071-1078,
        which hard-codes 4^4; here any N with C^N <= 65536)."""
        n, C = self.num_D2D, self.num_CH
        if C ** n > 65536:
            raise ValueError("brute-force search over %d^%d joint actions is not feasible" % (C, n))
        import itertools
        return np.array(list(itertools.product(range(C), repeat=n)), int)

    def _brute_force(self, joint):
        """-> (index, reward, (V2V rates, V2I rates, interference)) of the best joint action for the CURRENT simulator
        state; np.argmax keeps the first maximiser like the reference (:1093, :1371)."""
        res = [self.dump_act(a.reshape(self.num_D2D, 1)) for a in joint]
        rewards = np.array([self.v2v_weight * np.sum(r[0]) + self.v2i_weight * np.sum(r[1]) for r in res])
        best = int(np.argmax(rewards))
        return best, float(rewards[best]), res[best]

    def test_run(self, num_episodes, num_test_step, opt_flag=False):
        """Evaluation loop (BS_brain.py:986-1162): greedy policy of the trained network vs the random-action baseline
        and, with opt_flag, the brute-force optimum over all C^N joint actions (the reference hard-codes 4^4,
        :1071-1078; here any N with C^N <= 65536).  Same return tuple as the reference: 15 arrays with opt_flag,
        10 without."""
        n, C = self.num_D2D, self.num_CH
        self.num_Episodes, self.num_Test_Step = num_episodes, num_test_step
        w_v2v, w_v2i = self.v2v_weight, self.v2i_weight

        def book():
            return [np.zeros(num_episodes), np.zeros((num_episodes, num_test_step)), np.zeros((num_episodes, num_test_step, n)),
                    np.zeros((num_episodes, num_test_step, C)), np.zeros((num_episodes, num_test_step, C))]

        def record(b, ep, st, v2v, v2i, interference):
            b[1][ep, st] = w_v2v * np.sum(v2v) + w_v2i * np.sum(v2i)
            b[0][ep] += b[1][ep, st]
            b[2][ep, st, :] = np.sum(v2v, axis=1)
            b[3][ep, st, :] = v2i
            b[4][ep, st, :] = interference

        rl, ra, opt = book(), book(), book()
        if opt_flag:
            joint = self._joint_actions()
        for ep in range(num_episodes):
            self.env.new_random_game(self.num_D2D)
            for st in range(num_test_step):
                record(ra, ep, st, *self.dump_act(self.select_action_random(None)))
                if opt_flag:
                    best, reward, res = self._brute_force(joint)
                    if reward > 0:                                                        # at least one feasible solution
                        record(opt, ep, st, *res)
                d2d_state, adj = self.observe()
                q = self._predict(d2d_state[None], adj[None])[:, 0, :]
                action = np.argmax(q, axis=1).reshape(n, self.num_Neighbor).astype(int)
                record(rl, ep, st, *self.act(action))
        return tuple(rl + ra + opt) if opt_flag else tuple(rl + ra)

    def checkpoint_dir(self, root=None):
        """Folder the training driver saves into and the evaluation loads from (BS_brain.py:1231-1236; the reference
        joins with '\\' and only works on Windows)."""
        name = 'Train-Result-RealFB-%s-Batch-%s-Gamma-%s-V2Iweight-%s' % (self.num_Feedback, self.batch_size, self.gamma, self.v2i_weight)
        return os.path.join(root if root is not None else os.getcwd(), name)

    def evaluate_training_diff_trials(self, num_episodes, num_test_step, opt_flag, fixed_epsilon, num_evaluate_trials,
                                      model_dir=None, num_train_steps=20, load=True):
        """Evaluation of the TRAINING PROCESS (BS_brain.py:1164-1451): for every saved checkpoint (one per 5 training
        episodes, :1218,:1228) and every trial, an episode under a FIXED epsilon-greedy policy (:1376-1397) next to the
        random-action baseline (:1330-1338) and -- per step of the first checkpoint -- the brute-force optimum (:1282-
        1328); with opt_flag also the optimum of every step (:1340-1374).  Trial t re-seeds the Python / numpy RNGs with
        t + 1 before every episode (:1262-1265).  Same return tuples as the reference: 9 arrays with opt_flag, 5 without.
        model_dir: checkpoint folder (default: checkpoint_dir()); load=False evaluates the weights already in the brain
        for every checkpoint (tests with a recording brain)."""
        import random
        n, C, nn = self.num_D2D, self.num_CH, self.num_Neighbor
        self.num_Episodes = int(num_episodes // 5)
        self.num_Test_Step = num_test_step
        n_ep, n_st, n_tr = self.num_Episodes, num_test_step, num_evaluate_trials
        w_v2v, w_v2i = self.v2v_weight, self.v2i_weight
        folder = model_dir if model_dir is not None else self.checkpoint_dir()
        ev_opt_return, ev_opt_reward = np.zeros(n_tr), np.zeros((n_tr, n_st))
        ra_return, ra_reward = np.zeros((n_tr, n_ep)), np.zeros((n_tr, n_ep, n_st))
        if opt_flag:
            opt_return, opt_reward = np.zeros((n_tr, n_ep)), np.zeros((n_tr, n_ep, n_st))
            opt_v2v, opt_v2i, opt_intf = np.zeros((n_tr, n_ep, n_st, n)), np.zeros((n_tr, n_ep, n_st, C)), np.zeros((n_tr, n_ep, n_st, C))
        ret, rew = np.zeros((n_tr, n_ep)), np.zeros((n_tr, n_ep, n_st))
        joint = self._joint_actions()
        for trial in range(n_tr):
            for ep in range(n_ep):
                if load:
                    tag = '-Episode-%d-Step-%d-Batch-%d.h5' % ((ep + 1) * 5, num_train_steps, self.batch_size)
                    self.brain.model.load_weights(os.path.join(folder, 'Q-Network_model_weights' + tag))
                    self.brain.target_model.load_weights(os.path.join(folder, 'Target-Network_model_weights' + tag))
                random.seed(trial + 1)
                np.random.seed(trial + 1)
                self.env.new_random_game(self.num_D2D)
                for st in range(n_st):
                    if ep == 0:                                  # ground truth once per trial (:1282)
                        _, reward, _ = self._brute_force(joint)
                        if reward > 0:
                            ev_opt_reward[trial, st] = reward
                            ev_opt_return[trial] += reward
                    v2v, v2i, _ = self.dump_act(self.select_action_random(None))
                    ra_reward[trial, ep, st] = w_v2v * np.sum(v2v) + w_v2i * np.sum(v2i)
                    ra_return[trial, ep] += ra_reward[trial, ep, st]
                    if opt_flag:
                        _, reward, (v2v, v2i, intf) = self._brute_force(joint)
                        if reward > 0:
                            opt_reward[trial, ep, st] = reward
                            opt_return[trial, ep] += reward
                            opt_v2v[trial, ep, st, :] = np.sum(v2v, axis=1)
                            opt_v2i[trial, ep, st, :] = v2i
                            opt_intf[trial, ep, st, :] = intf
                    if np.random.random() < fixed_epsilon:
                        action = _random_channels(n, nn, C).reshape(n, 1) if nn == 1 else _random_channels(n, nn, C)
                    else:
                        d2d_state, adj = self.observe()
                        q = self._predict(d2d_state[None], adj[None])[:, 0, :]
                        action = np.argmax(q, axis=1).reshape(n, nn).astype(int)
                    v2v, v2i, _ = self.act(action)
                    rew[trial, ep, st] = w_v2v * np.sum(v2v) + w_v2i * np.sum(v2i)
                    ret[trial, ep] += rew[trial, ep, st]
        if opt_flag:
            return ret, rew, ra_return, ra_reward, opt_return, opt_reward, opt_v2v, opt_v2i, opt_intf
        return ev_opt_return, ret, rew, ra_return, ra_reward
