"""Wall-clock sections of the batched DQN loop (bench.py --workload cfg2loop --envs 50): every simulator / agent / engine call
wrapped with perf_counter, per-step totals, medians and the calls that took > 2 ms.  THREADS=n sets the simulator's OpenMP
threads.  (cProfile misattributes the OpenMP calls: it reported 1 ms for a 45 us mt_uniforms.)"""
import os, sys, time, random
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import v2xgnn
from v2xgnn.rl import Agent, RL_Config, native_sim
from v2xgnn.rl.train import start_env_batched
import torch
random.seed(1001); np.random.seed(1001)
cfg = RL_Config(); cfg.set_train_value(64, 0.5, 4096, 1, 0.1)
env = start_env_batched(20, 50, 1001)
agent = Agent(env.n_Veh, env.n_RB, env.n_Neighbor, 64, env, cfg, seed=1001, device=0, use_graph=True)
_ctx = torch.cuda.stream(torch.cuda.Stream()) if os.environ.get("STREAM", "1") != "0" else None     # hipGraphs need a non-default stream (as bench.py)
if _ctx is not None: _ctx.__enter__()
agent.train(1, 2)
if os.environ.get('THREADS'): native_sim.set_threads(int(os.environ['THREADS']))
rec = {}
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); rec.setdefault(label, []).append(time.perf_counter() - t0); return r
    setattr(obj, name, g)
for nm in ("mt_uniforms", "channels", "observe", "reward", "interference_db", "advance", "advance_start", "advance_wait", "observe_packed"):
    wrap(native_sim, nm, "native." + nm)
for nm in ("observe", "act", "new_random_game", "renew_positions", "renew_channels_fastfading", "renew_channel", "renew_neighbor", "_advance_channels", "act_for_training"):
    if hasattr(env, nm): wrap(env, nm, "env." + nm)
for nm in ("act_deferred", "next_packed_observation", "finish_step", "observe_packed"):
    wrap(env, nm, "env." + nm)
wrap(agent, "_draw_replay_indices"); wrap(agent.device_replay, "stage_early"); wrap(agent.device_replay, "prefetch_indices")
wrap(agent.device_replay, "_upload_indices"); wrap(agent.device_replay, "_gather")
_sync = torch.cuda.Stream.synchronize
def _sync_timed(self):
    t0 = time.perf_counter(); r = _sync(self); rec.setdefault("stream.synchronize", []).append(time.perf_counter() - t0); return r
torch.cuda.Stream.synchronize = _sync_timed
_copy = torch.Tensor.copy_
def _copy_timed(self, *a, **k):
    t0 = time.perf_counter(); r = _copy(self, *a, **k); rec.setdefault("tensor.copy_", []).append(time.perf_counter() - t0); return r
torch.Tensor.copy_ = _copy_timed
_copyto = np.copyto
def _copyto_timed(*a, **k):
    t0 = time.perf_counter(); r = _copyto(*a, **k); rec.setdefault("np.copyto", []).append(time.perf_counter() - t0); return r
np.copyto = _copyto_timed
wrap(env, "_advance", "env._advance"); wrap(env, "_start_job", "env._start_job")
wrap(agent, "_packed_iteration"); wrap(agent, "_predict_packed"); wrap(agent.device_replay, "add_many_packed"); wrap(agent.device_replay, "flush")
wrap(agent.brain, "update_target_model"); wrap(agent.brain.model, "consume_fit_shuffle")
_cpu = torch.Tensor.cpu
def _cpu_timed(self, *a, **k):
    t0 = time.perf_counter(); r = _cpu(self, *a, **k); rec.setdefault("tensor.cpu()", []).append(time.perf_counter() - t0); return r
torch.Tensor.cpu = _cpu_timed
wrap(agent, "_predict"); wrap(agent, "_generate_batched"); wrap(agent, "_replay_on_device"); wrap(agent, "train_observe")
wrap(agent.device_replay, "add_many"); wrap(agent.device_replay, "sample"); wrap(agent.memory, "sample_indices")
wrap(agent.brain.model.engine, "dqn_step"); wrap(agent.brain.model.engine, "forward", "engine.forward")
torch.cuda.synchronize(); t0 = time.perf_counter()
agent.train(5, 20)
torch.cuda.synchronize(); wall = time.perf_counter() - t0
print("THREADS", os.environ.get('THREADS'), "ms per train step %.3f" % (1e3 * wall / 100))
for k, v in sorted(rec.items(), key=lambda kv: -sum(kv[1])):
    print("%-28s calls %4d  total/step %.3f ms  median %.0f us" % (k, len(v), 1e3 * sum(v) / 100, 1e6 * np.median(v)))
print("memory len:", len(agent.memory.samples) if hasattr(agent.memory, 'samples') else None, "capacity", getattr(agent.memory, 'capacity', None))
for k, v in rec.items():
    big = [(i, round(1e3 * t, 2)) for i, t in enumerate(v) if t > 2e-3]
    if big: print("outliers (call index, ms):", k, big)
