"""Where a replay step (Agent.replay with the HBM-resident memory, BS_brain.py:555-748) spends its wall time: cProfile of 200
replays on a filled memory, no rollouts in between."""
import cProfile, pstats, random, sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from v2xgnn.rl import RL_Config, Agent
from v2xgnn.rl.train import start_env_batched
random.seed(1001); np.random.seed(1001)
cfg = RL_Config(); cfg.set_train_value(64, 0.5, 4096, 1, 0.1)
env = start_env_batched(20, 50, 1001)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    agent = Agent(20, 4, 1, 64, env, cfg, seed=1, use_graph=True)
    agent.num_Episodes, agent.num_Train_Step = 1, 30
    for _ in range(6):
        agent.generate_d2d_transition(50); agent.replay()
    ts = []
    for _ in range(50):
        torch.cuda.synchronize(); t = time.perf_counter()
        agent.replay()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print("replay ms (no new transitions): median %.3f  min %.3f" % (1e3 * np.median(ts), 1e3 * min(ts)))
    ts = []
    for _ in range(20):
        agent.generate_d2d_transition(50)
        torch.cuda.synchronize(); t = time.perf_counter()
        agent.replay()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print("replay ms (50 new transitions each): median %.3f  min %.3f" % (1e3 * np.median(ts), 1e3 * min(ts)))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200):
        agent.replay()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(22)
