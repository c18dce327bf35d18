import cProfile, pstats, random, sys, time
import numpy as np
sys.path.insert(0, '.')
from v2xgnn.rl import RL_Config, Agent
from v2xgnn.rl.train import start_env
random.seed(1001); np.random.seed(1001)
cfg = RL_Config(); cfg.set_train_value(64, 0.5, 4096, 1, 0.1)
env = start_env(20)
agent = Agent(20, 4, 1, 64, env, cfg, seed=1, use_graph=True)
agent.num_Episodes, agent.num_Train_Step = 1, 30
for _ in range(3):
    agent.generate_d2d_transition(50); agent.replay()
import torch
ts = []
for _ in range(10):
    agent.generate_d2d_transition(50)
    torch.cuda.synchronize(); t = time.perf_counter()
    agent.replay()
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
print("replay ms:", np.round(1e3 * np.array(ts), 2))
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    agent.generate_d2d_transition(50); agent.replay()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
