"""Soak: many DQN train steps + predicts; device memory must stay flat (no leaked graphs / events / buffers)."""
import random, sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from v2xgnn.rl import RL_Config, Agent
from v2xgnn.rl.train import start_env
random.seed(3); np.random.seed(3)
cfg = RL_Config(); cfg.set_train_value(64, 0.5, 2048, 1, 0.1)
env = start_env(20)
with torch.cuda.stream(torch.cuda.Stream()):
    agent = Agent(20, 4, 1, 64, env, cfg, seed=1, use_graph=True)
    agent.num_Episodes, agent.num_Train_Step = 1, 200
    free0 = None
    for it in range(120):
        agent.generate_d2d_transition(50)
        agent.replay()
        if it % 7 == 0:
            agent.brain.update_target_model()
        if it in (20, 119):
            torch.cuda.synchronize()
            free, total = torch.cuda.mem_get_info()
            print("iter", it, "device memory in use MB", (total - free) / 2**20, flush=True)
            if free0 is None:
                free0 = free
    print("growth MB between iter 20 and 119:", (free0 - free) / 2**20)
