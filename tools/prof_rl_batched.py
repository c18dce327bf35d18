"""cProfile of the batched DQN loop (bench.py --workload cfg2loop --envs 50): where the host time of a train step goes."""
import cProfile
import os
import pstats
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import v2xgnn  # noqa: E402,F401
from v2xgnn.rl import Agent, RL_Config  # noqa: E402
from v2xgnn.rl.train import start_env_batched  # noqa: E402
import torch  # noqa: E402

links, feat, batch, envs = 20, 64, 4096, int(os.environ.get("ENVS", "50"))
random.seed(1001)
np.random.seed(1001)
cfg = RL_Config()
cfg.set_train_value(feat, 0.5, batch, 1, 0.1)
env = start_env_batched(links, envs, 1001)
agent = Agent(env.n_Veh, env.n_RB, env.n_Neighbor, feat, env, cfg, seed=1001, device=0, use_graph=True)
agent.train(1, 2)
torch.cuda.synchronize()
t0 = time.perf_counter()
agent.train(2, 20)
torch.cuda.synchronize()
print("ms per train step: %.3f" % (1e3 * (time.perf_counter() - t0) / 40))
pr = cProfile.Profile()
pr.enable()
agent.train(2, 20)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(32)
st.sort_stats('cumulative').print_stats(25)
