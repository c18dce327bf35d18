import cProfile, pstats, random, sys
import numpy as np
sys.path.insert(0, '.')
from v2xgnn.rl import RL_Config
from v2xgnn.rl.train import start_env, run_train
random.seed(1001); np.random.seed(1001)
cfg = RL_Config(); cfg.set_train_value(64, 0.5, 4096, 1, 0.1); cfg.Num_Episodes, cfg.Num_Train_Steps = 1, 10
env = start_env(20)
pr = cProfile.Profile(); pr.enable()
run_train(env, cfg, seed=1, use_graph=True)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
