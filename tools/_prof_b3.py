import sys, time, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from v2xgnn.rl.batched_env import BatchedEnviron
up = [3.5 / 2, 3.5 / 2 + 3.5, 250 + 3.5 / 2, 250 + 3.5 + 3.5 / 2, 500 + 3.5 / 2, 500 + 3.5 + 3.5 / 2]
dn = [250 - 3.5 - 3.5 / 2, 250 - 3.5 / 2, 500 - 3.5 - 3.5 / 2, 500 - 3.5 / 2, 750 - 3.5 - 3.5 / 2, 750 - 3.5 / 2]
le = [3.5 / 2, 3.5 / 2 + 3.5, 433 + 3.5 / 2, 433 + 3.5 + 3.5 / 2, 866 + 3.5 / 2, 866 + 3.5 + 3.5 / 2]
ri = [433 - 3.5 - 3.5 / 2, 433 - 3.5 / 2, 866 - 3.5 - 3.5 / 2, 866 - 3.5 / 2, 1299 - 3.5 - 3.5 / 2, 1299 - 3.5 / 2]
def mk(E, native):
    env = BatchedEnviron(dn, up, le, ri, 750, 1299, n_envs=E, seeds=[7 + 104729 * e for e in range(E)], native=native)
    env.new_random_game(20); return env
for E in (10, 50, 100):
    for native in (False, True):
        env = mk(E, native); rng = np.random.default_rng(0)
        t = time.perf_counter(); n_it = 2000 // E
        for it in range(n_it):
            s, adj = env.observe(4); env.act(rng.integers(0, 4, size=(E, 20, 1)))
        print("E=%d native=%s: %.1f us per env-step" % (E, native, (time.perf_counter() - t) / (n_it * E) * 1e6))
env = mk(50, True); rng = np.random.default_rng(0)
pr = cProfile.Profile(); pr.enable()
for it in range(40):
    s, adj = env.observe(4); env.act(rng.integers(0, 4, size=(50, 20, 1)))
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(10)
