"""Host cost of the native simulator's per-step calls against the OpenMP thread count (the calls are ~4 ms apart in the
DQN loop: the threads are asleep when the next one arrives)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import v2xgnn  # noqa: E402,F401
from v2xgnn.rl import native_sim  # noqa: E402

E, n, rb = 50, 20, 4
n_draws = n + n * n + 2 * n * rb + 2 * n * n * rb
keys = np.random.default_rng(1).integers(0, 2**32, size=(E, 624), dtype=np.uint32)
pos = np.full(E, 624, np.int32)
print("cores:", len(os.sched_getaffinity(0)))
for thr in (1, 2, 4, 8, 16, 32):
    native_sim.set_threads(thr)
    for gap in (0.0, 0.004):
        ts = []
        for _ in range(30):
            time.sleep(gap)
            t0 = time.perf_counter()
            native_sim.mt_uniforms(keys, pos, n_draws)
            ts.append(time.perf_counter() - t0)
        print("threads %2d gap %.0f ms: mt_uniforms median %.0f us  min %.0f us" % (thr, gap * 1e3, 1e6 * np.median(ts), 1e6 * min(ts)))

# the same call inside a process that has the GPU runtime up (the DQN loop's situation)
import torch  # noqa: E402
from v2xgnn import GnnSpec, GnnEngine  # noqa: E402
torch.zeros(1, device="cuda")
eng = GnnEngine(GnnSpec(n_nodes=20, feat_dim=64))
lib = native_sim._load()
import ctypes as C  # noqa: E402
out = np.empty((E, n_draws))
for thr in (1, 8, 32):
    native_sim.set_threads(thr)
    for reuse in (False, True):
        ts = []
        for _ in range(30):
            time.sleep(0.004)
            t0 = time.perf_counter()
            if reuse:
                lib.v2xsim_mt_uniforms(E, keys.ctypes.data_as(C.POINTER(C.c_uint32)), pos.ctypes.data_as(C.POINTER(C.c_int32)),
                                       out.ctypes.data_as(C.POINTER(C.c_double)), n_draws)
            else:
                native_sim.mt_uniforms(keys, pos, n_draws)
            ts.append(time.perf_counter() - t0)
        print("GPU runtime up, threads %2d, %s: median %.0f us  min %.0f us" % (thr, "reused buffer" if reuse else "fresh np.empty", 1e6 * np.median(ts), 1e6 * min(ts)))
