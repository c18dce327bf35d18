"""Latency of the rollout-path predict (B = 1) through BS / GnnQModel: dict payload and compact arrays."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from v2xgnn import BS
from v2xgnn.packing import PackedBatch
for (N, F) in ((4, 16), (20, 64)):
    for graph in (False, True):
        brain = BS(N, 3, 1, F, 1, 4, seed=1, use_graph=graph)
        rng = np.random.default_rng(0)
        x, e = rng.normal(size=(1, N, 9)), rng.normal(size=(1, N, 4))
        adj = np.ones((1, N, N)) - np.eye(N)[None]
        for q in range(N):
            adj[0, (q + 1) % N, q] = 0
        m = brain.model
        for _ in range(20):
            m.predict_arrays(x, e, adj)
        t = time.perf_counter()
        for _ in range(300):
            m.predict_arrays(x, e, adj)
        dt = (time.perf_counter() - t) / 300
        pb = PackedBatch.from_dense(x, e, adj)
        t = time.perf_counter()
        for _ in range(300):
            m.engine.forward(pb)
        de = (time.perf_counter() - t) / 300
        t = time.perf_counter()
        for _ in range(300):
            PackedBatch.from_dense(x, e, adj)
        dp = (time.perf_counter() - t) / 300
        print("N=%d F=%d graph=%s: predict_arrays %.0f us | engine.forward(packed) %.0f us | packing %.0f us" % (N, F, graph, dt * 1e6, de * 1e6, dp * 1e6))
