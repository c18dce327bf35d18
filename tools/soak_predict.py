"""Soak of the one-launch predict (grid barriers between the layers): 30,000 forwards of 1..12 graphs interleaved with fit
steps and weight copies on two models sharing the GPU; the barrier counters must come back to zero every time."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from v2xgnn import GnnSpec, PackedBatch, GnnEngine
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import random_inputs
rng = np.random.default_rng(0)
spec = GnnSpec(n_nodes=20, feat_dim=64)
a, b = GnnEngine(spec, use_graph=True), GnnEngine(spec)
batches = {}
for B in range(1, 13):
    x, e, adj = random_inputs(rng, B, 20, ref_topology=True)
    batches[B] = PackedBatch.from_dense(x, e, adj)
xt, et, at = random_inputs(rng, 256, 20, ref_topology=True)
train = a.to_device(PackedBatch.from_dense(xt, et, at))
yt = torch.from_numpy(rng.normal(2.5, 1.0, size=(256 * 20, 4)).astype(np.float32)).cuda()
t0 = time.perf_counter()
ref = {B: None for B in batches}
for it in range(30000):
    B = 1 + it % 12
    eng = a if it % 3 else b
    q = eng.forward(batches[B])
    assert np.all(np.isfinite(q))
    if it % 50 == 0:
        a.train_step(train, yt, want_loss=False)
    if it % 500 == 0:
        b.copy_weights_from(a) if hasattr(b, "copy_weights_from") else b.set_weights(a.get_weights())
print("30000 predicts + 600 fit steps in %.1f s: %.1f us per predict on average" % (time.perf_counter() - t0, (time.perf_counter() - t0) / 30000 * 1e6))
