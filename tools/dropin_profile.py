#!/usr/bin/env python3
"""Where the time of the dict boundary goes (bench.py leg `dropin_ref_config`): stage by stage, N=4 F=16 B=512."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
from v2xgnn import BS  # noqa: E402
from v2xgnn.packing import feed_to_packed, AdjacencyCache  # noqa: E402


def med(fn, n=200, warm=5):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return 1e6 * float(np.median(ts))


def main():
    N, F, B, C = 4, 16, 512, 4
    rng = np.random.default_rng(1001)
    brain = BS(N, 3, 1, F, 1, C, seed=7)
    x, e, adj, _ = bench.synth_batch(rng, B, N)
    d = {}
    for k in range(N):
        d['D%d_Node_Input' % (k + 1)] = np.ascontiguousarray(x[:, k, :], np.float64)
        d['D%d_Edge_Input' % (k + 1)] = np.ascontiguousarray(e[:, k, :], np.float64)
        d['D%d_Neighbor_Input' % (k + 1)] = np.zeros((B, F))
    d['Adjacency_Matrix'] = np.kron(adj.astype(np.float64), np.eye(F))
    spec = brain.model.spec
    cache = AdjacencyCache()
    feed_to_packed(spec, d, True, cache)
    print("pack, full scan        %8.1f us" % med(lambda: feed_to_packed(spec, d, True, None)))
    print("pack, cache hit        %8.1f us" % med(lambda: feed_to_packed(spec, d, True, cache)))
    print("pack, no validation    %8.1f us" % med(lambda: feed_to_packed(spec, d, False, None)))
    pb = feed_to_packed(spec, d, True, cache)
    eng = brain.model.engine
    print("engine.forward(host)   %8.1f us" % med(lambda: eng.forward(pb)))
    y = np.random.default_rng(1).normal(size=(B * N, C)).astype(np.float32)
    print("engine.train_step(host)%8.1f us" % med(lambda: eng.train_step(pb, y)))
    db = eng.to_device(pb)
    yd = torch.from_numpy(y).cuda()

    def dev_fwd():
        eng.forward(db)
        torch.cuda.synchronize()

    def dev_train():
        eng.train_step(db, yd, want_loss=False)
        torch.cuda.synchronize()
    print("engine.forward(dev)+sync %6.1f us" % med(dev_fwd))
    print("engine.train_step(dev)+sync %3.1f us" % med(dev_train))
    print("BS.predict             %8.1f us" % med(lambda: brain.predict(d)))
    p = brain.predict(d)
    yt = {'D%d_Decide_Output' % (k + 1): p[k] + 0.1 for k in range(N)}
    print("BS.train_dnn           %8.1f us" % med(lambda: brain.train_dnn(d, yt, B)))
    one = {k: v[:1].copy() for k, v in d.items()}
    print("BS.predict_one_step    %8.1f us" % med(lambda: brain.predict_one_step(one), 500))
    pb1 = feed_to_packed(spec, one, True, None)
    print("  pack (B=1)           %8.1f us" % med(lambda: feed_to_packed(spec, one, True, None), 500))
    print("  engine.forward (B=1) %8.1f us" % med(lambda: eng.forward(pb1), 500))


if __name__ == "__main__":
    main()
