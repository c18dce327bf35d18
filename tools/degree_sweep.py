"""Degree sweep of the fused graph-layer kernels: fit steps on 20- and 28-link graphs of in-degree 2, 4, 8 and
N - 2 at batch 4096; per-kernel HIP-event averages.  Sparse lanes walk only their set bits (kernels_fused.hpp, gather_all);
V2X_DEGREE_AWARE is not a switch of the library -- the "before" column comes from the commit before the walk became degree-aware."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ["V2X_FUSED_COMPL"] = "0"
import torch  # noqa: E402
import bench  # noqa: E402
from v2xgnn import GnnSpec, PackedBatch, GnnEngine  # noqa: E402
from util import fixed_indegree_adj  # noqa: E402

B, F = int(os.environ.get("SWEEP_BATCH", "4096")), 64
print("batch %d, feat_dim %d, 2 layers, per-node weights; us per launch (HIP events, eager), ms per step (hipGraph replay)" % (B, F))
for N in (20, 28):
    for deg in (2, 4, 8, N - 2):
        rng = np.random.default_rng(100 * N + deg)
        x, e, _, y = bench.synth_batch(rng, B, N)
        adj = fixed_indegree_adj(rng, 64, N, deg)
        adj = np.tile(adj, (B // 64, 1, 1))
        eng = GnnEngine(GnnSpec(n_nodes=N, feat_dim=F), use_graph=True)
        db = eng.to_device(PackedBatch.from_dense(x, e, adj))
        yd = torch.from_numpy(y).cuda()
        with torch.cuda.stream(torch.cuda.Stream()):
            for _ in range(10):
                eng.train_step(db, yd, want_loss=False)
            torch.cuda.synchronize()
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(200):
                eng.train_step(db, yd, want_loss=False)
            t1.record(); torch.cuda.synchronize()
            ms = t0.elapsed_time(t1) / 200
            eng.profile(True)
            for _ in range(30):
                eng.train_step(db, yd, want_loss=False)
            torch.cuda.synchronize()
            prof = eng.profile_read()
            eng.profile(False)
        k = {n: 1e3 * t / c for n, (c, t) in prof.items()}
        print("N %2d in-degree %2d: %s  step %.4f ms  fwd %.1f  bwd %.1f  (%s)" % (N, deg, eng.path_info(db)["aggregation"], ms,
              k.get("k_gnn_fwd_fused", 0), k.get("k_gnn_bwd_fused", 0), eng.path_info(db)["graph_layers"]))
        eng.close()
