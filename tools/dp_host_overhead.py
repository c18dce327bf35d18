"""Host time and wall time of one data-parallel step in its three forms (single rank, RCCL path forced): one all-reduce,
per-bucket all-reduces overlapped with the backward phases, sharded optimizer step.  `--wide`: configs[3]'s per-GPU share
(100 links x 256 features x 3 layers, 1024 graphs, 207.6 MB of gradients in L + 2 = 5 buckets) instead of the headline model."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29611")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import bench  # noqa: E402
from v2xgnn import GnnSpec, PackedBatch, GnnEngine  # noqa: E402
from v2xgnn.dp import DataParallelTrainer  # noqa: E402

wide = "--wide" in sys.argv
dist.init_process_group(os.environ.get("V2X_BENCH_BACKEND", "nccl"), rank=0, world_size=1)
N, F, L, B = (100, 256, 3, 1024) if wide else (20, 64, 2, int(os.environ.get("DP_BATCH", "4096")))     # DP_BATCH=512: the 8-GPU share
rng = np.random.default_rng(1001)
x, e, adj, y = bench.synth_batch(rng, B, N)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    eng = GnnEngine(GnnSpec(n_nodes=N, feat_dim=F, n_mp_layers=L), use_graph=os.environ.get("DP_GRAPH", "0") == "1")       # DP_GRAPH=1: the phases as hipGraph replays (round 5's form)
    db = eng.to_device(PackedBatch.from_dense(x, e, adj))
    yd = torch.from_numpy(y).cuda()
    n = 30 if wide else 300
    print("model: %d links x %d features x %d layers, %d graphs, %.1f MB of gradients, buckets (MB): %s"
          % (N, F, L, B, 4e-6 * eng.n_params, [round(4e-6 * c, 2) for _, c in eng.grad_buckets()]))
    for name, kw in (("one all-reduce", {}), ("per-bucket all-reduce, overlapped", dict(overlap=True)),
                     ("sharded optimizer (reduce-scatter + Adam on the slice + all-gather)", dict(shard_optimizer=True))):
        tr = DataParallelTrainer(eng, force=True, **kw)
        for _ in range(5 if wide else 30):
            tr.train_step(db, yd, B, want_loss=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            tr.train_step(db, yd, B, want_loss=False)
        host = (time.perf_counter() - t0) / n
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / n
        print("%-75s host issue %.1f us/step, wall %.1f us/step" % (name + ":", host * 1e6, wall * 1e6))

    def t(fn, n=n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        d = (time.perf_counter() - t0) / n
        torch.cuda.synchronize()
        w = (time.perf_counter() - t0) / n
        return d * 1e6, w * 1e6
    nb = len(eng.grad_buckets())
    g = eng.grad_tensor()
    for k in range(nb):
        print("phase %d call                 host %.1f us, wall %.1f us" % ((k,) + t(lambda: [eng.forward_backward_phase(db, yd, p, n_global=B, want_loss=False) for p in range(k + 1)][-1])))
    for k, (o, c) in enumerate(eng.grad_buckets()):
        b = g[o:o + c]
        print("all_reduce bucket %d (%.1f MB) host %.1f us, wall %.1f us" % ((k, 4e-6 * c) + t(lambda: dist.all_reduce(b, async_op=True).wait())))
    print("all_reduce whole gradient    host %.1f us, wall %.1f us" % t(lambda: dist.all_reduce(g, async_op=True).wait()))
    print("apply_gradients              host %.1f us, wall %.1f us" % t(lambda: eng.apply_gradients()))
    print("train_step (1-GPU path)      host %.1f us, wall %.1f us" % t(lambda: eng.train_step(db, yd, want_loss=False)))
dist.destroy_process_group()
