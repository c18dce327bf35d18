"""Host time of every call of one data-parallel step (single rank, RCCL path forced): is the step GPU- or host-bound?"""
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

dist.init_process_group(os.environ.get("V2X_BENCH_BACKEND", "nccl"), rank=0, world_size=1)
N, F, B = 20, 64, 4096
rng = np.random.default_rng(1001)
x, e, adj, y = bench.synth_batch(rng, B, N)
eng = GnnEngine(GnnSpec(n_nodes=N, feat_dim=F), use_graph=True)
db = eng.to_device(PackedBatch.from_dense(x, e, adj))
yd = torch.from_numpy(y).cuda()
for overlap in (True, False):
    tr = DataParallelTrainer(eng, force=True, overlap=overlap)
    for _ in range(30):
        tr.train_step(db, yd, B, want_loss=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 300
    for _ in range(n):
        tr.train_step(db, yd, B, want_loss=False)
    host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n
    print("overlap=%s: host issue time %.1f us/step, wall %.1f us/step" % (overlap, host * 1e6, wall * 1e6))
# the pieces
tr = DataParallelTrainer(eng, force=True, overlap=True)
tr.train_step(db, yd, B, want_loss=False)
g = tr._grad
b0, b1 = tr._buckets
def t(fn, n=300):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    d = (time.perf_counter() - t0) / n; torch.cuda.synchronize(); return d * 1e6
print("phase 0 call      %.1f us" % t(lambda: eng.forward_backward_phase(db, yd, 0, n_global=B)))
print("phase 1 call      %.1f us" % t(lambda: eng.forward_backward_phase(db, yd, 1, n_global=B, want_loss=False)))
print("all_reduce async  %.1f us" % t(lambda: dist.all_reduce(b0, async_op=True).wait()))
print("apply_gradients   %.1f us" % t(lambda: eng.apply_gradients()))
print("train_step (1 GPU path) %.1f us" % t(lambda: eng.train_step(db, yd, want_loss=False)))
dist.destroy_process_group()
