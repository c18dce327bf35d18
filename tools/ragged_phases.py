"""Phase time stamps of one workgroup of the ragged fused forward (V2X_FUSED_TS=1 build of k_gnn_fwd_ragged)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["V2X_FUSED_TS"] = "1"
import bench  # noqa: E402
import v2xgnn  # noqa: E402
from v2xgnn import GnnSpec, PackedBatch, GnnEngine  # noqa: E402
import torch  # noqa: E402

sizes, offs, row_ptr, col_idx, x, e, y = bench.synth_ragged(np.random.default_rng(1001), 2048, 8, 128)
eng = GnnEngine(GnnSpec(n_nodes=1, feat_dim=64, share_weights=True, variable_graphs=True))
db = eng.to_device(PackedBatch(2048, 0, v2xgnn.pack_xe(x, e), row_ptr, col_idx, graph_off=offs))
yd = torch.from_numpy(y).cuda()
for _ in range(5):
    eng.train_step(db, yd)
torch.cuda.synchronize()
buf = (C.c_int64 * 1024)()
assert eng._lib.v2x_debug_phase_stamps(eng._h, buf, 1024) == 0
t = np.array(buf[:512], np.int64).reshape(8, 64)
t0 = t[:, 0].min()
for w in range(8):
    row = t[w][t[w] > 0]
    print("wave %d:" % w, " ".join("%6.2f" % ((v - t0) / 100.0) for v in row))
print("marks (us): 0 start | 1 prologue loads issued | 2 barrier | 3 records + embed weights, barrier | per stage: MFMAs + h stores issued, "
      "barrier A, tile + next weights written, barrier B, sums, barrier C, aggregation + a stores issued")
