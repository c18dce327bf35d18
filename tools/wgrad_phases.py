"""Where does the graph layers' weight-gradient launch spend its time?  Shader-clock stamps of one workgroup of the
heaviest role (V2X_FUSED_TS=1): start | per 16-row block | loop end | slab written."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["V2X_FUSED_TS"] = "1"
import bench  # noqa: E402
from v2xgnn import GnnSpec, PackedBatch, GnnEngine  # noqa: E402

N, F, B = 20, 64, int(os.environ.get("PHASES_BATCH", "4096"))
rng = np.random.default_rng(1001)
x, e, adj, y = bench.synth_batch(rng, B, N)
eng = GnnEngine(GnnSpec(n_nodes=N, feat_dim=F))
db = eng.to_device(PackedBatch.from_dense(x, e, adj))
import torch  # noqa: E402
yd = torch.from_numpy(y).cuda()
for _ in range(5):
    eng.train_step(db, yd)
torch.cuda.synchronize()
buf = (C.c_int64 * (4 * 512))()
assert eng._lib.v2x_debug_phase_stamps(eng._h, buf, 4 * 512) == 0
t = np.array(buf[:], np.int64).reshape(4, 8, 64)[3][:4]
for w in range(4):
    row = t[w]
    n = int((row != 0).sum())
    clk = row[1:n - 1]
    wall = (row[n - 1] - row[0]) / 100.0
    d = np.diff(clk)
    print("wave %d: wall %.2f us, %d clk => %.2f GHz" % (w, wall, clk[-1] - clk[0], (clk[-1] - clk[0]) / wall / 1e3))
    print("   per block: " + " ".join("%d" % v for v in d[:-2]))
    print("   tail blocks + to exchange: %d | exchange + slab: %d" % (d[-2], d[-1]))
