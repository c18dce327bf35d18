"""Where does the fused forward kernel spend its time?  Phase time stamps of workgroup 7 (V2X_FUSED_TS=1 build)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["V2X_FUSED_TS"] = "1"
import bench  # noqa: E402
import v2xgnn  # noqa: E402
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
buf = (C.c_int64 * 1024)()
rc = eng._lib.v2x_debug_phase_stamps(eng._h, buf, 1024)
assert rc == 0
for name, t in zip(("forward", "backward"), np.array(buf[:], np.int64).reshape(2, 8, 64)):
    t0 = t[:, 0].min()
    print(name)
    for w in range(8):
        row = t[w][t[w] > 0]
        print("  wave %d:" % w, " ".join("%6.2f" % ((v - t0) / 100.0) for v in row))
print(eng.path_info(db))
print("split-tile marks (workgroup 8), forward: 0 start | 1 CSR + sets | 2 embed published | 3 partners' h_0 in | 4 barrier | per stage: gathers, updates published, barrier, partners' rows in, barrier | end;  backward: 0 start | 1 tile + masks | per stage: gathers, dgrad published, barrier, rows in, barrier | end")
print("backward marks: 0 start | 1 tile + masks ready | per stage: gathers done, slot done x n, barrier, tile replaced | end")
print("forward marks: 0 start | 1 row_ptr ends known | 2 CSR slice in LDS | 3 masks built (complement form) | 4 embed done | 5 barrier | per stage: gathers done, slot done x n, (stores drained,) barrier, tile replaced | end   (us)")
