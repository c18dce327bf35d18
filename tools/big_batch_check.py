import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench, v2xgnn, torch
from v2xgnn import GnnSpec, PackedBatch, GnnEngine
N, F, B = 20, 64, 65536
rng = np.random.default_rng(1)
x, e, adj, y = bench.synth_batch(rng, B, N)
spec = GnnSpec(n_nodes=N, feat_dim=F)
shapes = v2xgnn.keras_list_shapes(spec)
w = [np.zeros(s, np.float32) if len(s) == 1 else rng.uniform(-np.sqrt(6.0 / sum(s)), np.sqrt(6.0 / sum(s)), size=s).astype(np.float32) for s in shapes]
eng = GnnEngine(spec); eng.set_weights(w)
pb = PackedBatch.from_dense(x, e, adj)
db = eng.to_device(pb); yd = torch.from_numpy(y).cuda(); torch.cuda.synchronize()
q = eng.forward(db).cpu().numpy().astype(np.float64)
ab = np.abs(q - y); quad = np.minimum(ab, 1.0)
ref = (0.5 * quad * quad + (ab - quad)).reshape(B, N, 4).mean(axis=(0, 2))
loss = eng.forward_backward(db, yd).cpu().numpy()
print("loss rel err", np.abs(loss - ref).max() / ref.max(), "finite grads", np.isfinite(eng.get_grad_flat()).all())
# additivity: two halves
g = eng.get_grad_flat().astype(np.float64)
acc = np.zeros_like(g)
for r in range(2):
    sh = pb.shard(r, 2)
    eng.forward_backward(sh, y.reshape(B, N, 4)[r * B // 2:(r + 1) * B // 2].reshape(-1, 4), n_global=B)
    acc += eng.get_grad_flat()
print("halves vs full max rel", np.abs(acc - g).max() / np.abs(g).max())
t = time.perf_counter()
for _ in range(10): eng.train_step(db, yd, want_loss=False)
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t) * 100, "graphs/s", B * 10 / (time.perf_counter() - t))
