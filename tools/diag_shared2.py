import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import v2xgnn
from v2xgnn import GnnSpec, PackedBatch, GnnEngine
from oracle import compact as oc
from util import f32_params, oracle_step, ospec
import bench
N, F, B = 20, 64, 4096
os.environ["V2X_SMALL_PREDICT"] = "0"
rng = np.random.default_rng(2026)
x, e, adj, _ = bench.synth_batch(rng, B, N)
spec = GnnSpec(n_nodes=N, feat_dim=F, share_weights=True)
P = f32_params(spec, rng)
pb = PackedBatch.from_dense(x, e, adj)
graph = ((np.arange(B + 1) * N).astype(np.int32), pb.row_ptr, pb.col_idx)
eng = GnnEngine(spec)
eng.set_weights(oc.params_to_list(P))
q = eng.forward(pb)
y = (q + np.random.default_rng(99).normal(0, 1.2, size=q.shape)).astype(np.float32)
ref = oracle_step(spec, P, x.reshape(B * N, -1), e.reshape(B * N, -1), graph, y, q_at=q)
eng.forward_backward(pb, y)
g_full = eng.get_grad_flat().astype(np.float64)
got = v2xgnn.flat_to_keras_list(spec, g_full.astype(np.float32))
refl = oc.params_to_list(ref['grads'])
res0 = got[0] - refl[0]
print("array0 residual: per-column max", np.round(np.abs(res0).max(axis=0), 3))
print("array0 residual rows (k) for worst col", np.round(res0[:, np.abs(res0).max(axis=0).argmax()], 4))
b_emb = got[3] - refl[3]
print("embed bias residual top:", np.argsort(-np.abs(b_emb.ravel()))[:8], np.round(np.sort(-np.abs(b_emb.ravel()))[:8], 5))
# candidates and their pre-gate magnitudes in stage 0
pre = ref['cache']['relu_pre'][0]
rr, ff = np.nonzero(np.abs(pre) <= 2e-5 * np.abs(pre).max())
print("stage0 candidates (2e-5):", rr.size, " |pre|/max sorted:", np.round(np.sort(np.abs(pre[rr, ff]) / np.abs(pre).max())[:10] * 1e6, 2), "e-6")
# chunks
yb = y.reshape(B, N, 4)
acc = np.zeros_like(g_full)
os_ = ospec(spec)
M = ref['cache']['M']
worst = []
for c in range(16):
    g0, g1 = c * 256, (c + 1) * 256
    sh, _ = pb.slice_graphs(g0, g1)
    eng.forward_backward(sh, yb[g0:g1].reshape(-1, 4), n_global=B)
    gc = eng.get_grad_flat().astype(np.float64)
    acc += gc
    # oracle gradient of the chunk: dq zero outside
    dq = np.zeros_like(ref['dq']); dq[g0 * N:g1 * N] = ref['dq'][g0 * N:g1 * N]
    gr = oc.params_to_list(oc.backward(os_, P, ref['cache'], dq))
    gl = v2xgnn.flat_to_keras_list(spec, gc.astype(np.float32))
    worst.append(max(float(np.abs(a - b).max() / (np.abs(rb).max() or 1)) for a, b, rb in zip(gl, gr, refl)))
print("per-chunk max|err|/max|ref_full| :", " ".join("%.0e" % w for w in worst))
print("full vs sum of chunks (GPU): max rel", np.abs(acc - g_full).max() / np.abs(g_full).max())
accl = v2xgnn.flat_to_keras_list(spec, acc.astype(np.float32))
print("sum of GPU chunks vs oracle: ", " ".join("%.0e" % (np.abs(a - b).max() / (np.abs(b).max() or 1)) for a, b in zip(accl, refl)))


# find the flipped units: per-graph gradients of the chunks that disagree
for c in [i for i, w in enumerate(worst) if w > 1e-5]:
    for g in range(c * 256, (c + 1) * 256):
        sh, _ = pb.slice_graphs(g, g + 1)
        eng.forward_backward(sh, yb[g:g + 1].reshape(-1, 4), n_global=B)
        gl = v2xgnn.flat_to_keras_list(spec, eng.get_grad_flat())
        dq = np.zeros_like(ref['dq']); dq[g * N:(g + 1) * N] = ref['dq'][g * N:(g + 1) * N]
        gr = oc.params_to_list(oc.backward(os_, P, ref['cache'], dq))
        errs = [float(np.abs(a - b).max() / (np.abs(rb).max() or 1)) for a, b, rb in zip(gl, gr, refl)]
        if max(errs) > 1e-5:
            print("graph", g, "per-array err", " ".join("%.0e" % v for v in errs))
            for t, bi in ((4, 19), (3, 17), (2, 15) , (1, 7), (0, 3)):     # bias arrays of dense2, dense1, dense0, stage1, stage0 in the Keras list (shared: 4 per stage, 2 per dense)
                pass
            names = {19: 'dense3.b', 17: 'dense2.b?', }
            for bi in range(len(gl)):
                if gl[bi].ndim == 1:
                    d = gl[bi] - gr[bi]
                    f = int(np.abs(d).argmax())
                    if np.abs(d[f]) > 1e-5 * (np.abs(refl[bi]).max() or 1):
                        print("   bias array", bi, "feature", f, "residual %.4e" % d[f])
            # candidate pre-activations of this graph, every ReLU tensor: smallest |pre|/max
            for t, pre in enumerate(ref['cache']['relu_pre']):
                blk = np.abs(pre[g * N:(g + 1) * N]) / np.abs(pre).max()
                r, f = np.unravel_index(blk.argmin(), blk.shape)
                print("   relu tensor", t, "min |pre|/max %.2e at node %d feature %d (pre %.3e, max %.3e)" % (blk[r, f], r, f, pre[g * N + r, f], np.abs(pre).max()))
# resolution trace
import util
got_s = oc.params_from_list(os_, got, np.float64)
for rtol in (4e-6, 2e-5):
    r2, nc, nf = util.resolve_relu_gates(os_, P, ref['cache'], ref['dq'], got_s, ref['grads'], ref['pre_gate'], rtol)
    print("rtol", rtol, "cand", nc, "flips", nf, "after:", " ".join("%.0e" % (np.abs(a - b).max() / (np.abs(b).max() or 1)) for a, b in zip(got, oc.params_to_list(r2))))
