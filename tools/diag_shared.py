"""diagnostic: per-array gradient error of the engine vs the float64 oracle, shared weights, over batch sizes / switches"""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import v2xgnn
from v2xgnn import GnnSpec, PackedBatch, GnnEngine
from oracle import compact as oc
from util import f32_params, oracle_step
import bench

N, F = 20, 64
def run(B, shared, env):
    for k, v in env.items(): os.environ[k] = v
    os.environ["V2X_SMALL_PREDICT"] = "0"
    try:
        rng = np.random.default_rng(2025 + shared)
        x, e, adj, _ = bench.synth_batch(rng, B, N)
        spec = GnnSpec(n_nodes=N, feat_dim=F, share_weights=shared)
        P = f32_params(spec, rng)
        pb = PackedBatch.from_dense(x, e, adj)
        graph = ((np.arange(B + 1) * N).astype(np.int32), pb.row_ptr, pb.col_idx)
        eng = GnnEngine(spec)
    finally:
        for k in list(env) + ["V2X_SMALL_PREDICT"]: del os.environ[k]
    eng.set_weights(oc.params_to_list(P))
    q = eng.forward(pb)
    y = (q + np.random.default_rng(99).normal(0, 1.2, size=q.shape)).astype(np.float32)
    ref = oracle_step(spec, P, x.reshape(B * N, -1), e.reshape(B * N, -1), graph, y, q_at=q)
    eng.forward_backward(pb, y)
    got = v2xgnn.flat_to_keras_list(spec, eng.get_grad_flat())
    worst = []
    for i, (a, b) in enumerate(zip(got, oc.params_to_list(ref['grads']))):
        sc = np.abs(b).max() or 1.0
        worst.append(float(np.abs(a - b).max() / sc))
    print("B=%5d shared=%d env=%s  fwd max rel %.1e | per-array max|err|/max|ref|: %s" % (
        B, shared, env, np.abs(q - ref['q']).max() / np.abs(ref['q']).max(), " ".join("%.0e" % w for w in worst)), flush=True)
    eng.close()

for B in (256, 1024, 4096):
    run(B, True, {})
run(4096, True, {"V2X_FUSED": "0"})
run(4096, True, {"V2X_MLP_WG": "0"})
run(4096, True, {"V2X_WG_EMBED_MERGE": "0"})
run(4096, True, {"V2X_FUSED": "0", "V2X_MLP_WG": "0"})
run(4096, False, {})
