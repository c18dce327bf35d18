"""Randomised shape sweep of the whole model (forward, per-output loss, every gradient array) against the float64 oracle --
run on a GPU box:
    python tests/sweep_shapes.py [n_cases]
Every case is one seeded draw, judged exactly as tests/test_gpu_shapes.py judges its permanent subset: plain rounding
tolerances, ReLU gates at fp32 rounding distance of 0 identified explicitly and taken the kernels' way (tests/util.py),
nothing redrawn.  The grid: 7 link counts x 4 feature widths x 3 depths x shared / per-node weights x 3 batch sizes = 504
shapes, reference topology or a random adjacency."""
import itertools, os, sys, traceback
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from v2xgnn import GnnSpec, PackedBatch, GnnEngine
import v2xgnn
from oracle import compact as oc
from util import f32_params, random_inputs, oracle_step, assert_fwd_close, assert_close, assert_grads_match_oracle

rng = np.random.default_rng(7)
cases = list(itertools.product([1, 2, 3, 7, 20, 33, 40], [16, 32, 64, 128], [1, 2, 4], [False, True], [1, 17, 130]))
rng.shuffle(cases)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = flips = 0
os.environ["V2X_SMALL_PREDICT"] = "0"          # forward() on the training path's kernels (the loss is differentiated at its q)
for (N, F, L, shared, B) in cases[:n]:
    spec = GnnSpec(n_nodes=N, feat_dim=F, n_mp_layers=L, share_weights=shared)
    status = "ok"
    try:
        P = f32_params(spec, rng)
        topo = bool(rng.integers(0, 2)) and N > 2
        x, e, adj = random_inputs(rng, B, N, ref_topology=topo)
        pb = PackedBatch.from_dense(x, e, adj)
        eng = GnnEngine(spec)
        eng.set_weights(oc.params_to_list(P))
        graph = ((np.arange(B + 1) * N).astype(np.int32), pb.row_ptr, pb.col_idx)
        q = eng.forward(pb)
        y = (q + rng.normal(0, 1.2, size=q.shape)).astype(np.float32)
        step = oracle_step(spec, P, x.reshape(B * N, -1), e.reshape(B * N, -1), graph, y, q_at=q)
        assert_fwd_close(q, step['q'], "forward")
        loss = eng.forward_backward(pb, y)
        assert_close(loss, step['loss'], 2e-4, 1e-6, "per-output Huber loss")
        _, n_cand, n_flip = assert_grads_match_oracle(v2xgnn.flat_to_keras_list(spec, eng.get_grad_flat()), P, step, "gradients")
        flips += n_flip
        if n_flip:
            status = "ok (%d of %d candidate ReLU gates taken the kernels' way)" % (n_flip, n_cand)
        eng.close()
    except AssertionError as exc:
        status = "MISMATCH ref_topo=%s: %s" % (topo, str(exc).splitlines()[0][:200])
    except Exception as exc:                      # noqa
        status = "ERROR %s: %s" % (type(exc).__name__, str(exc)[:160])
        traceback.print_exc()
    if not status.startswith("ok"):
        bad += 1
    print("N=%2d F=%3d L=%d shared=%d B=%3d  %s" % (N, F, L, shared, B, status), flush=True)
print("cases: %d, bad: %d, ReLU gates resolved: %d" % (min(n, len(cases)), bad, flips))
