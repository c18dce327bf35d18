"""Ragged-batch sweep (variable graph sizes, shared weights) against the float64 oracle -- run on a GPU box."""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import v2xgnn
from v2xgnn import GnnSpec, PackedBatch, GnnEngine
from oracle import compact as oc
from oracle.spec import GnnSpec as OSpec
from util import f32_params, FWD_RTOL, FWD_ATOL, GRAD_RTOL, GRAD_ATOL_REL

rng = np.random.default_rng(11)
bad = 0
for case in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    F = int(rng.choice([16, 32, 64, 128])); L = int(rng.choice([1, 2, 3])); G = int(rng.choice([1, 3, 10, 40]))
    hi = int(rng.choice([2, 9, 40, 128])); dense = bool(rng.integers(0, 2))
    sizes = rng.integers(1, hi + 1, size=G)
    spec = GnnSpec(n_nodes=1, feat_dim=F, n_mp_layers=L, share_weights=True, variable_graphs=True)
    osp = OSpec(n_nodes=1, feat_dim=F, n_mp_layers=L, share_weights=True)
    try:
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
        R = int(offs[-1])
        row_ptr, cols, max_e = [0], [], 0
        for n in sizes:
            if dense:
                adj = ~np.eye(n, dtype=bool)
                for q in range(n):
                    if n > 1:
                        adj[rng.choice([p for p in range(n) if p != q]), q] = False
            else:
                adj = rng.uniform(size=(n, n)) < min(0.5, 6.0 / max(n, 1))
            e_g = 0
            for q in range(n):
                src = np.nonzero(adj[:, q])[0]
                cols.append(src); row_ptr.append(row_ptr[-1] + len(src)); e_g += len(src)
            max_e = max(max_e, e_g)
        col_idx = np.concatenate(cols).astype(np.int32) if cols else np.zeros(0, np.int32)
        x = np.concatenate([rng.normal(0.84, 0.39, size=(R, 4)), rng.normal(0.6, 0.21, size=(R, 4)), np.full((R, 1), 10.0)], 1).astype(np.float32)
        e = rng.normal(0.88, 0.11, size=(R, 4)).astype(np.float32)
        pb = PackedBatch(G, 0, v2xgnn.pack_xe(x, e), np.array(row_ptr, np.int32), col_idx, max_e, graph_off=offs, max_nodes=int(sizes.max()))
        P = f32_params(spec, rng)
        eng = GnnEngine(spec)
        eng.set_weights(oc.params_to_list(P))
        M = oc.csr_to_matrix(offs, pb.row_ptr, pb.col_idx, np.float64)
        q_ref, cache = oc.forward(osp, P, x.astype(np.float64), e.astype(np.float64), M)
        q = eng.forward(pb)
        scale = max(1.0, np.abs(q_ref).max())
        ok_f = np.all(np.abs(q - q_ref) <= FWD_RTOL * np.abs(q_ref) + FWD_ATOL * scale)
        y = (q_ref + rng.normal(0, 1.2, size=q_ref.shape)).astype(np.float32)
        err = q.astype(np.float64) - y
        dq = np.clip(err, -1, 1) / (R * 4)
        g_ref = oc.backward(osp, P, cache, dq)
        eng.forward_backward(pb, y)
        got = v2xgnn.flat_to_keras_list(spec, eng.get_grad_flat())
        ok_g, detail = True, ''
        for ai, (a, b) in enumerate(zip(got, oc.params_to_list(g_ref))):
            sc = float(np.abs(b).max()) or 1.0
            if np.any(np.abs(a - b) > GRAD_RTOL * np.abs(b) + GRAD_ATOL_REL * sc):
                ok_g = False; detail = "arr#%d%s rel %.1e" % (ai, a.shape, float((np.abs(a - b) / sc).max()))
        eng.close()
        status = "ok" if ok_f and ok_g else "MISMATCH fwd=%s grad=%s %s" % (ok_f, ok_g, detail)
    except Exception as exc:                              # noqa
        status = "ERROR %s: %s" % (type(exc).__name__, str(exc)[:140])
    if status != "ok":
        bad += 1
    print("F=%3d L=%d graphs=%2d max=%3d dense=%d rows=%4d  %s" % (F, L, G, hi, dense, int(np.sum(sizes)), status), flush=True)
print("bad cases:", bad)
