import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from v2xgnn import GnnSpec, PackedBatch, GnnEngine
import v2xgnn
z = np.load(sys.argv[1])
N, F, L, shared, B = int(z['N']), int(z['F']), int(z['L']), bool(z['shared']), int(z['B'])
spec = GnnSpec(n_nodes=N, feat_dim=F, n_mp_layers=L, share_weights=shared)
shapes = v2xgnn.keras_list_shapes(spec)
ws, pos = [], 0
for sh in shapes:
    n = int(np.prod(sh)); ws.append(z['w'][pos:pos + n].reshape(sh).astype(np.float32)); pos += n
for rep in range(3):
    eng = GnnEngine(spec)
    eng.set_weights(ws)
    pb = PackedBatch.from_dense(z['x'], z['e'], z['adj'])
    q = eng.forward(pb)
    eng.forward_backward(pb, z['y'])
    g = eng.get_grad_flat()
    print("rep", rep, "q equal:", np.array_equal(q, z['q']), "grad equal to the sweep's:", np.array_equal(g, z['g']),
          "max|dg|", np.abs(g - z['g']).max(), "argmax", int(np.argmax(np.abs(g - z['g']))))
    eng.close()
