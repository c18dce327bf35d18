"""Soak of the split-tile fused graph layers (tagged-word hand-overs between the workgroups of a tile, kernels_fused_split.hpp)
under UNEVEN load: fit steps of a split-tile engine (512- and 1024-graph shares, hipGraph replay) on one stream while a second
engine keeps the chip busy with whole-tile fit steps of a 4096-graph batch on another stream (so that the members of a tile
start at different times, queue behind foreign workgroups and poll while their CU's memory path is loaded); every few hundred
steps the split engine's parameters must be BIT-identical to those of a whole-tile engine that made the same steps alone.
A stale or torn row handed over between two members would change the bits."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from v2xgnn import GnnSpec, PackedBatch, GnnEngine  # noqa: E402

N, F = 20, 64
steps = int(os.environ.get("SOAK_STEPS", "6000"))
rng = np.random.default_rng(11)
spec = GnnSpec(n_nodes=N, feat_dim=F)


def engine(split):
    if split is not None:
        os.environ["V2X_FUSED_SPLIT"] = str(split)
    os.environ["V2X_FUSED_COMPL"] = "0"
    try:
        return GnnEngine(spec, use_graph=True)
    finally:
        os.environ.pop("V2X_FUSED_SPLIT", None)
        os.environ.pop("V2X_FUSED_COMPL", None)


sp, whole, noise = engine(None), engine(0), engine(0)
whole.set_weights(sp.get_weights())
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
data = {}
for B in (512, 1024, 4096):
    x, e, adj, y = bench.synth_batch(rng, B, N)
    pb = PackedBatch.from_dense(x, e, adj)
    with torch.cuda.stream(sa):
        data[B] = (sp.to_device(pb), torch.from_numpy(y).cuda())
assert "split" in sp.path_info(data[512][0])["graph_layers"] and "split" in sp.path_info(data[1024][0])["graph_layers"]
assert whole.path_info(data[512][0])["graph_layers"] == "fused"
torch.cuda.synchronize()
t0 = time.perf_counter()
checks = 0
for it in range(steps):
    B = 512 if (it // 7) % 2 == 0 else 1024
    db, yd = data[B]
    with torch.cuda.stream(sb):                      # the noise: bursts of whole-tile steps of the full batch
        if it % 3 != 2:
            noise.train_step(data[4096][0], data[4096][1], want_loss=False)
    with torch.cuda.stream(sa):
        sp.train_step(db, yd, want_loss=False)
    if it % 300 == 299:
        torch.cuda.synchronize()
        with torch.cuda.stream(sa):                  # the reference makes the same 300 steps alone
            for k in range(it - 299, it + 1):
                Bk = 512 if (k // 7) % 2 == 0 else 1024
                whole.train_step(data[Bk][0], data[Bk][1], want_loss=False)
            torch.cuda.synchronize()
        a, b = sp.get_flat(), whole.get_flat()
        assert np.all(np.isfinite(a))
        assert np.array_equal(a, b), ("split-tile parameters differ from the whole-tile ones after step", it, float(np.abs(a - b).max()))
        checks += 1
print("%d split-tile fit steps (512- and 1024-graph shares) next to %d whole-tile steps of 4096 graphs on a second stream: "
      "parameters bit-identical to the whole-tile engine's at all %d checks; %.1f s" % (steps, steps * 2 // 3, checks, time.perf_counter() - t0))
