#!/usr/bin/env python3
"""profiles/hbm_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; kilobytes per dispatch).

    python tools/make_traffic_json.py <fetch.db> <write.db> > profiles/hbm_traffic.json

HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (KB -> B): on gfx950 FETCH_SIZE reports half of the bytes of a
wide coalesced read stream (MI355X_MICROARCH.md "HBM"); WRITE_SIZE calibrates exactly on this engine's kernels
(k_gemm_rows writes R*F*4 = 20,971,520 B and reports 20,481 KB)."""
import json
import re
import sqlite3
import sys

NAMES = [(r"k_agg<false>", "k_agg_fwd"), (r"k_agg<true>", "k_agg_bwd"),
         (r"k_gemm_rows<\d+, false, (false|true), false>", "k_node_fwd_embed"),
         (r"k_gemm_rows<\d+, true, true, false>", "k_node_fwd"), (r"k_gemm_rows<\d+, true, false, true>", "k_node_dgrad"),
         (r"k_mlp_fwd", "k_mlp_fwd"), (r"k_mlp_bwd", "k_mlp_bwd"), (r"k_mlp_train_wg", "k_mlp_train_wg"), (r"k_mlp_train", "k_mlp_train"), (r"k_wgrad<\d+, 0>", "k_wgrad_gnn"),
         (r"k_wgrad<\d+, 1>", "k_wgrad_dense"), (r"k_wgrad<\d+, 2>", "k_wgrad_all"),
         (r"k_agg_small<false", "k_agg_fwd"), (r"k_agg_small<true", "k_agg_bwd"), (r"k_reduce_adam", "k_reduce_adam"),
         (r"k_gnn_fwd_fused", "k_gnn_fwd_fused"), (r"k_gnn_bwd_fused", "k_gnn_bwd_fused"), (r"k_pack_weights", "k_pack_weights")]


def per_dispatch(path, counter):
    c = sqlite3.connect(path)
    agg = {}
    for name, val in c.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        name = name.replace("v2x::", "").replace("void ", "")
        for pat, mine in NAMES:
            if re.match(pat, name):
                a = agg.setdefault(mine, [0, 0.0])
                a[0] += 1
                a[1] += val
                break
    return {k: v[1] / v[0] for k, v in agg.items()}


def main(fetch_db, write_db):
    import hashlib
    import os
    lib = os.environ.get("V2XGNN_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                    "globecom2020-resourceallocationgnn_amd", "libv2xgnn.so"))
    f, w = per_dispatch(fetch_db, "FETCH_SIZE"), per_dispatch(write_db, "WRITE_SIZE")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out = {"lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(),     # bench.py reports these figures only for THIS binary
           "src_sha256": bench.src_sha256(),                                     # ... or a rebuild of the same sources
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python bench.py --no-graph",
           "formula": "bytes = (2*FETCH_SIZE_KB + WRITE_SIZE_KB) * 1024", "fetch_kb": f, "write_kb": w,
           "bytes_per_launch": {k: int((2 * f.get(k, 0.0) + w.get(k, 0.0)) * 1024) for k in sorted(set(f) | set(w))}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
