#!/usr/bin/env python3
"""Per-kernel roofline table (markdown) from a bench.py JSON line and the PMC traffic file:
    python tools/roofline_table.py profiles/r01d_bench.json profiles/hbm_traffic.json > profiles/r01d_roofline.md
Columns: measured time per launch, algorithmic bytes / flops per launch (bench.py models = SURVEY.md 8(d8)/(d9) terms),
their floors at 8 TB/s and 157.3 TFLOP/s, PMC HBM traffic, and the additive model  boundary + bytes/6 TB/s + flops/150 TF
that the overlap micro-benchmark (tools/overlapbench.hip) predicts for this chip."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main(bench_json, traffic_json):
    d = json.load(open(bench_json))
    t = json.load(open(traffic_json))["bytes_per_launch"] if traffic_json and os.path.exists(traffic_json) else {}
    B, N, F, L = 4096, 20, 64, 2
    E = B * N * (N - 2)
    R = B * N
    print("# Per-kernel roofline, headline workload (%s)\n" % d["config"]["workload"])
    print("step: %.4f ms, %.2f M graph-instances/s\n" % (d["ms_per_step"], d["value"] / 1e6))
    print("| kernel | launches/step | us/launch | alg. MB | alg. GFLOP | HBM floor us | MFMA floor us | roof (max) frac | PMC MB | additive model us |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    per_step = {"k_agg_fwd": 3, "k_agg_bwd": 3, "k_node_fwd": 2, "k_node_dgrad": 2}
    tot_meas = tot_model = 0.0
    for k, v in d["kernels"].items():
        by = bench.algorithmic_bytes(k, B, N, F, E, L)
        fl = bench.algorithmic_flops(k, R, F, L)
        us = v["avg_us"]
        n = per_step.get(k, 1)
        t_h = by / 8e12 * 1e6 if by else 0.0
        t_m = fl / 157.3e12 * 1e6 if fl else 0.0
        pm = t.get(k)
        real_bytes = pm if pm else (by or 0)
        model = 7.0 + real_bytes / 6e12 * 1e6 + (fl or 0) / 150e12 * 1e6
        tot_meas += n * us
        tot_model += n * model
        print("| %s | %d | %.1f | %s | %s | %.1f | %.1f | %.2f | %s | %.1f |" % (
            k, n, us, "%.1f" % (by / 1e6) if by else "-", "%.2f" % (fl / 1e9) if fl else "-", t_h, t_m,
            max(t_h, t_m) / us, "%.1f" % (pm / 1e6) if pm else "-", model))
    print("\nsum of kernels: measured %.0f us, additive model %.0f us (7 us per kernel boundary + PMC bytes at 6 TB/s + flops at "
          "150 TFLOP/s, no overlap)." % (tot_meas, tot_model))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
