#!/usr/bin/env python3
"""Per-kernel roofline table (markdown) of ANY bench.py workload from its JSON line (+ optional rocprofv3 PMC summaries):
    python tools/roofline_table2.py bench.json [fetch.txt write.txt mfma.txt] > profiles/r03_roofline_cfgX.md
Per kernel ROLE (HIP-event names of bench.py's `kernels`): launches per step, us per launch, algorithmic MB / GFLOP per
launch (bench.py models = SURVEY.md 8 d8 / d9 terms), their floors at 8 TB/s and 157.3 TFLOP/s, the fraction of the larger
floor.  Per kernel SYMBOL (rocprofv3): PMC HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, kB), MFMA-busy share."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def pmc(path):
    out = {}
    if not path or not os.path.exists(path):
        return out
    for line in open(path):
        m = re.match(r"^(\S.*?)\s{2,}(\S+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$", line.rstrip("\n"))
        if m and m.group(1).startswith("k_"):
            out.setdefault(m.group(1).strip(), {})[m.group(2)] = (int(m.group(3)), float(m.group(5)))
    return out


def main(argv):
    d = json.loads(open(argv[0]).read().strip().splitlines()[-1])
    cfg = d["config"]
    w = cfg["workload"]
    m = re.search(r"(\d+(?:-\d+)?) V2V links, feat_dim=(\d+), (\d+)-layer", w)
    links, F, L = m.group(1), int(m.group(2)), int(m.group(3))
    B = cfg["graphs_per_gpu"]
    ragged = "-" in links
    steps = 50
    print("# Per-kernel roofline: %s\n" % w)
    print("step: %.4f ms, %.3f M graph-instances/s, kernel path %s\n" % (d["ms_per_step"], d["value"] / 1e6, cfg.get("kernel_path")))
    if ragged:
        import numpy as np
        sizes = bench.synth_ragged(np.random.default_rng(1001), int(re.search(r"global batch (\d+)", w).group(1)), *[int(v) for v in links.split("-")])[0]
        n_sh = int(re.search(r"cut into (\d+) shard", w).group(1))
        sizes = sizes[:B] if n_sh > 1 else sizes           # shard 0 (contiguous, balanced by edges + nodes: close enough for a table)
        R, E = int(sizes.sum()), int((sizes * (sizes - 2)).sum())
        Bq, Nq = 1, R
    else:
        N = int(links)
        R, E, Bq, Nq = B * N, B * N * (N - 2), B, N
    print("| kernel (role) | launches/step | us/launch | alg. MB | alg. GFLOP | HBM floor us | MFMA floor us | frac of the larger floor |")
    print("|---|---|---|---|---|---|---|---|")
    tot = 0.0
    for k, v in d["kernels"].items():
        by = bench.algorithmic_bytes(k, Bq, Nq, F, E, L)
        fl = bench.algorithmic_flops(k, R, F, L)
        n = v["calls"] / steps
        t_h = by / 8e12 * 1e6 if by else 0.0
        t_m = fl / 157.3e12 * 1e6 if fl else 0.0
        tot += n * v["avg_us"]
        print("| %s | %g | %.1f | %s | %s | %.1f | %.1f | %s |" % (k, n, v["avg_us"], "%.1f" % (by / 1e6) if by else "-",
              "%.2f" % (fl / 1e9) if fl else "-", t_h, t_m, "%.2f" % (max(t_h, t_m) / v["avg_us"]) if max(t_h, t_m) > 0 else "-"))
    print("\nsum of kernels: %.0f us per step (HIP events around eager launches)." % tot)
    if len(argv) >= 4:
        f, wr, mf = pmc(argv[1]), pmc(argv[2]), pmc(argv[3])
        print("\n| kernel (symbol, rocprofv3) | launches | PMC HBM MB / launch | MFMA busy | WAIT_ANY | WAIT_INST_ANY | ACTIVE |")
        print("|---|---|---|---|---|---|---|")
        for k in sorted(set(f) | set(wr) | set(mf)):
            fk = f.get(k, {}).get("FETCH_SIZE", (0, 0.0))
            wk = wr.get(k, {}).get("WRITE_SIZE", (0, 0.0))
            c = mf.get(k, {})
            busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0.0))[1]
            sq = c.get("SQ_BUSY_CYCLES", (0, 0.0))[1]
            wc = c.get("SQ_WAVE_CYCLES", (0, 0.0))[1]
            pct = lambda name: "%.0f%%" % (100 * c[name][1] / wc) if wc and name in c else "-"
            print("| %s | %d | %.1f | %s | %s | %s | %s |" % (k, fk[0] or wk[0], (2 * fk[1] + wk[1]) * 1024 / 1e6,
                  "%.0f%%" % (100 * busy / (32 * sq)) if sq else "-", pct("SQ_WAIT_ANY"), pct("SQ_WAIT_INST_ANY"), pct("SQ_ACTIVE_INST_ANY")))
        print("\n(PMC: separate rocprofv3 passes, FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md; MFMA busy = "
              "SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES); the wait / active columns are shares of SQ_WAVE_CYCLES.)")


if __name__ == "__main__":
    main(sys.argv[1:])
