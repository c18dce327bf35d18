#!/bin/bash
# GPU-side timeline of the DQN loop (configs[2], 50 simulators): per train step, where the device is busy and where it waits for the host
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/rl_loop_tl
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --memory-copy-trace -d $OUT -o loop --output-format csv -- python $R/bench.py --workload cfg2loop --envs 50 --episodes 5 > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-300
python - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:]))
for f in glob.glob(out + "/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", "")))
rows.sort()
marks = [i for i, r in enumerate(rows) if "k_dqn_targets" in r[2]]
print("steps seen", len(marks))
steps = marks[-60:]                       # the last 60 train steps (annealed epsilon: every step scores)
gap_by = collections.Counter(); busy = 0; span = 0
for a, b in zip(steps, steps[1:]):
    seg = rows[a:b + 1]
    span += seg[-1][0] - seg[0][0]
    for x, y in zip(seg, seg[1:]):
        busy += x[1] - x[0]
        g = y[0] - x[1]
        if g > 0: gap_by[(x[2][-30:], y[2][-30:])] += g
n = len(steps) - 1
print("per step: span %.1f us, busy %.1f us, idle %.1f us" % (span / n / 1e3, busy / n / 1e3, (span - busy) / n / 1e3))
for (x, y), g in gap_by.most_common(14):
    print("  idle %7.1f us/step between %-32s and %s" % (g / n / 1e3, x, y))
PY
rm -f $OUT/*trace.csv $OUT/*agent_info.csv
