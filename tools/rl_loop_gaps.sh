#!/bin/bash
# where the GPU sits idle for more than 2 ms during the timed DQN loop (kernel + copy trace): neighbours of every such gap
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/rl_loop_gaps
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --memory-copy-trace -d $OUT -o loop --output-format csv -- python $R/bench.py --workload cfg2loop --envs 50 --episodes 5 > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-200
python - "$OUT" <<'PY'
import csv, sys, glob
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-50:]))
for f in glob.glob(out + "/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", "")[-16:]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "k_dqn_targets" in r[2]]
t_first = rows[marks[2]][0]            # the timed loop starts after the two warm-up steps
print("launches", len(rows), "dqn steps", len(marks))
for i in range(marks[2], len(rows) - 1):
    g = rows[i + 1][0] - rows[i][1]
    if g > 1.5e6:
        step = sum(1 for m in marks if m <= i) - 2
        print("gap %.2f ms at +%.1f ms (after timed step %d): %s -> %s | then %s" % (g / 1e6, (rows[i][1] - t_first) / 1e6, step, rows[i][2], rows[i + 1][2],
              ", ".join(r[2][-24:] for r in rows[i + 2:i + 6])))
big = sorted(rows[marks[2]:], key=lambda r: r[0] - r[1])[:5]
print("longest launches:", [(r[2][-30:], round((r[1] - r[0]) / 1e3, 1)) for r in big])
PY
rm -f $OUT/*trace.csv $OUT/*agent_info.csv
