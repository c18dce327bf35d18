#!/bin/bash
# device-side time of the DQN loop (configs[2], 50 simulators): rocprofv3 kernel statistics of bench.py --workload cfg2loop
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/rl_loop_prof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o loop --output-format csv -- python $R/bench.py --workload cfg2loop --envs 50 --episodes 5 > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-400
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.3f ms over %d launches" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print("%-70s calls %6s total %9.3f ms avg %8.2f us" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
rm -f $OUT/*kernel_trace.csv $OUT/*agent_info.csv
