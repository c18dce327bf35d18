# round 6: launch form (hipGraph replay against eager launches) -- the default line's choice and the DQN loops both ways
python bench.py --no-cpu-baseline --no-dropin --no-other-workloads --min-seconds 2 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('default:', d['ms_per_step'], d['value'], d['config']['launch'], d['config']['launch_probe_ms'], d['fast_path'] and d['fast_path']['ms_per_step'])"
for G in 8 4 2; do python bench.py --no-cpu-baseline --no-dropin --no-other-workloads --min-seconds 1 --shard-of $G 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('shard-of $G:', d['ms_per_step'], d['config']['launch'], d['config']['launch_probe_ms'])"; done
for wl in cfg4 cfg5; do python bench.py --no-cpu-baseline --workload $wl --shard-of 8 --min-seconds 1 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$wl:', d['ms_per_step'], d['config']['launch'], d['config']['launch_probe_ms'])"; done
for L in graph eager; do for E in 50 1; do
  for i in 1 2; do python bench.py --workload cfg2loop --envs $E --episodes 3 --launch $L 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg2loop envs $E launch $L:', d['ms_per_step'])"; done; done; done
