# the other workloads of profiles/: one JSON line each (no CPU leg, roofline section kept)
O=gpurun_out/r2j; mkdir -p $O
B="python bench.py --no-cpu-baseline"
$B --share-weights > $O/bench_shared.json 2>/dev/null
$B --workload cfg4 > $O/bench_cfg4.json 2>/dev/null
$B --workload cfg5 > $O/bench_cfg5.json 2>/dev/null
$B --batch 65536 --steps 20 --warmup 3 > $O/bench_b65536.json 2>/dev/null
$B --scaling strong > $O/bench_strong.json 2>/dev/null
V2X_FUSED=0 V2X_MLP_WG=0 $B > $O/bench_layerwise.json 2>/dev/null
V2X_MLP_WG=0 $B > $O/bench_mlp_split.json 2>/dev/null
V2X_FUSED_COMPL=0 $B > $O/bench_edge_gather.json 2>/dev/null
python bench.py --workload cfg0 > $O/bench_cfg0_episode.json 2>/dev/null
python bench.py --workload cfg0 --envs 10 > $O/bench_cfg0_episode_envs10.json 2>/dev/null
python bench.py --workload cfg2loop > $O/bench_cfg2loop.json 2>/dev/null
python bench.py --workload cfg2loop --envs 10 > $O/bench_cfg2loop_envs10.json 2>/dev/null
python tools/fused_phases.py > $O/fused_phases.txt 2>/dev/null
python tools/mlpwg_phases.py > $O/mlpwg_phases.txt 2>/dev/null
for f in $O/bench_*.json; do echo "$f: $(python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print(d.get('ms_per_step'), d.get('value'), d.get('unit'), (d.get('roofline') or {}).get('kernel'), (d.get('roofline') or {}).get('frac'), {k:v for k,v in d.items() if k in ('ms_per_train_step','episode_seconds','split')})")"; done
