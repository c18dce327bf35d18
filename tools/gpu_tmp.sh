cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -k "agg or ragged or cfg3 or cfg4 or shapes or wide or configs" 2>&1 | grep -v amdgpu.ids | tail -5
for wl in cfg4 cfg5; do python bench.py --workload $wl --shard-of 8 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', d['ms_per_step'], d['value'], {k:v['avg_us'] for k,v in d['kernels'].items()})"; done
