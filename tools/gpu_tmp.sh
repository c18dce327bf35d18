cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for g in 8 4; do for e in 1 0; do V2X_MLP_WG=$e python bench.py --shard-of $g --no-cpu-baseline --no-edge-gather --min-seconds 0.5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MLP_WG=$e shard-of $g', d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"; done; done
