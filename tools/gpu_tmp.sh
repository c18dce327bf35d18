cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export V2XGNN_LIB=$GRAFT_REPO_ROOT/globecom2020-resourceallocationgnn_amd/libv2xgnn_exp.so
for nt in 0 8 0 8; do V2X_WIDE_NT=$nt python bench.py --workload cfg4 --shard-of 8 --no-cpu-baseline --min-seconds 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('WIDE_NT=$nt', d['ms_per_step'], {k:v['avg_us'] for k,v in list(d['kernels'].items())[:9]})"; done
V2X_WIDE_NT=8 timeout 600 python -m pytest tests -m gpu -x -q -k "cfg3 or wide or 256 or 128" 2>&1 | grep -v amdgpu.ids | tail -2
