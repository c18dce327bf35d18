# A/B of experimental builds on one box: the shipped library against globecom2020-resourceallocationgnn_amd/libv2xgnn_exp*.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do for lib in libv2xgnn.so $(cd globecom2020-resourceallocationgnn_amd; ls libv2xgnn_exp*.so); do
V2XGNN_LIB=$GRAFT_REPO_ROOT/globecom2020-resourceallocationgnn_amd/$lib python bench.py --no-cpu-baseline --no-edge-gather --min-seconds 1 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"; done; done
