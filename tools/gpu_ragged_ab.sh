# ragged (configs[4] share) A/B on one box + its parity tests
cd $GRAFT_REPO_ROOT
run() {
  env "$@" python bench.py --workload $WL --shard-of 8 --no-cpu-baseline --no-fast-path --no-dropin --no-other-workloads --min-seconds 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$WL $*', d['ms_per_step'], d['config']['kernel_path']['graph_layers'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
}
python -m pytest tests/test_gpu_fused.py tests/test_gpu_fullsize.py tests/test_gpu_model.py tests/test_gpu_kernels.py tests/test_gpu_configs.py -m gpu -x -q -k "cfg4 or ragged or variable" 2>&1 | tail -5
for i in 1 2; do
WL=cfg5 run V2X_RAGGED_PLAN_FOLD=0
WL=cfg5 run V2X_RAGGED_PLAN_FOLD=1
done
