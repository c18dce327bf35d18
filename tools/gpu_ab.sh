# headline A/B on one box + the fused kernels' parity tests
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fused.py tests/test_gpu_fullsize.py tests/test_gpu_model.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do
python bench.py --no-cpu-baseline --no-dropin --no-other-workloads --min-seconds 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], 'fast', d['fast_path']['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
done
