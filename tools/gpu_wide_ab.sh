# configs[3] share (wide path) on one box + its parity tests
cd $GRAFT_REPO_ROOT
run() {
  env "$@" python bench.py --workload cfg4 --shard-of 8 --no-cpu-baseline --no-fast-path --no-dropin --no-other-workloads --min-seconds 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
}
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_shapes.py tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -x -q -k "cfg3 or 128 or 256 or wide" 2>&1 | tail -3
for i in 1 2; do
run V2X_WIDE_MERGE=1
done
run V2X_WIDE_MERGE=0
