# A/B of the wide-path variants on one box (configs[3] share): environment switches of csrc/v2xgnn.hip
cd $GRAFT_REPO_ROOT
run() {
  env "$@" python bench.py --workload cfg4 --shard-of 8 --no-cpu-baseline --no-fast-path --min-seconds 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
}
for i in 1 2; do
run V2X_WIDE_MERGE=0
run V2X_WIDE_MERGE=1
run V2X_WIDE_MERGE=1 V2X_WIDE_FOLD=0
done
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_shapes.py tests/test_gpu_model.py -m gpu -x -q -k "cfg3 or 128 or 256 or wide" 2>&1 | tail -3
V2X_WIDE_FOLD=0 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_shapes.py -m gpu -x -q -k "cfg3 or 128 or 256 or wide" 2>&1 | tail -3
python -m pytest tests/test_gpu_shapes.py -m gpu -x -q -k out_of_step 2>&1 | tail -40
