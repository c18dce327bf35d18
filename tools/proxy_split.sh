# shares of the metric's global batch: whole-tile fused kernels (V2X_FUSED_SPLIT=0) against the split-tile ones (library's choice / forced K)
F="--no-cpu-baseline --no-dropin --no-other-workloads --no-fast-path --min-seconds 0.3"
timeout 300 python -m pytest tests/test_gpu_fused.py -x -q -k split 2>&1 | tail -3
for G in ${GS:-8 4 2}; do
  for K in ${KS:-0 -1}; do
    echo "shard-of $G V2X_FUSED_SPLIT=$K: $(V2X_FUSED_SPLIT=$K bash tools/quick_bench.sh $F --shard-of $G)"
  done
done
