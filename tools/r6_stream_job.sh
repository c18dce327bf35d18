# round 6: k_mlp_stream at the shares -- the share steps with and without it (HIP-event kernel times)
O=gpurun_out/r6s; mkdir -p $O
if [ "$1" = "tests" ]; then python -m pytest tests/test_gpu_dense0_role.py -x -q -p no:cacheprovider 2>&1 | tail -4; fi
F="--no-cpu-baseline --no-dropin --no-other-workloads --no-fast-path --min-seconds 0.3"
for G in ${GS:-8 4 2}; do
  for S in 0 1; do
    echo "shard-of $G V2X_MLP_STREAM=$S: $(V2X_MLP_STREAM=$S bash tools/quick_bench.sh $F --shard-of $G)"
  done
done | tee $O/shares.txt
