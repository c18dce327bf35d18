# A/B of launch-structure switches at the shares of the global batch (ms per step + per-kernel us)
F="--no-cpu-baseline --no-dropin --no-other-workloads --no-fast-path --min-seconds 0.3"
for G in ${GS:-8 4}; do
  echo "shard-of $G default: $(bash tools/quick_bench.sh $F --shard-of $G)"
  echo "shard-of $G V2X_MLP_WG=0: $(V2X_MLP_WG=0 bash tools/quick_bench.sh $F --shard-of $G)"
  echo "shard-of $G V2X_MLP_WGS_PER_CU=1: $(V2X_MLP_WGS_PER_CU=1 bash tools/quick_bench.sh $F --shard-of $G)"
  echo "shard-of $G V2X_TWO_STREAMS=1: $(V2X_TWO_STREAMS=1 bash tools/quick_bench.sh $F --shard-of $G)"
done
