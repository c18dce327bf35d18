# configs[4] share: 320-row tiles with the weight image in LDS (V2X_RAGGED_SMALL=0) against 160-row tiles, 4-wave workgroups, weights from L2
F="--no-cpu-baseline --no-dropin --no-other-workloads --no-fast-path --min-seconds 0.5 --workload cfg5 --shard-of 8"
for S in 0 1 0 1; do
  echo "V2X_RAGGED_SMALL=$S: $(V2X_RAGGED_SMALL=$S bash tools/quick_bench.sh $F)"
done
