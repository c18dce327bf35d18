# configs[3] share: whole 128-row tiles only (V2X_WIDE_TAIL=0) against the last round as 64-row half tiles
F="--no-cpu-baseline --no-dropin --no-other-workloads --no-fast-path --min-seconds 1.0 --workload cfg4 --shard-of 8"
for T in 0 1 0 1; do
  echo "V2X_WIDE_TAIL=$T: $(V2X_WIDE_TAIL=$T bash tools/quick_bench.sh $F)"
done
