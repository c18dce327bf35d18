# configs[3] share: whole 128-row tiles only (V2X_WIDE_TAIL=0), the last round as 64-row half tiles (1, the default), and
# additionally every under-filled launch (Dense-0) as half tiles throughout (2)
F="--no-cpu-baseline --no-dropin --no-other-workloads --no-fast-path --min-seconds 1.0 --workload cfg4 --shard-of 8"
for T in 0 1 2 0 1 2; do
  echo "V2X_WIDE_TAIL=$T: $(V2X_WIDE_TAIL=$T bash tools/quick_bench.sh $F)"
done
