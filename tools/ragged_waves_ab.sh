# configs[4] share with 8 (shipped), 12 and 16 waves per ragged workgroup (experimental builds under build_exp/: -DV2X_RG_WAVES=12 / 16)
F="--no-cpu-baseline --no-dropin --no-other-workloads --no-fast-path --min-seconds 0.5 --workload cfg5 --shard-of 8"
echo "8 waves: $(bash tools/quick_bench.sh $F)"
for W in 12 16; do
  echo "$W waves: $(V2XGNN_LIB=$PWD/build_exp/libv2xgnn_w$W.so bash tools/quick_bench.sh $F)"
done
