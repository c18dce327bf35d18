# round 6: rows per workgroup of the graph-layer weight-gradient launch at the shares of the global batch (k_wgrad runs ONE chunk
# per slot and role at <= 896 rows: 60 workgroups on 256 CUs)
F="--no-cpu-baseline --no-dropin --no-other-workloads --no-fast-path --min-seconds 0.3"
for G in ${GS:-8 4 2}; do
  echo "shard-of $G default: $(bash tools/quick_bench.sh $F --shard-of $G)"
  for R in ${ROWS:-64 128 256 512}; do
    echo "shard-of $G V2X_WG_CHUNK_GNN=$R: $(V2X_WG_CHUNK_GNN=$R bash tools/quick_bench.sh $F --shard-of $G)"
  done
done
