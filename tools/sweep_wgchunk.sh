# rows per workgroup of the graph layers' weight-gradient launch (k_wgrad_gnn)
for c in 1024 512 640 704 768 896; do V2X_WG_CHUNK_GNN=$c python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('gnn chunk $c:', d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items() if k in ('k_wgrad_gnn','k_reduce_adam')})"; done
