# rows per workgroup of the graph layers' weight-gradient launch (k_wgrad_gnn): heavy roles x embed role
for c in "896 0" "704 2048" "704 4096" "768 2048" "768 4096" "896 2048" "704 1024" "640 4096"; do set -- $c; V2X_WG_CHUNK_GNN=$1 V2X_WG_CHUNK_EMBED=$2 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('gnn chunk $1 embed $2:', d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items() if k in ('k_wgrad_gnn','k_reduce_adam')})"; done
