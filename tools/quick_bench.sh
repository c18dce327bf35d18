#!/bin/bash
# one-line A/B helper: ms/step + per-kernel average us of the default bench (HIP-event profile), env passed through
python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], {k:v['avg_us'] for k,v in (d.get('kernels') or {}).items()})"
