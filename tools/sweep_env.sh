#!/bin/bash
# bench (graph, 200 steps) under a list of "ENV=val[,ENV=val] [TF_BENCH_SET]" settings, one line each.
# usage: sweep_env.sh "TF_SHADE_STATIC_256=192" "TF_SHADE_STATIC_256=128;shade_wgs_beside_sort=480" ...
mkdir -p gpurun_out
for spec in "$@"; do
  envs="${spec%%;*}"; sets=""; [[ "$spec" == *";"* ]] && sets="${spec#*;}"
  ( for kv in ${envs//,/ }; do export "$kv"; done; export TF_BENCH_SET="$sets"
    python bench.py --steps 200 --warmup 20 --no-baselines ${BENCH_ARGS} 2>/dev/null | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$spec', l['ms_per_step'], l.get('eval',{}).get('value'))" )
done
