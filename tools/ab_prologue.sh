#!/bin/bash
# bench (graph) + one replay's kernel timeline.  usage: ab_prologue.sh [TF_BENCH_SET value]
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
export TF_BENCH_SET="$1"
python bench.py --steps 200 --warmup 20 --no-baselines 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', l['ms_per_step'])"
rm -rf gpurun_out/kt5
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt5 -- python bench.py --steps 40 --warmup 5 --no-baselines > /dev/null 2>&1
python tools/graph_trace.py $(find gpurun_out/kt5 -name '*kernel_trace.csv' | head -1)
