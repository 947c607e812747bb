#!/bin/bash
# One captured training step's kernel timeline (start offset, duration) from a rocprofv3 kernel trace of bench.py, after a
# 200-step timing of the same build.  TF_BENCH_SET="attr=value,..." overrides model attributes (bench.py build_scene).
set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
python bench.py --steps 200 --warmup 20 --no-baselines 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', l['ms_per_step'])"
rm -rf gpurun_out/kt5
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt5 -- python bench.py --steps 40 --warmup 5 --no-baselines > /dev/null 2>&1
python tools/graph_trace.py $(find gpurun_out/kt5 -name '*kernel_trace.csv' | head -1)
