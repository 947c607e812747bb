#!/bin/bash
# rocprofv3 kernel trace of tools/bench_configs.py for one configuration (IMPL / EARLY / REPS / WAVES from the environment)
set -e
export TMPDIR=/tmp
CFG=${1:-C2_vm300}
OUT=gpurun_out/prof_sc
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o run -- python3 tools/bench_configs.py $CFG > $OUT/run.log 2> $OUT/run.err
python3 tools/summarize_prof.py stats $OUT/kt last=20 > $OUT/stats.csv
cat $OUT/run.log
cut -c1-150 $OUT/stats.csv
