#!/bin/bash
# PMC pass (own run, --kernel-trace only) over tools/bench_configs.py: instruction mix and waits of the scatter kernels
set -e
export TMPDIR=/tmp
CFG=${1:-C2_vm300}
OUT=gpurun_out/pmc_sc
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS -d $OUT/a -o run -- python3 tools/bench_configs.py $CFG > $OUT/run.log 2> $OUT/run.err
python3 tools/summarize_prof.py pmc $OUT/a last=20 > $OUT/pmc_a.csv || true
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/b -o run -- python3 tools/bench_configs.py $CFG > $OUT/run.log 2> $OUT/run.err
python3 tools/summarize_prof.py pmc $OUT/b last=20 > $OUT/pmc_b.csv || true
grep -i "tile_scatter\|^name\|bin_scatter" $OUT/pmc_a.csv $OUT/pmc_b.csv | cut -c1-400
