#!/bin/bash
# Collects the evidence committed under profiles/ (run on the GPU box through gpurun):
#   bench lines (train with baselines + psnr, eval), rocprofv3 kernel stats of the same bench command, PMC passes
#   (FETCH_SIZE, WRITE_SIZE, SQ counters — each in its own pass, --kernel-trace only, as the guide prescribes).
#   The profiled runs skip the bench's eval leg (--no-eval): every forward launch they see is a TRAINING forward.
set -e
R=${1:-r03}
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
python3 bench.py > gpurun_out/prof/${R}_bench_train.json 2> gpurun_out/prof/${R}_bench_train.err
python3 bench.py --mode eval --no-baselines > gpurun_out/prof/${R}_bench_eval.json 2> gpurun_out/prof/${R}_bench_eval.err
python3 bench.py --no-baselines --no-graph --steps 100 > gpurun_out/prof/${R}_bench_train_eager.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kt -o train -- python3 bench.py --no-baselines --no-eval --steps 40 --warmup 10 > gpurun_out/prof/${R}_bench_train_profiled.json 2> gpurun_out/prof/kt.err
python3 tools/summarize_prof.py stats gpurun_out/prof/kt last=40 > gpurun_out/prof/${R}_train_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d gpurun_out/prof/pmc_$c -o train -- python3 bench.py --no-baselines --no-eval --no-graph --steps 10 --warmup 3 > /dev/null 2> gpurun_out/prof/pmc_$c.err
  python3 tools/summarize_prof.py pmc gpurun_out/prof/pmc_$c last=10 > gpurun_out/prof/${R}_train_pmc_$(echo $c | tr A-Z a-z).csv
done
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 -d gpurun_out/prof/pmc_sq -o train -- python3 bench.py --no-baselines --no-eval --no-graph --steps 10 --warmup 3 > /dev/null 2> gpurun_out/prof/pmc_sq.err || true
python3 tools/summarize_prof.py pmc gpurun_out/prof/pmc_sq last=10 > gpurun_out/prof/${R}_train_pmc_sq.csv || true
rm -rf gpurun_out/prof/kt gpurun_out/prof/pmc_*/ 2>/dev/null || true
ls -la gpurun_out/prof
# the shading backward's products run on the bf16 pipe since round 3: its op count in a pass of its own (a counter name this
# rocprofv3 does not know fails only this pass)
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d gpurun_out/prof/pmc_bf -o train -- python3 bench.py --no-baselines --no-eval --no-graph --steps 10 --warmup 3 > /dev/null 2> gpurun_out/prof/pmc_bf.err || true
python3 tools/summarize_prof.py pmc gpurun_out/prof/pmc_bf last=10 > gpurun_out/prof/${R}_train_pmc_mfma_bf16.csv || true
rm -rf gpurun_out/prof/pmc_bf/ 2>/dev/null || true
