#!/bin/bash
# lock-step comparisons of the two PSNR-parity students (tests/psnr_event_diag.py) on the GPU box
set -e
mkdir -p gpurun_out/diag
run() { name=$1; shift; echo "== $name: $*"; timeout -k 10 400 python -m tests.psnr_event_diag "$@" --out gpurun_out/diag/$name.json > gpurun_out/diag/$name.log 2>&1; grep -E '"cut"|"event"' gpurun_out/diag/$name.log | cut -c1-400; }
run base --iters 400 --cuts 250 300 350 400
run swap_mask --iters 400 --cuts 250 300 350 400 --swap mask
run swap_params --iters 400 --cuts 250 300 350 400 --swap params
run no_mask --iters 400 --cuts 250 300 350 400 --no-mask
run no_up --iters 400 --cuts 250 300 350 400 --no-upsample
run eager_pair --iters 400 --cuts 250 300 350 400 --pair eager
run hip_pair --iters 400 --cuts 250 300 350 400 --pair hip
run seed6 --iters 400 --cuts 250 300 350 400 --seed 6
