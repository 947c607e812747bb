set -e
mkdir -p gpurun_out; export TMPDIR=/tmp
export TF_DP_FORCE_EXCHANGE=1
rm -rf gpurun_out/kt6
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt6 -- python bench.py --steps 40 --warmup 5 --no-baselines > /dev/null 2>&1
python tools/graph_trace.py $(find gpurun_out/kt6 -name '*kernel_trace.csv' | head -1)
