#!/bin/bash
# One bench.py line per BASELINE configuration (1 GPU; C4 / C5 are meant for 2 / 8 GPUs: `--config C4 --gpus 2` etc. is the
# driver's call on a node that has them).  Replaces tools/bench_configs.py of round 2 (random targets, pre-training counts).
mkdir -p gpurun_out
for c in C1 C2 C3 C4 C5; do
  timeout -k 10 400 python bench.py --config $c --no-baselines --steps ${STEPS:-50} --warmup 10 > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err || { echo "$c failed"; tail -5 gpurun_out/bench_$c.err; }
  python - <<P
import json
try:
    d = json.load(open("gpurun_out/bench_$c.json"))
    print("$c", d["config"]["workload"]); print("   train %.3f ms/step = %.2f M rays/s (eager %.3f ms), eval %.2f M rays/s, workspace %.2f GiB, per ray %s" % (
        d["ms_per_step"], d["value"] / 1e6, d["config"]["eager_ms_per_step"] or 0, d.get("eval", {}).get("value", 0) / 1e6,
        d["config"]["training_workspace_GiB"], {k: round(v, 1) for k, v in d["config"]["per_ray"].items()}))
    print("   " + "  ".join("%s %.0fus(%.2f)" % (k[3:], v["ms_per_step"] * 1e3, v["frac"]) for k, v in d["kernels"].items()))
except Exception as e:
    print("$c: no line", e)
P
done
