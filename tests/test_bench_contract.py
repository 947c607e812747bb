"""bench.py prints ONE JSON line with the keys the driver reads (contract of the task statement) — short runs of both
modes; the baseline legs (cpu_baseline, rocm_eager_baseline) are skipped here, they take minutes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_bench_line_has_the_contract_keys(mode):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-baselines", "--steps", "6", "--warmup", "2",
                          "--mode", mode], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["unit"] == "rays/s" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 4096 * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert d["value"] > 1e6


def _bench(args, env=None, timeout=600):
    e = dict(os.environ)
    e.update(env or {})
    e.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                         timeout=timeout, cwd=ROOT, env=e)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_gpus_flag_starts_that_many_ranks_cpu():
    """`python bench.py --gpus 2` (no torchrun environment) must start 2 ranks itself — as child processes of a parent
    that never touches the GPU — and report the world size the process group saw.  --dry-run: rendezvous, barrier and
    MAX-over-ranks timing only (gloo), no GPU work, so this runs in the CPU container."""
    d = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"], {"TF_DIST_BACKEND": "gloo"}, timeout=300)
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["dry_run"] is True
    d = _bench(["--steps", "3", "--warmup", "1", "--dry-run"], timeout=120)
    assert d["n_gpus"] == 1


@pytest.mark.gpu
def test_gpus_flag_starts_that_many_ranks_gpu():
    """The real step with two ranks sharing the one GPU of the test box (gloo between them: RCCL refuses two ranks per
    device): split graphs around the gradient exchange, n_gpus == 2, global batch 2 x 4096."""
    d = _bench(["--gpus", "2", "--steps", "4", "--warmup", "2", "--no-baselines"], {"TF_DIST_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8192
    assert abs(d["value"] - 8192 * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
