#!/usr/bin/env python3
"""bench.py — rays/s of the TensoRF ray-marching hot path on MI355X (BASELINE.json metric).

Workload (`config.workload`): by default BASELINE config 2 — TensorVMSplit, 300^3 grid, density_n_comp [16,16,16],
app_n_comp [48,48,48], app_dim 27, MLP_Fea shading, N = 1039 samples/ray, batches of 4096 rays drawn (seeded
permutation, SimpleSampler-style) from synthetic Blender-Lego 800x800 views, 'trained-like' field state
(no dataset or checkpoint exists offline).  `--config C1|C3|C4|C5` selects the other BASELINE
configurations at their full sizes (recon_amd.synthetic.baseline_scene: the scenes of tests/test_full_size.py), e.g.
`--config C4 --gpus 2` (LLFF NDC rays, random background, 2 ranks) or `--config C5 --gpus 8` (T&T 640^3).

A "step" is what train.py:323-376 does per iteration on this path: renderer(...) forward -> MSE ->
backward -> Adam step, on one 4096-ray batch per GPU (weak scaling: the global batch is 4096 x n_gpus,
gradients all-reduced over RCCL).  `--mode eval` times the forward-only renderer instead.

    python bench.py [--gpus N --steps K --warmup W --mode train|eval --config C1..C5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Extra objects: `roofline` (dominant kernel, durations measured live with HIP
events on the launch stream), `kernels` (all five kernels), `cpu_baseline` (the CPU oracle = plain-PyTorch
restatement of the reference, timed on this box's host cores) and `rocm_eager_baseline` (the same
restatement run with device='cuda' = the reference's PyTorch-ROCm path).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_MFMA_PEAK_TF = 157.3    # dense fp32 matrix peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", choices=["train", "eval"], default="train")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--grid", type=int, default=300)
    ap.add_argument("--views", type=int, default=3)
    ap.add_argument("--config", choices=["C1", "C2", "C3", "C4", "C5"], default="C2",
                    help="BASELINE.json configuration (C2 = the headline; the others are recon_amd.synthetic.baseline_scene)")
    ap.add_argument("--no-psnr", action="store_true", help="skip the equal-iterations PSNR legs")
    ap.add_argument("--no-eval", action="store_true",
                    help="train mode: skip the eval leg (profiling runs: every forward launch is then a TRAINING forward)")
    ap.add_argument("--no-baselines", action="store_true", help="skip the CPU / ROCm-eager baseline legs")
    ap.add_argument("--no-graph", action="store_true", help="drive the train step eagerly instead of replaying a hipGraph")
    ap.add_argument("--host-inputs", action="store_true",
                    help="rays / targets stay in (pinned) host memory; every step gathers its batch on the CPU and copies "
                         "it over PCIe like train.py:297-298 (diagnostic: the PCIe-inclusive rate, never the headline)")
    ap.add_argument("--profile-eager", action="store_true",
                    help="eager modes: keep the per-kernel HIP events inside the timed region (for rocprofv3 runs)")
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam(fused=True) instead of the one-launch tf_adam_step")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / timing protocol only (no GPU work): what the CPU test of `--gpus N` runs")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks ourselves, as CHILD
    processes of a parent that never touches the GPU (a process that has initialised HIP must not exec another
    program), one rank per GPU through torch.distributed.run, and relay rank 0's JSON line.  Returns the exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["TF_BENCH_CHILD"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    for l in proc.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if proc.returncode == 0 and len(lines) == 1:
        print(lines[0])
        return 0
    print(f"bench.py: the {args.gpus}-rank run failed (rc {proc.returncode}, {len(lines)} JSON lines)", file=sys.stderr)
    return proc.returncode or 1


SCENES = {"C1": "C1_vm128", "C2": "C2_vm300", "C3": "C3_cp300_mlp", "C4": "C4_ndc", "C5": "C5_tt640"}
LABELS = {"C1": "Lego 800^2 @ 128^3 grid", "C3": "Lego 800^2 @ 300^3 grid, TensorCP", "C4": "LLFF-like NDC rays @ 300^3-equivalent grid",
          "C5": "T&T-like rays @ 640^3-equivalent grid"}


def build_scene(recon, dev, args, need_rays, seed=0):
    """(model, rays, targets, n_samples, reso, ndc_ray, white_bg): the configuration's 'trained-like' field and rays
    resident in HBM; targets = the field's own rendering plus zero-mean noise, so the gradients are non-trivial while
    the trained-like state (and with it the per-ray sample counts of the workload) stays put over the run."""
    from recon_amd import synthetic as S
    if args.config == "C2":
        torch.manual_seed(seed)
        aabb = torch.tensor(S.LEGO_AABB, device=dev)
        reso = recon.N_to_reso(args.grid ** 3, aabb)
        model = recon.TensorVMSplit(S.lego_args(), aabb, reso, S.LEGO_NEAR_FAR, dev)
        S.make_trained_like(model, recon.AlphaGridMask)
        n_samples = min(int(1e6), recon.cal_n_samples(reso, 0.5))        # train.py:208
        rays, ndc, white = S.blender_rays(args.views), False, True
    else:
        name = SCENES[args.config]
        if args.config == "C3" and args.mode == "eval":
            name = "C3_cp300_sh"         # BASELINE config 3 names the SH head; it cannot be trained (nor can the reference's)
        model, rays, n_samples, ndc, white = S.baseline_scene(name, dev, n_rays=min(max(need_rays, 1 << 16), 1 << 20),
                                                              views=args.views)
        reso = model.gridSize.tolist()
    for kv in filter(None, os.environ.get("TF_BENCH_SET", "").split(",")):     # tuning runs: model attribute overrides
        k, v = kv.split("=")
        if not hasattr(model, k):
            raise SystemExit(f"TF_BENCH_SET: the model has no attribute {k!r}")
        setattr(model, k, type(getattr(model, k))(int(v)))
    if not ndc:
        keep = S.bbox_hit_mask(rays, model.aabb.cpu())                    # filtering_rays(bbox_only=True), train.py:291
        rays = rays[keep]
    rays = rays.to(dev)
    with torch.no_grad():
        teacher = recon.OctreeRender_trilinear_fast(rays, model, None, chunk=4096, N_samples=n_samples, white_bg=white,
                                                    ndc_ray=ndc, device=dev)[0]
    g = torch.Generator().manual_seed(seed + 1)
    targets = (teacher + 0.1 * torch.randn(rays.shape[0], 3, generator=g).to(dev)).clamp(0, 1)
    return model, rays, targets, n_samples, reso, ndc, white


def kernel_table(events, stats, cfg, n_steps):
    """Per kernel: average launch duration, launches per step, time per step, and the roofline fraction of its
    ALGORITHMIC bytes / flops per launch (SURVEY §8d figures; R rays, S_b in-box / S_s density / S_a shaded samples per
    batch, C_d / C_a = total density / appearance components):
      march_forward      40 R + 32 S_b + 24 C_d S_s                          (rays, alpha cells + positions, density taps)
      shade_forward      24 C_a S_a bytes, 2 (in_c F + F^2 + 3 F + C_a app_dim) S_a flop
      composite          16 R + 16 S_a
      march_backward     8 S_s + 16 S_s                                      (saved features re-read, scatter entries written)
      shade_backward     saved rows (X, H1, H2, V) + dV rows: 4 (in_c + 2 F + 2 C_a) S_a + 24 S_a bytes, 2 x the forward flops
      binned_scatter     taps of the OTHER factor re-read for the product rule, 24 C S, + the gradient rows / entries
                         consumed: density 16 S_s, appearance (4 C_a + 12) S_a
      binned_sort_pair   count + fill passes over both coordinate lists, one int per (entry, key): (24 + 4 kpe) S each
      adam               28 B per parameter element whose 256-float piece has ever had a gradient + 4 B (gradient) for the rest
    A kernel's `frac` is against the HBM peak (bytes) or the fp32-MFMA peak (flops), whichever is larger."""
    R, Sb, Ss, Sa = stats["rays"], stats["bbox"], stats["density"], stats["shaded"]
    cd, ca = cfg["cd"], cfg["ca"]
    F, in_c = cfg["featureC"], cfg["in_c"]
    mlp_flops = 2 * (in_c * F + F * F + F * 3) + 2 * ca * cfg["app_dim"]
    bwd_flops = 2 * mlp_flops                  # weight-gradient + input-gradient GEMMs (the forward's rows are saved, not
    #                                            recomputed: round 1 re-ran the forward in the kernel and counted 3 x)
    algo = {
        "tf_march_forward": (R * 40 + 32 * Sb + 24 * cd * Ss, 0.0),
        "tf_shade_forward": (24 * ca * Sa, mlp_flops * Sa),
        "tf_composite_forward": (R * 16 + 16 * Sa, 0.0),
        "tf_march_backward": (8 * Ss + 16 * Ss, 0.0),
        "tf_shade_backward": ((4 * (in_c + 2 * F + 2 * ca) + 24) * Sa, bwd_flops * Sa),
        "tf_binned_scatter_density": (24 * cd * Ss + 16 * Ss, 0.0),
        "tf_binned_scatter_app": (24 * ca * Sa + (4 * ca + 12) * Sa, 0.0),
        "tf_binned_sort_pair": ((24 + 4 * cfg["kpe_d"]) * Ss + (24 + 4 * cfg["kpe_a"]) * Sa, 0.0),
        "tf_adam_step": (cfg["adam_bytes"], 0.0),
    }
    out = {}
    for name, pairs in events.items():
        ms = sum(a.elapsed_time(b) for a, b in pairs) / max(len(pairs), 1)
        by, fl = algo.get(name, (0.0, 0.0))
        gbps, tf = (by / ms / 1e6, fl / ms / 1e9) if ms > 0 else (0.0, 0.0)
        per_step = len(pairs) / max(n_steps, 1)
        out[name] = {"avg_ms": ms, "launches_per_step": per_step, "ms_per_step": ms * per_step, "algo_bytes": by,
                     "algo_flops": fl, "GBps": gbps, "TFLOPps": tf,
                     "bound": "mfma" if tf / FP32_MFMA_PEAK_TF > gbps / HBM_PEAK_GBS else "hbm",
                     "frac": max(tf / FP32_MFMA_PEAK_TF, gbps / HBM_PEAK_GBS)}
    return out


def pmc_traffic(kernel_key):
    """HBM-side bytes per launch of one kernel from the committed rocprofv3 PMC summaries (separate FETCH_SIZE and
    WRITE_SIZE passes of this same command, profiles/r*_train_pmc_*.csv): (2*FETCH_SIZE + WRITE_SIZE) KB — the
    factor 2 is the gfx950 correction for wide (16 B/lane) loads of MI355X_MICROARCH.md 'HBM'.  None when the
    summaries are not there."""
    import csv
    import glob
    vals = {}
    for kind in ("fetch", "write"):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_train_pmc_{kind}_size.csv")))
        if not files:
            return None
        for row in csv.DictReader(open(files[-1])):
            if kernel_key in row["kernel"]:
                vals[kind] = float(row["avg_value"])
    if len(vals) != 2:
        return None
    return (2.0 * vals["fetch"] + vals["write"]) * 1024.0


def cpu_model_name():
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                return l.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def oracle_baseline(model, rays_cpu, targets_cpu, n_samples, device, mode, steps, warmup, sweep_threads=None, ndc=False,
                    white=True):
    """The plain-PyTorch restatement of the reference path (oracle/ref_torch.py) as a timed baseline.
    sweep_threads: candidate torch thread counts (CPU leg) — one step each after the warm-up, the fastest count is
    then used for the timed steps (oversubscribing a big host with one thread per core is 10x slower than 8-32)."""
    from oracle import ref_torch as R
    dev = torch.device(device)
    cfg = R.FieldCfg(model=type(model).__name__, aabb=model.aabb.detach().to(dev), gridSize=model.gridSize.tolist(),
                     near_far=model.near_far, step_ratio=model.step_ratio, fea2denseAct=model.fea2denseAct,
                     density_n_comp=model.density_n_comp, app_n_comp=model.app_n_comp, app_dim=model.app_dim,
                     density_shift=model.density_shift, distance_scale=model.distance_scale,
                     shadingMode=model.shadingMode, pos_pe=model.pos_pe, view_pe=model.view_pe, fea_pe=model.fea_pe,
                     featureC=model.featureC).finalize()
    cfg.alpha_volume = model.alphaMask.alpha_volume[0, 0].to(dev)
    cfg.alpha_aabb = model.alphaMask.aabb.to(dev)
    params = {k: v.detach().to(dev).contiguous().clone().requires_grad_(mode == "train")
              for k, v in model.state_dict().items()}
    opt = torch.optim.Adam(list(params.values()), lr=1e-3, betas=(0.9, 0.99)) if mode == "train" else None
    sweep = list(sweep_threads or [])
    B = rays_cpu.shape[0] // (steps + warmup + len(sweep))
    times, sweep_times = [], {}
    for i in range(steps + warmup + len(sweep)):
        r = rays_cpu[i * B:(i + 1) * B].to(dev)
        t = targets_cpu[i * B:(i + 1) * B].to(dev)
        if dev.type == "cuda":
            torch.cuda.synchronize()
        in_sweep = warmup <= i < warmup + len(sweep)
        if in_sweep:
            torch.set_num_threads(sweep[i - warmup])
        elif sweep and i == warmup + len(sweep):
            torch.set_num_threads(min(sweep_times, key=sweep_times.get))
        t0 = time.perf_counter()
        if mode == "train":
            rgb, _, _ = R.render_rays(cfg, params, r, None, white_bg=white, is_train=True, ndc_ray=ndc, n_samples=n_samples)
            loss = torch.mean((rgb - t) ** 2)
            opt.zero_grad()
            loss.backward()
            opt.step()
        else:
            with torch.no_grad():
                R.render_rays(cfg, params, r, None, white_bg=white, is_train=False, ndc_ray=ndc, n_samples=n_samples)
        if dev.type == "cuda":
            torch.cuda.synchronize()
        if in_sweep:
            sweep_times[sweep[i - warmup]] = time.perf_counter() - t0
        elif i >= warmup:
            times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    if sweep:
        return B / med, B, len(times), {k: B / v for k, v in sweep_times.items()}
    return B / med, B, len(times)


def dry_run(args, world, rank, backend):
    """The contract's protocol without the hot path: barrier, timed region, MAX over ranks, one line from rank 0."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if backend != "nccl" or not torch.cuda.is_available() else "nccl")
        dist.barrier()
    t0 = time.perf_counter()
    n = torch.zeros(1)
    for _ in range(args.steps):
        n += 1
    if world > 1:
        dist.barrier()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    seen = torch.ones(1)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(seen)
    if rank == 0:
        print(json.dumps({"metric": f"rays/sec ({args.mode}), Lego 800^2 @ {args.grid}^3 grid", "value": None,
                          "unit": "rays/s", "n_gpus": int(seen.item()), "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": float(tt) / max(args.steps, 1) * 1e3, "dry_run": True}))
    if world > 1:
        dist.destroy_process_group()


class _stdout_to_stderr:
    """RCCL prints a version banner on stdout when its first communicator comes up; the contract is ONE JSON line on
    stdout, so file descriptor 1 points at stderr while the process group initialises and runs its first collective."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not (world == 1 and args.gpus <= 1):
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)", file=sys.stderr)
        sys.exit(2)
    if args.dry_run:
        return dry_run(args, world, int(os.environ.get("RANK", "0")), os.environ.get("TF_DIST_BACKEND", "nccl"))
    # everything but the final JSON line goes to stderr: the library under test prints progress lines like the reference
    # does (alpha-mask statistics, "grid resized ..."), RCCL prints a banner, and the contract is ONE line on stdout
    guard = _stdout_to_stderr()
    guard.__enter__()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # TF_DIST_BACKEND=gloo rehearses the N>1 code path with several ranks sharing one GPU (RCCL refuses that);
    # the driver's runs use the default, "nccl" (= RCCL), one rank per GPU
    backend = os.environ.get("TF_DIST_BACKEND", "nccl")
    local = local % max(torch.cuda.device_count(), 1) if backend != "nccl" else local
    if world > 1 or int(os.environ.get("TF_DP_FORCE_EXCHANGE", "0")):     # the latter: 1-rank rehearsal of the N > 1 path
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        with _stdout_to_stderr():
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            else:
                dist.init_process_group(backend)
            warm = torch.zeros(1, device=torch.device("cuda", local))
            dist.all_reduce(warm)                 # brings the communicator up (and its banner out) now
            torch.cuda.synchronize()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    import recon_amd
    from recon_amd import parallel
    B = args.batch
    # the graphed train step needs a few eager steps of its own before it is captured (GraphedTrainStep: warm-up,
    # early-sort decision, capture — once per background outcome without a white background): they are set-up, run
    # before the W warm-up steps whatever W is
    n_setup = 8 if (args.mode == "train" and not args.no_graph) else 3      # eager modes: workspaces, weight copies
    n_steps = args.steps + args.warmup + n_setup
    model, rays, targets, n_samples, reso, ndc, white = build_scene(recon_amd, dev, args, B * world * n_steps)
    g = torch.Generator().manual_seed(20211202)
    perm = torch.randperm(rays.shape[0], generator=g)
    need = B * world * n_steps
    if perm.numel() < need:
        perm = perm.repeat((need + perm.numel() - 1) // perm.numel())
    perm = perm[:need].view(n_steps, B * world).to(dev)

    # train.py:272-273 (same optimizer, same groups / learning rates, same update rule): FusedAdam applies it in
    # one launch (tf_adam_step); --torch-adam uses torch's multi-tensor implementation instead
    use_graph = args.mode == "train" and not args.no_graph

    def make_opt(capturable):
        groups = model.get_optparam_groups(0.02, 1e-3)
        if args.torch_adam:
            return torch.optim.Adam(groups, betas=(0.9, 0.99), fused=True, capturable=capturable)
        return recon_amd.FusedAdam(groups, betas=(0.9, 0.99))

    opt = make_opt(use_graph)
    graphed = recon_amd.GraphedTrainStep(model, opt, B, n_samples, ndc_ray=ndc, white_bg=white) if use_graph else None
    model.lazy_sample_count = True   # the renderer's 6th return value syncs only when read (train.py never reads it)
    parallel.enable_overlapped_exchange(model)
    renderer = recon_amd.OctreeRender_trilinear_fast

    def train_step(i):
        ids = parallel.shard_ids(perm[i], rank, world)
        rays_train, rgb_train = rays[ids], targets[ids]

        def fwd_bwd():
            rgb_map, _, depth_map, _, _, n = renderer(rays_train, model, None, chunk=B, N_samples=n_samples, white_bg=white,
                                                       ndc_ray=ndc, device=dev, is_train=True)
            loss = torch.mean((rgb_map - rgb_train) ** 2)
            opt.zero_grad()
            loss.backward()
            return loss

        loss = model.retry_on_overflow(fwd_bwd)       # (a batch that outgrows the right-sized workspace is repeated)
        parallel.finish_gradient_exchange(model)      # (its density bucket left during the backward)
        if hasattr(opt, "kernel_events"):
            opt.kernel_events = model.kernel_events      # (FusedAdam brackets its launch itself)
        opt.step()
        return loss

    def eval_step(i):
        ids = parallel.shard_ids(perm[i], rank, world)
        with torch.no_grad():
            renderer(rays[ids], model, None, chunk=B, N_samples=n_samples, white_bg=white, ndc_ray=ndc, device=dev,
                     is_train=False)

    def graph_step(i):
        return graphed.step(rays, targets, parallel.shard_ids(perm[i], rank, world))

    if args.host_inputs:
        assert use_graph, "--host-inputs is implemented for the graphed train step"
        rays_h, targets_h, perm_h = rays.cpu().pin_memory(), targets.cpu().pin_memory(), perm.cpu()
        stage_r = [torch.empty(B, 6).pin_memory() for _ in range(2)]      # double-buffered: the copy of step i may
        stage_t = [torch.empty(B, 3).pin_memory() for _ in range(2)]      # still be in flight while step i+1 gathers
        stage_ev = [torch.cuda.Event(), torch.cuda.Event()]

        def graph_step(i):      # noqa: F811  (train mode, graphed)
            ids = parallel.shard_ids(perm_h[i], rank, world)
            b = i & 1
            stage_ev[b].synchronize()
            torch.index_select(rays_h, 0, ids, out=stage_r[b])
            torch.index_select(targets_h, 0, ids, out=stage_t[b])
            out = graphed.step(stage_r[b], stage_t[b])
            stage_ev[b].record()
            return out

    step = (graph_step if use_graph else train_step) if args.mode == "train" else eval_step
    # the scene set-up leaves ~10^6 long-lived Python objects behind; without this the cyclic collector re-walks
    # them every few steps of the eager paths (measured: 3 ms pauses on a 0.2 ms step)
    import gc
    gc.collect()
    gc.freeze()
    torch.manual_seed(1234 + rank)
    for i in range(n_setup + args.warmup):
        step(i)
    # the timed region carries no instrumentation (HIP events between launches and the sample-count reductions cost
    # ~40 us on a 0.2 ms eval step); per-kernel durations and sample statistics come from a separate pass below
    timed_eager = not use_graph and not args.profile_eager
    model.kernel_events = None if (use_graph or timed_eager) else {}
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = {"rays": 0, "bbox": 0, "density": 0, "shaded": 0}
    ctr_sum = torch.zeros(3, dtype=torch.int64, device=dev)
    for i in range(n_setup + args.warmup, n_steps):
        step(i)
        if args.profile_eager and not use_graph:
            ctr_sum += model.last["ws"].counters2d[:, :3].sum(0)
    host_issue = time.perf_counter() - t0        # host time to enqueue the timed steps (diagnostic: host- vs GPU-bound)
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist.is_initialized():
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)
    if args.mode == "train":
        model.check_scatter_status()     # raises if a scatter kernel of the timed steps refused an out-of-range position
    events = model.kernel_events
    model.kernel_events = None
    eager_ms = None
    n_e = args.steps
    if use_graph or timed_eager:
        # per-kernel durations: an eager pass of the SAME step (same kernels, same batch shapes) run right here,
        # bracketed launch by launch with HIP events on the launch stream (graph replays cannot carry events).  The
        # drop-in eager step itself (`eager_ms_per_step`: what a maintainer who only swaps the imports gets) is timed
        # first, WITHOUT the events — they cost ~40 % on this host-bound loop
        early = model.early_sort
        if use_graph:
            opt, model.static_jitter = make_opt(False), None
            # the drop-in loop runs the library's default ('auto'; TF_EARLY_SORT=0 / 1 pins it)
            model.early_sort = {"0": False, "1": True}.get(os.environ.get("TF_EARLY_SORT", "auto"), 'auto')
            model._last_sample_counts = None
        inst_step = train_step if args.mode == "train" else eval_step
        n_e = min(20, args.steps)
        for i in range(3):
            inst_step(i)
        if use_graph:
            # (the loop is host-paced — the renderer's float(num_valid) synchronises every step, renderer.py:24 — and a
            # shared host makes 20-step timings scatter by 20 %: the median of five blocks is reported)
            blocks = []
            for b in range(5):
                torch.cuda.synchronize()
                te = time.perf_counter()
                for i in range(n_e):
                    inst_step((args.warmup + b * n_e + i) % n_steps)
                torch.cuda.synchronize()
                blocks.append((time.perf_counter() - te) / n_e * 1e3)
            eager_ms = sorted(blocks)[2]
            print(f"[bench] eager blocks {['%.3f' % b for b in blocks]} ms/step; sorts beside the forward: "
                  f"{model.last.get('sorted_on') is not None} (early_sort = {model.early_sort!r})", file=sys.stderr)
            # per-kernel table: the captured step's own launches (its sorts run beside the shading kernel on a second
            # stream; here they are issued on ONE stream, so that every kernel is timed alone)
            model.early_sort, model._sort_inline = early, True
            for i in range(2):
                inst_step(i)
        else:
            eager_ms = elapsed / args.steps * 1e3
        model.kernel_events = {}
        torch.cuda.synchronize()
        for i in range(n_e):
            inst_step(args.warmup + i)
            ctr_sum += model.last["ws"].counters2d[:, :3].sum(0)
        torch.cuda.synchronize()
        events = model.kernel_events
        model.kernel_events = None

    if rank == 0:
        k = args.steps
        c = ctr_sum.tolist()
        c = [v * k / n_e for v in c]               # counters were summed over n_e instrumented steps
        stats = {"rays": B, "shaded": c[0] / k, "density": c[1] / k, "bbox": c[2] / k}
        import ctypes as C
        cp = type(model).__name__ == "TensorCP"
        lib = recon_amd._hip.lib()
        kpe = [int(lib.tf_bin_keys_per_entry(int(cp), C.byref((C.c_int * 3)(*([cc[0]] * 3 if cp else cc))))) if args.mode == "train"
               else 0 for cc in (model.density_n_comp, model.app_n_comp)]
        n_param = sum(p.numel() for p in model.parameters())
        adam_bytes = 28.0 * n_param
        if getattr(opt, "_touched", None) is not None:      # FusedAdam reads moments / parameters only where a gradient ever was
            words = opt._touched.cpu().numpy().astype("uint32")
            pieces = int(sum(bin(int(w)).count("1") for w in words))
            adam_bytes = 28.0 * 256 * pieces + 4.0 * max(n_param - 256 * pieces, 0)
        cfg = dict(cd=model.density_n_comp[0] if cp else sum(model.density_n_comp),
                   ca=model.app_n_comp[0] if cp else sum(model.app_n_comp), app_dim=model.app_dim,
                   featureC=model.featureC if model.shadingMode not in ("SH", "RGB") else 0,
                   in_c=getattr(model.renderModule, "in_mlpC", 0), kpe_d=kpe[0], kpe_a=kpe[1], adam_bytes=adam_bytes)
        kt = kernel_table(events, stats, cfg, n_e)
        # The dominant kernel = the kernel FUNCTION with the largest share of a step, all its launches together: the two
        # tf_binned_scatter calls (density, appearance) are launches of one kernel, bin_scatter_kernel
        groups = {n: [n] for n in kt if not n.startswith("tf_binned_scatter_")}
        sc = [n for n in kt if n.startswith("tf_binned_scatter_")]
        if sc:
            groups["tf_binned_scatter (density + appearance launches of bin_scatter_kernel)"] = sc
        dom = max(groups, key=lambda g: sum(kt[n]["ms_per_step"] for n in groups[g]))
        mem = groups[dom]
        ms = sum(kt[n]["ms_per_step"] for n in mem)
        launches = sum(kt[n]["launches_per_step"] for n in mem)
        by = sum(kt[n]["algo_bytes"] * kt[n]["launches_per_step"] for n in mem)
        fl = sum(kt[n]["algo_flops"] * kt[n]["launches_per_step"] for n in mem)
        gbps, tf = (by / ms / 1e6, fl / ms / 1e9) if ms > 0 else (0.0, 0.0)
        mf = tf / FP32_MFMA_PEAK_TF > gbps / HBM_PEAK_GBS
        roof = {"kernel": dom, "bound": "mfma" if mf else "hbm", "achieved": tf if mf else gbps,
                "peak": FP32_MFMA_PEAK_TF if mf else HBM_PEAK_GBS, "unit": "TFLOP/s" if mf else "GB/s",
                "frac": max(tf / FP32_MFMA_PEAK_TF, gbps / HBM_PEAK_GBS), "traffic": None,
                "avg_launch_ms": ms / max(launches, 1e-9), "launches_per_step": launches, "ms_per_step": ms,
                "note": "achieved = algorithmic bytes (flops) per launch / average launch duration (HIP events, eager pass); "
                        "traffic = (2 FETCH_SIZE + WRITE_SIZE) KB per launch from the committed PMC passes"}
        key = {"tf_shade_backward": "shade_backward_kernel", "tf_shade_forward": "shade_forward_",
               "tf_march_forward": "march_forward_kernel", "tf_march_backward": "march_backward_kernel",
               "tf_adam_step": "adam_kernel"}.get(dom, "bin_scatter_kernel" if sc and mem == sc else dom)
        roof["traffic"] = pmc_traffic(key)
        value = B * world * k / elapsed
        # whole-step view against the HBM roofline: algorithmic bytes per ray of the forward, plus — in
        # training — the taps re-read and the gradient taps written by the backward and the optimizer's 7 streams of
        # every parameter, amortised over the batch
        cd_, ca_ = cfg["cd"], cfg["ca"]
        per = {kk: stats[kk] / B for kk in ("bbox", "density", "shaded")}
        b_fwd = 40.0 + 32.0 * per["bbox"] + 24.0 * cd_ * per["density"] + 24.0 * ca_ * per["shaded"]
        b_ray = b_fwd if args.mode != "train" else \
            b_fwd + 2.0 * (24.0 * cd_ * per["density"] + 24.0 * ca_ * per["shaded"]) + 28.0 * n_param / B
        step_hbm = {"algo_bytes_per_ray": b_ray, "achieved_GBps": b_ray * value / world / 1e9, "peak_GBps": HBM_PEAK_GBS,
                    "frac": b_ray * value / world / 1e9 / HBM_PEAK_GBS,
                    "note": "per GPU; the factor tensors are L2 / Infinity-Cache resident, so this is algorithmic traffic, "
                            "not HBM traffic (see roofline.traffic for the dominant kernel's measured bytes)"}
        line = {
            "metric": f"rays/sec ({args.mode}), " + (f"Lego 800^2 @ {args.grid}^3 grid" if args.config == "C2" else LABELS[args.config]),
            "value": value, "unit": "rays/s", "n_gpus": dist.get_world_size() if dist.is_initialized() and not parallel.FORCE_EXCHANGE else world, "steps": k, "warmup": args.warmup,
            "ms_per_step": elapsed / k * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE config {args.config[1]}: {type(model).__name__} {list(reso)} grid, "
                                   f"{list(model.density_n_comp)}/{list(model.app_n_comp)} comps, {model.shadingMode}, N={n_samples}, "
                                   + ("NDC rays, random background, " if ndc else "")
                                   + f"{B}-ray batch per GPU, {args.mode} step"
                                   + (" = fwd+bwd+Adam" if args.mode == "train" else " = renderer forward"),
                       "mode": args.mode, "batch_per_gpu": B, "global_batch": B * world, "n_samples": n_samples,
                       "per_ray": {"in_bbox": stats["bbox"] / B, "density": stats["density"] / B,
                                   "shaded": stats["shaded"] / B},
                       "parallelism": f"ray-sharded dp{world}", "lazy_sample_count": True,
                       "launch": ("one hipGraph replay per step" if not graphed.split else
                                  "3 hipGraph replays per step (backward to the density gradients | shading backward | Adam) with "
                                  "the two RCCL all-reduces of the gradient buckets between them")
                                 if use_graph else "eager",
                       "optimizer": "torch.optim.Adam(fused)" if args.torch_adam else "Adam, one launch (tf_adam_step)",
                       "inputs": "host, gathered on the CPU + H2D per step" if args.host_inputs else "resident in HBM",
                       "eager_ms_per_step": eager_ms, "host_issue_ms_per_step": host_issue / k * 1e3,
                       "training_workspace_GiB": round(model.workspace_bytes() / 2 ** 30, 3)},
            "roofline": roof,
            "step_hbm_roofline": step_hbm,
            "kernels": {n: {"avg_ms": round(v["avg_ms"], 5), "launches_per_step": round(v["launches_per_step"], 2),
                            "ms_per_step": round(v["ms_per_step"], 5), "GBps": round(v["GBps"], 1),
                            "TFLOPps": round(v["TFLOPps"], 2), "bound": v["bound"], "frac": round(v["frac"], 4)}
                        for n, v in kt.items()},
        }
        if args.mode == "train" and world == 1 and not args.no_eval:
            # the eval half of BASELINE's metric (the renderer forward on the same batches), so that the driver's default run
            # records it too: `python bench.py --mode eval` reports the same measurement as a line of its own
            for i in range(5):
                eval_step(i)
            torch.cuda.synchronize()
            te = time.perf_counter()
            n_ev = max(args.steps, 20)
            for i in range(n_ev):
                eval_step(i % n_steps)
            torch.cuda.synchronize()
            t_ev = (time.perf_counter() - te) / n_ev
            line["eval"] = {"metric": f"rays/sec (eval), Lego 800^2 @ {args.grid}^3 grid", "value": B / t_ev, "unit": "rays/s",
                            "ms_per_step": t_ev * 1e3, "steps": n_ev,
                            "note": "renderer forward (OctreeRender_trilinear_fast, is_train=False) on the same 4096-ray batches, eager launches"}
        if not args.no_baselines and world == 1:
            cores = os.cpu_count()
            # thread-count sweep: one thread per core oversubscribes a 100+-core host on this memory-bound eager path
            # (round 1: 121 rays/s with 256 threads against 1.4 k rays/s on 8) -> time the fastest of a few counts
            cand = sorted({t for t in (8, 16, 32, 64, cores) if t <= cores})
            torch.set_num_threads(cand[0])
            steps_cpu, warm_cpu = 5, 2                                   # 2 warm-ups, median of >= 5
            need_cpu = B * (steps_cpu + warm_cpu + len(cand))
            idx = perm.reshape(-1)[:need_cpu]
            v, bs, ns, sw = oracle_baseline(model, rays[idx].cpu(), targets[idx].cpu(), n_samples, "cpu", args.mode,
                                            steps_cpu, warm_cpu, sweep_threads=cand, ndc=ndc, white=white)
            best = max(sw, key=sw.get)
            line["cpu_baseline"] = {"value": v, "unit": "rays/s", "cores": best, "kind": "port", "host_cores": cores,
                                    "cpu_model": cpu_model_name(),
                                    "threads_sweep_rays_per_s": {str(k): round(x, 1) for k, x in sw.items()},
                                    "sample": f"median of {ns} {args.mode} steps of {bs} rays (same scene, same N) after "
                                              f"{warm_cpu} warm-up steps, oracle/ref_torch.py on CPU with {best} torch threads "
                                              f"(fastest of {cand}, one step each); host has {cores} cores"}
            if args.config == "C2":
                # BASELINE config 1 IS "128^3, 4096-ray batch, CPU PyTorch reference": the same leg on that scene
                from recon_amd import synthetic as S
                m1, r1, n1, _, _ = S.baseline_scene("C1_vm128", dev)
                r1 = r1[S.bbox_hit_mask(r1, m1.aabb.cpu())]
                i1 = torch.randperm(r1.shape[0], generator=torch.Generator().manual_seed(3))[:B * (steps_cpu + warm_cpu)]
                t1 = torch.rand(i1.numel(), 3, generator=torch.Generator().manual_seed(4))
                torch.set_num_threads(best)
                v1, bs1, ns1 = oracle_baseline(m1, r1[i1], t1, n1, "cpu", args.mode, steps_cpu, warm_cpu)
                line["cpu_baseline"]["config1"] = {
                    "value": v1, "unit": "rays/s", "cores": best,
                    "sample": f"BASELINE config 1 (TensorVMSplit 128^3, N={n1}, {bs1}-ray batches, trained-like state): "
                              f"median of {ns1} {args.mode} steps after {warm_cpu} warm-ups, oracle on CPU, {best} threads"}
                del m1
            steps_g, warm_g = 5, 2
            idx = perm.reshape(-1)[:B * (steps_g + warm_g)]
            v2, bs2, ns2 = oracle_baseline(model, rays[idx], targets[idx], n_samples, str(dev), args.mode, steps_g, warm_g,
                                           ndc=ndc, white=white)
            line["rocm_eager_baseline"] = {"value": v2, "unit": "rays/s",
                                           "sample": f"median of {ns2} {args.mode} steps of {bs2} rays, the same "
                                                     f"restatement run eagerly on this GPU (PyTorch-ROCm path)",
                                           "speedup": value / v2}
        if not args.no_baselines and not args.no_psnr and world == 1 and args.mode == "train" and args.config == "C2":
            # BASELINE metric, quality half ("rays/sec (train) + PSNR"): equal-iterations PSNR of the HIP path against the
            # eager PyTorch-ROCm oracle on a frozen synthetic teacher — same init, batches and jitter for both
            # (oracle/psnr_parity.py).  (a) 64^3 with one alpha-mask update and one 48^3 -> 64^3 up-sampling inside 400
            # iterations, deltas at four cut points; (b) the headline geometry: 300^3, N = 1039, 4096-ray batches, 300
            # iterations from a fresh field (loss.py:46-47 is the PSNR formula)
            from oracle import psnr_parity
            del graphed
            torch.cuda.empty_cache()
            line["psnr"] = psnr_parity.run(recon_amd, dev=str(dev), grid=64, iters=400, schedule=True, init_grid=48,
                                           cuts=(100, 200, 300, 350))
            line["psnr"]["headline_geometry"] = psnr_parity.run(recon_amd, dev=str(dev), grid=300, iters=300, views=12,
                                                                res=128, cuts=(100, 200), teacher_mask_res=128)
        guard.__exit__()
        guard = None
        print(json.dumps(line))
        sys.stdout.flush()
    if guard is not None:
        guard.__exit__()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
