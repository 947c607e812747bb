"""GPU, world_size 2: the data-parallel train step as the driver's N > 1 bench runs it — real HIP gradients, the
cell-level support-limited exchange, the split hipGraph step.  Both ranks share cuda:0, so the collective goes over
gloo (RCCL refuses two ranks on one device); everything else is the code path of `bench.py --gpus N`."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _scene(recon, dev):
    from recon_amd import synthetic as S
    torch.manual_seed(0)
    aabb = torch.tensor(S.LEGO_AABB, device=dev)
    args = S.lego_args(density_n_comp=(8, 8, 8), app_n_comp=(16, 16, 16))
    model = recon.TensorVMSplit(args, aabb, [72, 64, 80], S.LEGO_NEAR_FAR, dev)
    S.make_trained_like(model, recon.AlphaGridMask, mask_res=40, radius=0.6)
    allrays = S.blender_rays(1)
    rays = allrays[torch.randperm(allrays.shape[0], generator=torch.Generator().manual_seed(2))[:4096]].to(dev).contiguous()
    target = torch.rand(rays.shape[0], 3, generator=torch.Generator().manual_seed(9)).to(dev)
    return model, rays, target


def _worker(rank, world, port, q):
    import faulthandler
    import sys
    faulthandler.enable()
    faulthandler.dump_traceback_later(250, exit=True)     # a rank that hangs reports where, instead of the peer timing out
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import recon_amd as recon
    from recon_amd import parallel
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = "cuda:0"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {"rank": rank}
    try:
        model, rays, target = _scene(recon, dev)
        ids = parallel.shard_ids(torch.arange(2048, device=dev), rank, world)
        N = 200
        # 1. eager: exchanged gradient == mean of the ranks' local gradients, over the WHOLE buffer (nothing outside
        #    the exchanged cells may have been non-zero anywhere)
        print(f"[rank {rank}] scene ready", flush=True)
        torch.manual_seed(10 + rank)
        rgb, _, _ = model(rays[ids], None, white_bg=True, is_train=True, N_samples=N)
        torch.mean((rgb - target[ids]) ** 2).backward()
        local = model.grad_flat.clone()
        rows = parallel.gradient_support_rows(model)
        out["cells"] = None if rows is None else rows[1].numel() * rows[0] / local.numel()
        parallel.allreduce_gradients(model)
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local)
        expect = sum(parts) / world
        out["exchange_err"] = float((model.grad_flat - expect).abs().max())
        out["grad_max"] = float(expect.abs().max())
        out["shaded"] = int(model.last["ws"].counters2d[:, 0].sum())
        del rgb      # with it goes the autograd graph of the default stream (see GraphedTrainStep's docstring)
        print(f"[rank {rank}] eager exchange done", flush=True)
        # 2. the split hipGraph step: same parameters on every rank after several steps, loss goes down
        model.zero_grad(set_to_none=True)
        opt = recon.FusedAdam(model.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
        gs = recon.GraphedTrainStep(model, opt, ids.numel(), N, warmup=1, regularizers=True)
        gs.set_regularizer_weights(0.01, 8e-5, 0.01, 0.01)      # rank-invariant terms, added after the exchange
        out["split"] = gs.split
        losses = []
        for it in range(8):
            losses.append(float(gs.step(rays, target, ids)))
            print(f"[rank {rank}] graphed step {it} loss {losses[-1]:.5f}", flush=True)
        torch.cuda.synchronize()
        out["graphs"] = gs.graph is not None and gs.graph_opt is not None
        out["losses"] = losses
        digest = torch.stack([p.detach().double().sum() for p in model.parameters()])
        both = [torch.empty_like(digest) for _ in range(world)]
        dist.all_gather(both, digest)
        out["same_params"] = bool(torch.equal(both[0], both[1]))
        out["finite"] = all(bool(torch.isfinite(p).all()) for p in model.parameters())
        # 2b. no white background (LLFF, llff.py:141 — BASELINE config 4 at 2 GPUs): the random-background draw of
        #     tensorBase.py:380 picks one of two captured variants per step, the same one on every rank
        opt_b = recon.FusedAdam(model.get_optparam_groups(0.005, 1e-3), betas=(0.9, 0.99))
        gs_b = recon.GraphedTrainStep(model, opt_b, ids.numel(), N, warmup=1, white_bg=False)
        seen = []
        for it in range(12):
            float(gs_b.step(rays, target, ids))
            seen.append(gs_b._bg)
        torch.cuda.synchronize()
        flags = torch.tensor([float(b) for b in seen])
        both_f = [torch.empty_like(flags) for _ in range(world)]
        dist.all_gather(both_f, flags)
        out["bg_same_draws"] = bool(torch.equal(both_f[0], both_f[1])) and 0 < sum(seen) < len(seen)
        out["bg_variants"] = len(gs_b._graphs)
        digest = torch.stack([p.detach().double().sum() for p in model.parameters()])
        dist.all_gather(both, digest)
        out["bg_same_params"] = bool(torch.equal(both[0], both[1]))
        del gs_b, opt_b
        # 2c. direct scatter (binned_scatter off): the density line gradients leave their replicas only at the end of the
        #     backward, so the split step must send ONE bucket after it (round-2 advisor finding)
        model.binned_scatter = False
        model._train_ws, model._ws_cache = {}, {}
        opt_c = recon.FusedAdam(model.get_optparam_groups(0.005, 1e-3), betas=(0.9, 0.99))
        gs_c = recon.GraphedTrainStep(model, opt_c, ids.numel(), N, warmup=1)
        for it in range(5):
            float(gs_c.step(rays, target, ids))
        torch.cuda.synchronize()
        out["direct_one_bucket"] = bool(gs_c._one_bucket) and gs_c.graph is not None
        # the exchanged density-line gradient of the last step == mean of the ranks' local ones (eager, same mode)
        model.zero_grad(set_to_none=True)
        torch.manual_seed(30 + rank)
        rgb, _, _ = model(rays[ids], None, white_bg=True, is_train=True, N_samples=N)
        torch.mean((rgb - target[ids]) ** 2).backward()
        local = model.grad_flat.clone()
        parallel.allreduce_gradients(model)
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local)
        out["direct_exchange_err"] = float((model.grad_flat - sum(parts) / world).abs().max())
        del rgb
        digest = torch.stack([p.detach().double().sum() for p in model.parameters()])
        dist.all_gather(both, digest)
        out["direct_same_params"] = bool(torch.equal(both[0], both[1]))
        model.binned_scatter = True
        model._train_ws, model._ws_cache = {}, {}
        # 3. sharded evaluation: every rank ends up with the image a single process renders
        def render(block):
            with torch.no_grad():
                r = recon.OctreeRender_trilinear_fast(block, model, chunk=1024, N_samples=N, white_bg=True, device=dev)
            return r[0], r[2]
        rgb_all, dep_all = parallel.render_sharded(render, rays[:3001])
        rgb_one, dep_one = render(rays[:3001])
        out["eval_same"] = bool(torch.equal(rgb_all, rgb_one) and torch.equal(dep_all, dep_one))
    except Exception as e:      # report instead of hanging the peer
        import traceback
        out["error"] = traceback.format_exc()
    q.put(out)
    dist.destroy_process_group()


def test_two_rank_train_step_on_one_gpu():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=300) for _ in range(world)]
    except Exception:
        for p in procs:
            p.join(timeout=5)
        raise AssertionError(f"no result from the ranks; exit codes {[p.exitcode for p in procs]}")
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert "error" not in r, r["error"]
        assert r["cells"] is not None and r["cells"] < 0.6 and r["shaded"] > 2000 and r["grad_max"] > 0, r
        assert r["exchange_err"] <= 1e-6 * r["grad_max"], r
        assert r["split"] and r["graphs"] and r["same_params"] and r["finite"] and r["eval_same"], r
        assert min(r["losses"][3:]) < r["losses"][0], r["losses"]
        assert r["bg_same_draws"] and r["bg_variants"] == 2 and r["bg_same_params"], r
        assert r["direct_one_bucket"] and r["direct_same_params"] and r["direct_exchange_err"] <= 1e-6 * r["grad_max"], r
    print("cells exchanged: %.1f %% of the gradient buffer" % (100 * res[0]["cells"]))


def test_split_step_over_rccl_in_a_one_rank_group():
    """The three-graph data-parallel step over backend "nccl" (RCCL) — a one-rank group in a fresh child process
    (tests/rccl_rehearsal.py; TF_DP_FORCE_EXCHANGE=1): 8 steps with both collectives between the replays must follow
    the single-graph step's trajectory (the bars of tests/test_hip_backward.py::_same_trajectory), scatter status clean."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "rccl_rehearsal.py"), str(_free_port())],
                       capture_output=True, text=True, timeout=400, cwd=root, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (p.stdout[-1000:], p.stderr[-2000:])
    r = json.loads(lines[0])
    print(r)
    assert r["backend"] == "nccl" and r["world"] == 1
    assert r["single_graph"] and r["split_graphs"] and r["finite"]
    assert r["buckets"][0] >= 1 and r["buckets"][1] >= 1 and min(r["bucket_floats"]) > 0      # both buckets really travelled
    assert r["worst_loss_gap"] <= 2e-3 and r["worst_param_gap_mlp"] <= 0.35 and r["worst_param_gap_other"] <= 0.05, r
