"""Soak run of harness.train on the reference's default schedule shape (3000 iterations, 128^3 -> 300^3 in five
up-samplings, two alpha-mask updates with shrink, all four regularisers, FreeNeRF masks on) — eager and captured —
on a synthetic teacher: wall time, end PSNR, workspace footprint, overflow re-runs, scatter status."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import recon_amd as recon
from recon_amd import synthetic as S, harness

dev = "cuda:0"
aabb = torch.tensor(S.LEGO_AABB, device=dev)
args = S.lego_args()
torch.manual_seed(0)
teacher = recon.TensorVMSplit(args, aabb, [128] * 3, S.LEGO_NEAR_FAR, dev)
S.make_trained_like(teacher, recon.AlphaGridMask, mask_res=128)
with torch.no_grad():
    teacher.app_plane[0][:, :6] *= 12.0
    teacher.basis_mat.weight.mul_(3.0)
rays = S.blender_rays(30, H=200, W=200, seed=3).to(dev)
with torch.no_grad():
    gt = recon.OctreeRender_trilinear_fast(rays, teacher, chunk=8192, white_bg=True, device=dev)[0]
free = bool(int(os.environ.get("FREE_REG", "1")))
cfg = dict(n_iters=3000, batch_size=4096, N_voxel_init=128 ** 3, N_voxel_final=300 ** 3,
           upsamp_list=[600, 900, 1200, 1650, 2100], update_AlphaMask_list=[600, 1200], TV_weight_density=0.1,
           TV_weight_app=0.01, L1_weight_inital=8e-5, L1_weight_rest=4e-5, Ortho_weight=0.0, free_reg=free,
           n_samples_rule="upstream")
_psnr = harness.psnr
for graphed in (True, False):
    marks = []
    harness.psnr = lambda mse: (marks.append(time.perf_counter()), _psnr(mse))[1]      # wall clock at every log point
    torch.manual_seed(11)
    student = recon.TensorVMSplit(args, aabb, recon.N_to_reso(128 ** 3, aabb), S.LEGO_NEAR_FAR, dev)
    torch.manual_seed(12)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hist = harness.train(student, rays, gt, cfg, device=dev, log_every=500, seed=1, graphed=graphed)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    p = harness.evaluate_psnr(student, rays[:40000], gt[:40000], n_samples=hist["n_samples"][-1], device=dev)
    print(f"graphed={graphed} free_reg={free}: {cfg['n_iters']} iterations in {dt:.2f} s = {dt / cfg['n_iters'] * 1e3:.3f} ms/iteration, "
          f"PSNR {p:.2f} dB, final grid {student.gridSize.tolist()}, N {hist['n_samples'][-1]}, events {[e[:2] for e in hist['events']]}, "
          f"train PSNR log {[round(x[1], 2) for x in hist['psnr']]}, workspace {student.workspace_bytes() / 2 ** 30:.2f} GiB, "
          f"peak allocated {torch.cuda.max_memory_allocated() / 2 ** 30:.2f} GiB, ms/iteration per 500-iteration window "
          f"{[round((b - a) / 500 * 1e3, 3) for a, b in zip(marks[:-1], marks[1:])]}", flush=True)
    if graphed:      # where does a step of the trained field spend its time?  five eager steps with per-launch events
        opt = recon.FusedAdam(student.get_optparam_groups(0.002, 1e-4), betas=(0.9, 0.99))
        mk = recon.get_free_mask(pos_bl=student.pos_bit_length, view_bl=student.view_bit_length, fea_bl=student.fea_bit_length,
                                 den_bl=student.density_n_comp, app_bl=student.app_n_comp, step=2999, total_step=3000, device=dev) if free else None
        N = hist["n_samples"][-1]
        ids = torch.randperm(rays.shape[0], device=dev)[:4096 * 8].view(8, 4096)
        for i in range(8):
            if i == 3:
                student.kernel_events = {}
            rgb, _, _ = student(rays[ids[i]], mk, white_bg=True, is_train=True, N_samples=N)
            loss = torch.mean((rgb - gt[ids[i]]) ** 2)
            opt.zero_grad(); loss.backward(); opt.step()
        torch.cuda.synchronize()
        ev, student.kernel_events = student.kernel_events, None
        c = student.last["ws"].counters2d[:, :3].sum(0).tolist()
        print("   trained field, one 4096-ray batch: per ray %.1f in-bbox / %.1f density / %.1f shaded; " % (c[2] / 4096, c[1] / 4096, c[0] / 4096)
              + "  ".join("%s %.0fus" % (k[3:], sum(a.elapsed_time(b) for a, b in v) / len(v) * 1e3) for k, v in ev.items()), flush=True)
    if graphed:      # the captured step on the trained field, outside the harness loop: GPU time of the step itself
        del rgb, loss, opt       # (results of eager training forwards must not be alive when a step is captured: graph.py)
        student.zero_grad(set_to_none=True)
        for regs in (False, True):
            opt = recon.FusedAdam(student.get_optparam_groups(0.002, 1e-4), betas=(0.9, 0.99))
            gs = recon.GraphedTrainStep(student, opt, 4096, hist["n_samples"][-1], warmup=1, regularizers=regs)
            if regs:
                gs.set_regularizer_weights(0.0, 4e-5, 0.01, 0.001)
            perm = torch.randperm(rays.shape[0], device=dev)
            for i in range(10):
                gs.step(rays, gt, perm[i * 4096:(i + 1) * 4096])
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for i in range(10, 110):
                gs.step(rays, gt, perm[i * 4096:(i + 1) * 4096])
            host = time.perf_counter() - t1
            torch.cuda.synchronize(); dt1 = time.perf_counter() - t1
            print(f"   captured step on the trained field, regularizers={regs}: {dt1 / 100 * 1e3:.3f} ms per step (host issue {host / 100 * 1e3:.3f} ms), "
                  f"overflow re-runs {getattr(gs, 'overflow_reruns', 0)}", flush=True)
            del gs, opt
    if graphed:      # ... and the harness loop itself on the trained field, no schedule events: its own overhead per iteration
        cfg2 = dict(cfg, n_iters=400, upsamp_list=[], update_AlphaMask_list=[], N_voxel_init=300 ** 3, lr_init=0.002, lr_basis=1e-4)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        harness.train(student, rays, gt, cfg2, device=dev, log_every=0, seed=2, graphed=True)
        torch.cuda.synchronize()
        print(f"   harness.train(graphed=True) on the trained field, 400 iterations without events: {(time.perf_counter() - t1) / 400 * 1e3:.3f} ms per iteration", flush=True)
    del student
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
