#!/usr/bin/env python3
"""PSNR after equal iterations: HIP path vs the eager PyTorch-ROCm restatement (oracle) on a frozen synthetic
teacher field (SURVEY §8d 'Quality'), plus one run of the full intended schedule (alpha-mask updates, shrink,
coarse-to-fine up-sampling) on the HIP path.  Writes a JSON report (default profiles/r01_psnr_parity.json).

Part A keeps everything but the arithmetic identical: same initial parameters, same batch permutation, same
CPU-generator jitter stream, same Adam groups / learning-rate decay, fixed grid."""
import argparse, json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import recon_amd
from recon_amd import synthetic as S, harness
from oracle import ref_torch as R

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=64)
ap.add_argument("--iters", type=int, default=400)
ap.add_argument("--views", type=int, default=20)
ap.add_argument("--res", type=int, default=100)
ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r01_psnr_parity.json"))
args = ap.parse_args()
dev = "cuda:0"
B = 4096


def make_model(grid, seed):
    torch.manual_seed(seed)
    aabb = torch.tensor(S.LEGO_AABB, device=dev)
    return recon_amd.TensorVMSplit(S.lego_args(), aabb, [grid] * 3, S.LEGO_NEAR_FAR, dev)


# ---- teacher + data
teacher = make_model(args.grid, 123)
S.make_trained_like(teacher, recon_amd.AlphaGridMask, mask_res=64)
with torch.no_grad():   # give the teacher a position-dependent colour: boost a few appearance components
    teacher.app_plane[0][:, :6] *= 12.0
    teacher.app_plane[1][:, 6:12] *= 12.0
    teacher.basis_mat.weight.mul_(3.0)
rays_all = S.blender_rays(args.views + 1, H=args.res, W=args.res, seed=7)
n_test = args.res * args.res
rays_test, rays_train = rays_all[:n_test].to(dev), rays_all[n_test:].to(dev)
with torch.no_grad():
    gt_train = recon_amd.OctreeRender_trilinear_fast(rays_train, teacher, chunk=B, white_bg=True, device=dev)[0]
    gt_test = recon_amd.OctreeRender_trilinear_fast(rays_test, teacher, chunk=B, white_bg=True, device=dev)[0]
keep = S.bbox_hit_mask(rays_train.cpu(), torch.tensor(S.LEGO_AABB)).to(dev)
rays_train, gt_train = rays_train[keep], gt_train[keep]
print(f"teacher rendered: {rays_train.shape[0]} train rays, {n_test} test rays; test image mean {gt_test.mean().item():.3f}", flush=True)

n_iters = args.iters
lr_factor = 0.1 ** (1 / n_iters)
perm_rng = np.random.default_rng(11)
batches = []
cur, ids = rays_train.shape[0], None
for it in range(n_iters):   # SimpleSampler order, shared by both runs
    cur += B
    if cur + B > rays_train.shape[0]:
        ids = torch.from_numpy(perm_rng.permutation(rays_train.shape[0])).to(dev)
        cur = 0
    batches.append(ids[cur:cur + B])
report = {"grid": args.grid, "iters": n_iters, "batch": B, "train_rays": int(rays_train.shape[0])}

# ---- A1: HIP student
student = make_model(args.grid, 5)
init_state = {k: v.detach().clone() for k, v in student.state_dict().items()}
N = student.nSamples
opt = torch.optim.Adam(student.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
student.lazy_sample_count = True
torch.manual_seed(99)
t0 = time.perf_counter()
for it in range(n_iters):
    rgb = recon_amd.OctreeRender_trilinear_fast(rays_train[batches[it]], student, None, chunk=B, N_samples=N, white_bg=True,
                                               device=dev, is_train=True)[0]
    loss = torch.mean((rgb - gt_train[batches[it]]) ** 2)
    opt.zero_grad(); loss.backward(); opt.step()
    for g in opt.param_groups:
        g["lr"] *= lr_factor
torch.cuda.synchronize()
t_hip = time.perf_counter() - t0
psnr_hip = harness.evaluate_psnr(student, rays_test, gt_test, n_samples=N, device=dev)
print(f"HIP   : {n_iters} iters in {t_hip:.1f} s, final train loss {loss.item():.5f}, test PSNR {psnr_hip:.3f} dB", flush=True)

# ---- A2: eager oracle student (same init, same batches, same jitter stream)
cfg = R.FieldCfg(model="TensorVMSplit", aabb=torch.tensor(S.LEGO_AABB, device=dev), gridSize=[args.grid] * 3,
                 near_far=S.LEGO_NEAR_FAR, **{k: v for k, v in S.lego_args().items() if k not in ("alphaMask_thres",)}).finalize()
params = {k: v.detach().contiguous().clone().requires_grad_(True) for k, v in init_state.items()}
fast = [v for k, v in params.items() if "_plane." in k or "_line." in k]
slow = [v for k, v in params.items() if not ("_plane." in k or "_line." in k)]
opt_o = torch.optim.Adam([{"params": fast, "lr": 0.02}, {"params": slow, "lr": 1e-3}], betas=(0.9, 0.99))
torch.manual_seed(99)
t0 = time.perf_counter()
for it in range(n_iters):
    rgb, _, _ = R.render_rays(cfg, params, rays_train[batches[it]], None, white_bg=True, is_train=True, n_samples=N)
    loss_o = torch.mean((rgb - gt_train[batches[it]]) ** 2)
    opt_o.zero_grad(); loss_o.backward(); opt_o.step()
    for g in opt_o.param_groups:
        g["lr"] *= lr_factor
torch.cuda.synchronize()
t_eager = time.perf_counter() - t0
with torch.no_grad():
    out = R.render_chunked(cfg, params, rays_test, None, chunk=B, n_samples=N, white_bg=True, device=dev)[0]
psnr_eager = harness.psnr(torch.mean((out.clamp(0, 1) - gt_test) ** 2))
print(f"eager : {n_iters} iters in {t_eager:.1f} s, final train loss {loss_o.item():.5f}, test PSNR {psnr_eager:.3f} dB", flush=True)
report["equal_iterations"] = {"psnr_hip_db": psnr_hip, "psnr_eager_db": psnr_eager, "delta_db": psnr_hip - psnr_eager,
                              "train_seconds_hip": t_hip, "train_seconds_eager": t_eager,
                              "final_train_loss_hip": loss.item(), "final_train_loss_eager": loss_o.item()}

# ---- B: full intended schedule on the HIP path
student2 = make_model(32, 6)
cfgB = dict(n_iters=600, batch_size=B, N_voxel_init=32 ** 3, N_voxel_final=args.grid ** 3, upsamp_list=[200, 300, 400],
            update_AlphaMask_list=[150, 350], TV_weight_density=0.01, TV_weight_app=0.01, L1_weight_inital=8e-5,
            L1_weight_rest=4e-5, Ortho_weight=0.01)
t0 = time.perf_counter()
hist = harness.train(student2, rays_train, gt_train, cfgB, device=dev, log_every=100, seed=3)
torch.cuda.synchronize()
tB = time.perf_counter() - t0
psnrB = harness.evaluate_psnr(student2, rays_test, gt_test, device=dev)
print(f"sched : 600 iters with alpha-mask updates / shrink / up-sampling in {tB:.1f} s, events {hist['events']}, "
      f"train PSNR log {[(i, round(p, 2)) for i, p in hist['psnr']]}, test PSNR {psnrB:.3f} dB", flush=True)
report["full_schedule_hip"] = {"cfg": cfgB, "events": hist["events"], "train_psnr": hist["psnr"], "test_psnr_db": psnrB,
                               "seconds": tB, "final_grid": student2.gridSize.tolist()}
os.makedirs(os.path.dirname(args.out), exist_ok=True)
json.dump(report, open(args.out, "w"), indent=1)
print("wrote", args.out)
