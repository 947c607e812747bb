"""PSNR after equal iterations: the HIP path against the eager PyTorch-ROCm oracle on a frozen synthetic teacher
field (SURVEY §8d "Quality"; the reference computes PSNR at loss.py:46-47 inside train.py:411-445).  TEST INFRASTRUCTURE
(uses the oracle): imported by tests/test_psnr_parity.py and by bench.py's baseline legs.

Everything but the arithmetic is identical for the two students: initial parameters, SimpleSampler batch order, the
CPU-generator jitter stream, Adam groups and learning-rate decay, and — with `schedule=True` — one alpha-mask update
(tensorBase.py:233-256, no shrink) and one grid up-sampling (tensoRF.py:283-288) with the optimizer rebuilt
(train.py:450-481), each student rebuilding its own mask / resizing its own tensors."""
import math
import time

import numpy as np
import torch


def psnr_db(mse):
    return -10.0 * math.log(max(float(mse), 1e-12)) / math.log(10.0)      # loss.py:46-47


def run(recon, dev="cuda:0", grid=64, iters=400, views=20, res=100, batch=4096, schedule=False, seed=5,
        init_grid=None, mask_at=None, upsample_at=None):
    from recon_amd import synthetic as S
    from oracle import ref_torch as R
    aabb = torch.tensor(S.LEGO_AABB, device=dev)

    def make_model(g, sd):
        torch.manual_seed(sd)
        return recon.TensorVMSplit(S.lego_args(), aabb, [g] * 3, S.LEGO_NEAR_FAR, dev)

    # ---- teacher + data (rendered by the HIP path: both students see the same targets)
    teacher = make_model(grid, 123)
    S.make_trained_like(teacher, recon.AlphaGridMask, mask_res=64)
    with torch.no_grad():   # a position-dependent colour
        teacher.app_plane[0][:, :6] *= 12.0
        teacher.app_plane[1][:, 6:12] *= 12.0
        teacher.basis_mat.weight.mul_(3.0)
    rays_all = S.blender_rays(views + 1, H=res, W=res, seed=7)
    n_test = res * res
    rays_test, rays_train = rays_all[:n_test].to(dev), rays_all[n_test:].to(dev)
    with torch.no_grad():
        gt_train = recon.OctreeRender_trilinear_fast(rays_train, teacher, chunk=batch, white_bg=True, device=dev)[0]
        gt_test = recon.OctreeRender_trilinear_fast(rays_test, teacher, chunk=batch, white_bg=True, device=dev)[0]
    keep = S.bbox_hit_mask(rays_train.cpu(), torch.tensor(S.LEGO_AABB)).to(dev)
    rays_train, gt_train = rays_train[keep], gt_train[keep]
    B = min(batch, rays_train.shape[0])
    lr_factor = 0.1 ** (1 / iters)
    rng = np.random.default_rng(11)
    batches, cur, ids = [], rays_train.shape[0], None
    for _ in range(iters):   # SimpleSampler order (train.py:44-56), shared by both runs
        cur += B
        if cur + B > rays_train.shape[0]:
            ids = torch.from_numpy(rng.permutation(rays_train.shape[0])).to(dev)
            cur = 0
        batches.append(ids[cur:cur + B])
    g0 = init_grid if (schedule and init_grid) else grid
    mask_at = (iters * 3 // 8) if mask_at is None else mask_at
    upsample_at = (iters * 5 // 8) if upsample_at is None else upsample_at
    mask_reso = (64, 64, 64)

    # ---- HIP student
    student = make_model(g0, seed)
    init_state = {k: v.detach().clone() for k, v in student.state_dict().items()}
    N = min(int(1e6), recon.cal_n_samples([g0] * 3, 0.5))
    opt = recon.FusedAdam(student.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
    student.lazy_sample_count = True
    torch.manual_seed(99)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(iters):
        rgb = recon.OctreeRender_trilinear_fast(rays_train[batches[it]], student, None, chunk=B, N_samples=N, white_bg=True,
                                                device=dev, is_train=True)[0]
        loss = torch.mean((rgb - gt_train[batches[it]]) ** 2)
        opt.zero_grad()
        loss.backward()
        opt.step()
        for g in opt.param_groups:
            g["lr"] *= lr_factor
        if schedule and it == mask_at:
            student.updateAlphaMask(mask_reso)
        if schedule and it == upsample_at:
            student.upsample_volume_grid([grid] * 3)
            N = min(N, recon.cal_n_samples([grid] * 3, 0.5))           # train.py:472
            opt = recon.FusedAdam(student.get_optparam_groups(0.02 * lr_factor ** (it + 1), 1e-3 * lr_factor ** (it + 1)),
                                  betas=(0.9, 0.99))
    torch.cuda.synchronize()
    t_hip = time.perf_counter() - t0
    with torch.no_grad():
        out = recon.OctreeRender_trilinear_fast(rays_test, student, chunk=batch, N_samples=N, white_bg=True, device=dev)[0]
    psnr_hip = psnr_db(torch.mean((out.clamp(0, 1) - gt_test) ** 2))
    student.check_scatter_status()

    # ---- eager oracle student (same init, same batches, same jitter stream)
    cfg = R.FieldCfg(model="TensorVMSplit", aabb=aabb.clone(), gridSize=[g0] * 3, near_far=S.LEGO_NEAR_FAR,
                     **{k: v for k, v in S.lego_args().items() if k not in ("alphaMask_thres",)}).finalize()
    thres = S.lego_args()["alphaMask_thres"]
    params = {k: v.detach().contiguous().clone().requires_grad_(True) for k, v in init_state.items()}

    def make_opt(lr_xyz, lr_net):
        fast = [v for k, v in params.items() if "_plane." in k or "_line." in k]
        slow = [v for k, v in params.items() if not ("_plane." in k or "_line." in k)]
        return torch.optim.Adam([{"params": fast, "lr": lr_xyz}, {"params": slow, "lr": lr_net}], betas=(0.9, 0.99))

    opt_o = make_opt(0.02, 1e-3)
    N = min(int(1e6), R.cal_n_samples([g0] * 3, 0.5))
    torch.manual_seed(99)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(iters):
        rgb, _, _ = R.render_rays(cfg, params, rays_train[batches[it]], None, white_bg=True, is_train=True, n_samples=N)
        loss_o = torch.mean((rgb - gt_train[batches[it]]) ** 2)
        opt_o.zero_grad()
        loss_o.backward()
        opt_o.step()
        for g in opt_o.param_groups:
            g["lr"] *= lr_factor
        if schedule and it == mask_at:
            R.update_alpha_mask(cfg, params, mask_reso, thres)
        if schedule and it == upsample_at:
            params = R.upsample_params(cfg, params, [grid] * 3)
            N = min(N, R.cal_n_samples([grid] * 3, 0.5))
            opt_o = make_opt(0.02 * lr_factor ** (it + 1), 1e-3 * lr_factor ** (it + 1))
    torch.cuda.synchronize()
    t_eager = time.perf_counter() - t0
    with torch.no_grad():
        out = R.render_chunked(cfg, params, rays_test, None, chunk=batch, n_samples=N, white_bg=True, device=dev)[0]
    psnr_eager = psnr_db(torch.mean((out.clamp(0, 1) - gt_test) ** 2))
    return {"grid": grid, "init_grid": g0, "iters": iters, "batch": B, "train_rays": int(rays_train.shape[0]),
            "schedule": ({"alpha_mask_update_at": mask_at, "upsample_at": upsample_at, "mask_reso": list(mask_reso)}
                         if schedule else None),
            "psnr_hip_db": psnr_hip, "psnr_eager_db": psnr_eager, "delta_db": psnr_hip - psnr_eager,
            "train_seconds_hip": t_hip, "train_seconds_eager": t_eager,
            "final_train_loss_hip": float(loss.detach()), "final_train_loss_eager": float(loss_o.detach()),
            "data": "synthetic teacher only (no dataset exists offline): targets rendered from a frozen seeded field"}
