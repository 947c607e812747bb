"""GPU edge cases of the hot path (empty and ragged batches, rays that miss, the per-ray sample limit, a chunk that
is not a multiple of the 64 packed-list shards, training on a batch without shaded samples) against the oracle."""
import numpy as np
import pytest
import torch

from oracle import ref_torch as R
from tests._golden import Case
from tests.helpers import build_model, oracle_of
from tests.test_hip_forward import ATOL_RGB, RTOL

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(recon, name="vm_cubic_eval"):
    c = Case(name)
    return c, build_model(recon, c, DEV)


@pytest.mark.parametrize("n_rays", [0, 1, 63, 65, 4097])
def test_ragged_and_empty_batches(recon, n_rays):
    c, model = _model(recon)
    rays = c.rays.to(DEV)
    reps = (n_rays + rays.shape[0] - 1) // max(rays.shape[0], 1) + 1
    rays = rays.repeat(reps, 1)[:n_rays].contiguous()
    with torch.no_grad():
        rgb, depth, nvalid = model(rays, None, white_bg=True, is_train=False)
    assert rgb.shape == (n_rays, 3) and depth.shape == (n_rays,)
    if n_rays == 0:
        assert int(nvalid) == 0
        out = recon.OctreeRender_trilinear_fast(rays, model, None, chunk=4096, device=DEV)
        assert out[0].shape[0] == 0 and float(out[5]) == 0.0
        return
    cfg, params = oracle_of(model, DEV)
    with torch.no_grad():
        o_rgb, o_depth, o_n = R.render_rays(cfg, params, rays, None, white_bg=True, is_train=False)
    assert int(nvalid) == int(o_n)
    np.testing.assert_allclose(rgb.cpu().numpy(), o_rgb.cpu().numpy(), rtol=RTOL, atol=ATOL_RGB)
    np.testing.assert_allclose(depth.cpu().numpy(), o_depth.cpu().numpy(), rtol=RTOL, atol=1e-5)


def test_rays_that_miss_the_box_render_background(recon):
    c, model = _model(recon)
    o = torch.tensor([[5.0, 5.0, 5.0]], device=DEV).repeat(130, 1)
    d = torch.nn.functional.normalize(torch.tensor([[1.0, 0.3, 0.2]], device=DEV), dim=-1).repeat(130, 1)
    rays = torch.cat([o, d], 1)
    with torch.no_grad():
        rgb, depth, nvalid = model(rays, None, white_bg=True, is_train=False)
        cfg, params = oracle_of(model, DEV)
        o_rgb, o_depth, o_n = R.render_rays(cfg, params, rays, None, white_bg=True, is_train=False)
    assert int(nvalid) == 0 == int(o_n)
    assert torch.equal(rgb, torch.ones_like(rgb)) and torch.equal(o_rgb, torch.ones_like(o_rgb))
    np.testing.assert_allclose(depth.cpu().numpy(), o_depth.cpu().numpy(), rtol=RTOL, atol=1e-6)


def test_sample_count_limit(recon):
    """N_samples up to TF_MAX_SAMPLES (8192) per ray is supported and agrees with the oracle; above it the call fails
    loudly instead of truncating."""
    c, model = _model(recon)
    rays = c.rays.to(DEV)[:96].contiguous()
    with torch.no_grad():
        rgb, depth, nvalid = model(rays, None, white_bg=True, is_train=False, N_samples=8192)
        cfg, params = oracle_of(model, DEV)
        o_rgb, o_depth, o_n = R.render_rays(cfg, params, rays, None, white_bg=True, is_train=False, n_samples=8192)
    assert abs(int(nvalid) - int(o_n)) <= 1
    np.testing.assert_allclose(rgb.cpu().numpy(), o_rgb.cpu().numpy(), rtol=RTOL, atol=ATOL_RGB)
    with pytest.raises(recon._hip.HipError):
        model(rays, None, white_bg=True, is_train=False, N_samples=8193)


@pytest.mark.parametrize("none_grads", [True, False])
def test_training_step_without_shaded_samples(recon, none_grads):
    """A batch whose rays all miss: forward, backward and the optimizer run and nothing changes.  With
    `reference_none_grads` (default) no parameter gets a gradient — the reference's graph holds none of them without a
    valid sample (tensorBase.py:359, :370; its own backward() would even refuse the loss) — without it every gradient
    is a tensor of exact zeros and FusedAdam's device-side gates keep the step counts where they are."""
    c, model = _model(recon, "vm_cubic_train")
    model.reference_none_grads = none_grads
    o = torch.tensor([[5.0, 5.0, 5.0]], device=DEV).repeat(256, 1)
    d = torch.nn.functional.normalize(torch.tensor([[1.0, 0.3, 0.2]], device=DEV), dim=-1).repeat(256, 1)
    rays = torch.cat([o, d], 1)
    opt = recon.FusedAdam(model.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    rgb, _, n = model(rays, None, white_bg=True, is_train=True)
    loss = torch.mean((rgb - 0.5) ** 2)
    opt.zero_grad()
    loss.backward()
    for k, p in model.named_parameters():
        if none_grads:
            assert p.grad is None, k
        else:
            assert p.grad is not None and float(p.grad.abs().max()) == 0.0, k
    opt.step()
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k]), k
    if not none_grads:
        assert all(float(opt.state[p]["step"]) == 0.0 for p in model.parameters())


def test_generated_rays_match_the_loader_formulas(recon):
    """tf_generate_rays against the torch expressions of dataLoader/ray_utils.py restated here (more shapes and options
    than the reference-generated fixture of test_generated_rays_match_the_reference_fixture covers)."""
    from recon_amd import synthetic as S
    H_, W_, f = 37, 53, 61.5
    g = torch.Generator().manual_seed(2)
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
    c2w = torch.cat([q, torch.randn(3, 1, generator=g)], 1)
    j, i = torch.meshgrid(torch.arange(H_, dtype=torch.float32), torch.arange(W_, dtype=torch.float32), indexing="ij")
    # (1) OpenCV-style camera, unit directions (Blender loader), whole image and a pixel subset
    dirs = torch.stack([(i + 0.5 - W_ / 2) / f, (j + 0.5 - H_ / 2) / f, torch.ones_like(i)], -1)
    dirs = dirs / torch.norm(dirs, dim=-1, keepdim=True)
    ref = torch.cat([c2w[:, 3].expand(H_ * W_, 3), dirs.view(-1, 3) @ c2w[:, :3].T], 1)
    out = recon.generate_rays(H_, W_, f, c2w, device=DEV)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=2e-6, atol=2e-6)
    ids = torch.randperm(H_ * W_, generator=g)[:500]
    out = recon.generate_rays(H_, W_, f, c2w, pixel_ids=ids, device=DEV)
    np.testing.assert_allclose(out.cpu().numpy(), ref[ids].numpy(), rtol=2e-6, atol=2e-6)
    # (2) OpenGL-style camera, principal point off centre, no normalisation, NDC projection (forward-facing scenes)
    c2w2 = torch.cat([torch.eye(3) + 0.02 * torch.randn(3, 3, generator=g), 0.1 * torch.randn(3, 1, generator=g)], 1)
    cx, cy = 25.0, 19.5
    dirs = torch.stack([(i + 0.5 - cx) / f, -(j + 0.5 - cy) / (1.1 * f), -torch.ones_like(i)], -1).view(-1, 3)
    o, d = S.ndc_project(H_, W_, f, 1.0, c2w2[:, 3].expand(H_ * W_, 3), dirs @ c2w2[:, :3].T)
    out = recon.generate_rays(H_, W_, (f, 1.1 * f), c2w2, center=(cx, cy), normalize=False, opengl=True, ndc_near=1.0,
                              device=DEV)
    # the loader's projection takes ONE focal length; with fy = 1.1 fx the kernel's y scale (2 fy / H) is 1.1 x its
    ref2 = torch.cat([o, d], 1)
    got = out.cpu()
    np.testing.assert_allclose(got[:, [0, 2, 3, 5]].numpy(), ref2[:, [0, 2, 3, 5]].numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got[:, [1, 4]].numpy(), 1.1 * ref2[:, [1, 4]].numpy(), rtol=1e-5, atol=1e-5)


def test_generated_rays_match_the_reference_fixture(recon):
    """tf_generate_rays against rays the reference's own get_rays / ndc_rays_blender produced (dataLoader/ray_utils.py:
    66-107; tests/golden/aux_refs.npz: gen_golden.py imports the module with a stub for the kornia import, which those
    two functions never touch).  The camera-space pixel directions are the fixture's inputs."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "aux_refs.npz"))
    for tag in ("blender", "llff"):
        H_, W_, f = int(z[f"rays/{tag}/H"]), int(z[f"rays/{tag}/W"]), float(z[f"rays/{tag}/focal"])
        opengl = bool(int(z[f"rays/{tag}/opengl"]))
        c2w = torch.from_numpy(z[f"rays/{tag}/c2w"])
        out = recon.generate_rays(H_, W_, f, c2w, normalize=False, opengl=opengl, device=DEV).cpu().numpy()
        ref = np.concatenate([z[f"rays/{tag}/rays_o"], z[f"rays/{tag}/rays_d"]], 1)
        np.testing.assert_allclose(out, ref, rtol=2e-6, atol=2e-6)
        if opengl:      # llff.py:203: the NDC warp of the same rays, near plane 1
            out = recon.generate_rays(H_, W_, f, c2w, normalize=False, opengl=True, ndc_near=1.0, device=DEV).cpu().numpy()
            ref = np.concatenate([z[f"rays/{tag}/ndc_o"], z[f"rays/{tag}/ndc_d"]], 1)
            np.testing.assert_allclose(out, ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("scene", ["C2_vm300", "C4_ndc", "C5_tt640", "shrunk"])
def test_gradient_support_bounds_the_real_gradients(recon, scene):
    """parallel.gradient_support (what the data-parallel all-reduce exchanges) against the kernels: every entry of the
    gradient buffer outside the reported segments is exactly zero after a real training backward."""
    from recon_amd import parallel
    from tests.test_full_size import _scene
    if scene == "shrunk":       # mask aabb != model aabb, off-centre occupancy
        from recon_amd import synthetic as S
        torch.manual_seed(0)
        aabb = torch.tensor([[-1.2, -0.9, -1.4], [1.0, 1.3, 0.8]], device=DEV)
        model = recon.TensorVMSplit(S.lego_args(), aabb, [96, 80, 112], S.LEGO_NEAR_FAR, DEV)
        S.make_trained_like(model, recon.AlphaGridMask, mask_res=48, radius=0.5)
        vol = model.alphaMask.alpha_volume[0, 0].clone()
        vol[:, :, 30:] = 0                     # keep the low-x half of the ball only
        model.alphaMask = recon.AlphaGridMask(DEV, torch.tensor([[-1.5] * 3, [1.5] * 3], device=DEV), vol)
        rays, N, ndc = S.blender_rays(1)[::37][:4096].to(DEV).contiguous(), 300, False
    else:
        model, rays, N, ndc, _ = _scene(recon, scene)
    torch.manual_seed(1)
    rgb, _, _ = model(rays, None, white_bg=True, is_train=True, ndc_ray=ndc, N_samples=N)
    (rgb ** 2).mean().backward()
    segs = parallel.gradient_support(model)
    assert segs is not None, "the scene's alpha mask leaves most plane rows empty"
    flat = model.grad_flat
    inside = torch.zeros(flat.numel(), dtype=torch.bool, device=DEV)
    for a, b in segs:
        inside[a:b] = True
    frac = float(inside.float().mean())
    assert float(flat.abs().sum()) > 0 and frac < 0.9
    assert float(flat[~inside].abs().max()) == 0.0, "a gradient entry outside the exchanged support is non-zero"
    # the cell-level support (what allreduce_gradients exchanges by default): tighter, and still a superset
    rows = parallel.gradient_support_rows(model)
    assert rows is not None
    w, idx = rows
    in_rows = torch.zeros(flat.numel() // w, dtype=torch.bool, device=DEV)
    in_rows[idx] = True
    assert float(flat.view(-1, w)[~in_rows].abs().max()) == 0.0, "a gradient entry outside the exchanged cells is non-zero"
    cell_frac = idx.numel() * w / flat.numel()
    assert cell_frac < frac
    live = float((flat.view(-1, w)[in_rows].abs().amax(1) > 0).float().mean())
    print(scene, f"cells = {cell_frac:.2%} of the gradient buffer ({live:.0%} of them non-zero in this one step)")
    # tightness: the outermost exchanged rows of the largest plane are within a few rows of real data
    print(scene, f"support = {frac:.2%} of the gradient buffer, {len(segs)} segments")


def test_batch_gather_matches_indexing(recon):
    """tf_gather_batch (GraphedTrainStep's staging) == allrays[ray_idx], allrgbs[ray_idx] (train.py:297-298)."""
    import ctypes as C
    from recon_amd import _hip as H
    g = torch.Generator().manual_seed(5)
    rays, rgbs = torch.randn(10007, 6, generator=g).to(DEV), torch.rand(10007, 3, generator=g).to(DEV)
    ids = torch.randint(0, 10007, (4096,), generator=g).to(DEV)
    ids[7] = -1
    out_r, out_c = torch.zeros(4096, 6, device=DEV), torch.zeros(4096, 3, device=DEV)
    H.check(H.lib().tf_gather_batch(rays.data_ptr(), rgbs.data_ptr(), 10007, ids.data_ptr(), 4096, out_r.data_ptr(),
                                    out_c.data_ptr(), torch.cuda.current_stream().cuda_stream), "tf_gather_batch")
    assert torch.equal(out_r, rays[ids]) and torch.equal(out_c, rgbs[ids])
    # the staged form: the same gather, and the step's host-drawn jitter read out of pinned host memory by the same launch
    jit = torch.rand(4096, 1, generator=g).pin_memory()
    out_r.zero_(); out_c.zero_()
    out_j = torch.zeros(4096, device=DEV)
    # ... and a weight-pack job (padded copy, padded transpose, zeroed counter block) run by extra workgroups of that launch
    w = torch.randn(37, 21, generator=g).to(DEV)
    dst, dst_t = torch.full((48, 32), 7.0, device=DEV), torch.full((32, 48), 7.0, device=DEV)
    ctr = torch.full((1000,), 5, dtype=torch.int32, device=DEV)
    job = H.TfPackJob()
    job.n, job.zero, job.n_zero = 2, ctr.data_ptr(), ctr.numel()
    for it, (d, tr) in zip(job.item, ((dst, 0), (dst_t, 1))):
        it.src, it.dst, it.rows, it.cols, it.rows_pad, it.transpose = w.data_ptr(), d.data_ptr(), 37, 21, 48, tr
    H.check(H.lib().tf_gather_batch_staged(rays.data_ptr(), rgbs.data_ptr(), 10007, ids.data_ptr(), 4096, out_r.data_ptr(),
                                           out_c.data_ptr(), jit.data_ptr(), out_j.data_ptr(), 4096, C.byref(job),
                                           torch.cuda.current_stream().cuda_stream), "tf_gather_batch_staged")
    torch.cuda.synchronize()
    assert torch.equal(out_r, rays[ids]) and torch.equal(out_c, rgbs[ids]) and torch.equal(out_j.cpu(), jit.view(-1))
    want = torch.zeros(48, 32, device=DEV)
    want[:37, :21] = w
    assert torch.equal(dst, want) and torch.equal(dst_t, want.t()) and int(ctr.abs().sum()) == 0
    # without a batch: jitter and pack job only (the staging of steps whose batch arrives by plain copies)
    out_j.zero_(); dst.fill_(7.0)
    H.check(H.lib().tf_gather_batch_staged(None, None, 0, None, 0, None, None, jit.data_ptr(), out_j.data_ptr(), 4096,
                                           C.byref(job), torch.cuda.current_stream().cuda_stream), "tf_gather_batch_staged")
    torch.cuda.synchronize()
    assert torch.equal(out_j.cpu(), jit.view(-1)) and torch.equal(dst, want)


@pytest.mark.parametrize("n_rays", [1, 3, 9, 70])
def test_gradients_of_very_small_batches(recon, n_rays):
    """A handful of rays: the eight entry shards hold few samples each, so the shading backward's 64-sample chunks
    straddle shard boundaries (and skip empty shards).  Gradients against the oracle's autograd."""
    c, model = _model(recon, "vm_cubic_train")
    rays = c.rays.to(DEV)[: n_rays * 5 : 5].contiguous()
    target = torch.rand(rays.shape[0], 3, generator=torch.Generator().manual_seed(n_rays)).to(DEV)
    torch.manual_seed(7)
    rgb, _, nv = model(rays, None, white_bg=True, is_train=True)
    torch.mean((rgb - target) ** 2).backward()
    cfg, params = oracle_of(model, DEV)
    for p in params.values():
        p.requires_grad_(True)
    torch.manual_seed(7)
    o_rgb, _, o_n = R.render_rays(cfg, params, rays, None, white_bg=True, is_train=True, n_samples=-1)
    torch.mean((o_rgb - target) ** 2).backward()
    assert int(nv) == int(o_n)
    np.testing.assert_allclose(rgb.detach().cpu().numpy(), o_rgb.detach().cpu().numpy(), rtol=RTOL, atol=ATOL_RGB)
    top = max(float(params[k].grad.abs().max()) for k, _ in model.named_parameters())
    for k, p in model.named_parameters():
        og = params[k].grad
        if float(og.abs().max()) < 1e-6 * top:
            continue
        rel = (p.grad - og).norm().item() / og.norm().item()
        assert rel <= 2e-3, (k, rel)
