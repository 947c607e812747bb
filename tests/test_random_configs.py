"""GPU parity on randomly drawn configurations: grid sizes that are not multiples of the sort's tiles, component counts
that are not multiples of 4 or 16 (scalar code paths), both decompositions, every trainable head, featureC 64 / 128,
ReLU / softplus density, NDC and box rays, with and without alpha mask.  Forward: sample masks identical (threshold
ties tolerated as in test_full_size), RGB / depth within 1e-4.  Backward (trainable heads): per-tensor relative L2
error <= 2e-3 against the oracle's autograd on the same device."""
import numpy as np
import pytest
import torch

from oracle import ref_torch as R
from tests.helpers import oracle_of
from tests.test_hip_forward import ATOL_RGB, RTOL, bits_to_mask

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _draw(seed):
    from recon_amd import synthetic as S
    g = np.random.default_rng(seed)
    cp = bool(g.integers(0, 4) == 0)
    head = str(g.choice(["MLP_Fea", "MLP_PE", "MLP"]))
    comps = lambda choices: [int(g.choice(choices))] * (1 if cp else 1) if cp else [int(g.choice(choices)) for _ in range(3)]
    den = comps([4, 6, 8, 16, 20]) if not cp else [int(g.choice([8, 12, 30]))]
    app = comps([8, 12, 18, 24, 48]) if not cp else [int(g.choice([24, 40, 52]))]
    if seed >= 100:     # wide appearance bases: the shading backward's gather-V-twice LDS layout (shade_bwd.hip bwd_lds)
        app = [int(g.choice([64, 72, 96, 128])) for _ in range(3)] if not cp else [int(g.choice([192, 288, 384]))]
    args = dict(step_ratio=float(g.choice([0.5, 0.8])), fea2denseAct=str(g.choice(["softplus", "relu"])),
                density_n_comp=den, app_n_comp=app, app_dim=int(g.choice([12, 16, 27])), density_shift=-10.0,
                distance_scale=25.0, alphaMask_thres=0.001, shadingMode=head, pos_pe=int(g.integers(0, 4)),
                view_pe=int(g.integers(0, 4)), fea_pe=int(g.integers(0, 3)), featureC=int(g.choice([64, 128])))
    grid = [int(g.integers(17, 70)) for _ in range(3)]
    ndc = bool(g.integers(0, 5) == 0) and not cp
    mask = bool(g.integers(0, 3) > 0)
    return args, grid, cp, ndc, mask


def _build(recon, seed):
    from recon_amd import synthetic as S
    args, grid, cp, ndc, use_mask = _draw(seed)
    torch.manual_seed(seed)
    aabb = torch.tensor(S.LLFF_AABB if ndc else [[-1.4, -1.1, -1.5], [1.2, 1.5, 1.3]], device=DEV)
    nf = S.LLFF_NEAR_FAR if ndc else [2.0, 6.0]
    if cp:
        model = recon.TensorCP(args, aabb, grid, near_far=nf, device=DEV)
    else:
        model = recon.TensorVMSplit(args, aabb, grid, nf, DEV)
    S.make_trained_like(model, recon.AlphaGridMask, mask_res=int(24 + seed % 17), radius=0.9)
    if args["fea2denseAct"] == "relu":      # relu(f - ...) needs positive features to have any density
        with torch.no_grad():
            for p in (model.density_line if cp else model.density_plane):
                p.abs_()
            for p in model.density_line:
                p.abs_()
    if not use_mask:
        model.alphaMask = None
    R_ = 700 + 37 * (seed % 5)
    rays = (S.llff_ndc_rays(R_, seed=seed) if ndc else S.blender_rays(1)[seed::997][:R_]).to(DEV).contiguous()
    mask = None
    if seed % 3 == 0:     # frequency / component masks of the training loop (utils.get_free_mask, train.py:303-318)
        mask = recon.get_free_mask(pos_bl=model.pos_bit_length, view_bl=model.view_bit_length, fea_bl=model.fea_bit_length,
                                   den_bl=model.density_n_comp, app_bl=model.app_n_comp, step=37 * (seed % 7) + 5,
                                   total_step=300, device=DEV)
    return model, rays, ndc, args, mask


@pytest.mark.parametrize("seed", list(range(40)) + list(range(100, 108)))
def test_random_configuration(recon, seed):
    model, rays, ndc, args, mask = _build(recon, seed)
    N = min(int(model.nSamples), 400)
    white = not ndc
    model._debug_masks = True
    with torch.no_grad():
        rgb, depth, nvalid = model(rays, mask, white_bg=white, is_train=False, ndc_ray=ndc, N_samples=N)
        cfg, params = oracle_of(model, DEV)
        o_rgb, o_depth, o_n, mid = R.render_rays(cfg, params, rays, mask, white_bg=white, is_train=False, ndc_ray=ndc,
                                                 n_samples=N, keep=True)
    ws, R_ = model.last["ws"], rays.shape[0]
    assert np.array_equal(bits_to_mask(ws.dbg_bbox, R_, N), mid["bbox_valid"].cpu().numpy()), "bbox mask"
    assert np.array_equal(bits_to_mask(ws.dbg_valid, R_, N), mid["ray_valid"].cpu().numpy()), "ray_valid mask"
    app, o_app = bits_to_mask(ws.dbg_app, R_, N), mid["app_mask"].cpu().numpy()
    flips = np.argwhere(app != o_app)
    assert len(flips) <= 2, len(flips)
    ok = np.ones(R_, bool)
    ok[flips[:, 0]] = False
    np.testing.assert_allclose(rgb.cpu().numpy()[ok], o_rgb.cpu().numpy()[ok], rtol=RTOL, atol=ATOL_RGB)
    np.testing.assert_allclose(depth.cpu().numpy()[ok], o_depth.cpu().numpy()[ok], rtol=RTOL, atol=1e-5)
    shaded = int(o_n)
    # ---- backward
    model._debug_masks = False
    target = torch.rand(R_, 3, generator=torch.Generator().manual_seed(seed)).to(DEV)
    torch.manual_seed(100 + seed)
    if ndc:
        model._jitter_override = torch.rand(1, N)
    out, _, _ = model(rays, mask, white_bg=True, is_train=True, ndc_ray=ndc, N_samples=N)
    loss = torch.mean((out - target) ** 2)
    loss.backward()
    for p in params.values():
        p.requires_grad_(True)
    torch.manual_seed(100 + seed)
    jit = torch.rand(1, N).to(DEV) if ndc else None
    o_out, _, _ = R.render_rays(cfg, params, rays, mask, white_bg=True, is_train=True, ndc_ray=ndc, n_samples=N, jitter=jit)
    o_loss = torch.mean((o_out - target) ** 2)
    o_loss.backward()
    assert abs(loss.item() - o_loss.item()) <= 2e-5 * abs(o_loss.item())
    worst = 0.0
    top = max(float(params[k].grad.abs().max()) for k, _ in model.named_parameters() if params[k].grad is not None)
    for k, p in model.named_parameters():
        og = params[k].grad
        if og is None or float(og.abs().max()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        if float(og.abs().max()) < 1e-5 * top:
            # e.g. the density factors of a saturated ReLU field (sigma * delta * 25 >> 1): d alpha / d sigma ~ e^-30, the
            # gradient (1e-14 against 1e-7 elsewhere) is the rounding residue of the `1 - alpha + 1e-10` terms on both
            # sides — only its size is compared
            assert float(p.grad.abs().max()) <= 10 * float(og.abs().max()) + 1e-30, k
            continue
        rel = (p.grad - og).norm().item() / og.norm().item()
        worst = max(worst, rel)
        assert rel <= 2e-3 or len(flips) > 0, (k, rel)
    print(f"seed {seed}: {type(model).__name__} {args['shadingMode']} C={args['density_n_comp']}/{args['app_n_comp']} "
          f"grid {model.gridSize.tolist()} ndc={ndc} alpha={model.alphaMask is not None} free_mask={mask is not None} shaded/ray={shaded / R_:.1f} "
          f"flips={len(flips)} worst grad L2 rel={worst:.1e}")


@pytest.mark.parametrize("grid,N", [([2, 2, 2], 64), ([3, 9, 5], 1), ([9, 4, 17], 2), ([8, 8, 8], 65), ([16, 7, 24], 129)])
def test_tiny_grids_and_sample_counts(recon, grid, N):
    """Degenerate sizes: grids smaller than one sort tile / line bucket, one or two samples per ray, sample counts just
    past a wave (65) and two waves (129).  Forward against the oracle, backward runs and matches in the L2 sense."""
    from recon_amd import synthetic as S
    torch.manual_seed(11)
    aabb = torch.tensor(S.LEGO_AABB, device=DEV)
    model = recon.TensorVMSplit(S.lego_args(density_n_comp=(8, 4, 8), app_n_comp=(16, 8, 12)), aabb, grid, S.LEGO_NEAR_FAR, DEV)
    with torch.no_grad():
        for p in model.density_plane:
            p.fill_(2.0)
        for p in model.density_line:
            p.fill_(1.0)
    rays = S.blender_rays(1)[5::1201][:333].to(DEV).contiguous()
    with torch.no_grad():
        rgb, depth, nvalid = model(rays, None, white_bg=True, is_train=False, N_samples=N)
        cfg, params = oracle_of(model, DEV)
        o_rgb, o_depth, o_n = R.render_rays(cfg, params, rays, None, white_bg=True, is_train=False, n_samples=N)
    assert abs(int(nvalid) - int(o_n)) <= 1
    bad = int(((rgb - o_rgb).abs() > ATOL_RGB + RTOL * o_rgb.abs()).any(-1).sum())
    assert bad <= 1, bad                     # (one threshold tie at most)
    torch.manual_seed(5)
    out, _, _ = model(rays, None, white_bg=True, is_train=True, N_samples=N)
    out.square().mean().backward()
    for p in params.values():
        p.requires_grad_(True)
    torch.manual_seed(5)
    o_out, _, _ = R.render_rays(cfg, params, rays, None, white_bg=True, is_train=True, n_samples=N)
    o_out.square().mean().backward()
    top = max(float(v.grad.abs().max()) for v in params.values() if v.grad is not None)
    for k, p in model.named_parameters():
        og = params[k].grad
        if og is None or float(og.abs().max()) < 1e-5 * top:
            continue
        assert (p.grad - og).norm().item() <= 5e-3 * og.norm().item(), k


@pytest.mark.parametrize("seed", [0, 1, 4, 7, 11, 13])
def test_lifecycle_kernels_on_random_configurations(recon, seed):
    """tf_alpha_points (compute_alpha), tf_sample_alpha_points (AlphaGridMask.sample_alpha) and tf_filter_rays
    (filtering_rays) against the oracle's pieces on random fields, points inside and outside the boxes."""
    model, rays, ndc, args, _ = _build(recon, seed)
    if model.alphaMask is None:
        from recon_amd import synthetic as S
        S.make_trained_like(model, recon.AlphaGridMask, mask_res=29, radius=0.85)
    cfg, params = oracle_of(model, DEV)
    g = torch.Generator().manual_seed(seed)
    lo, hi = model.aabb[0].cpu(), model.aabb[1].cpu()
    pts = (lo + (hi - lo) * (torch.rand(5000, 3, generator=g) * 1.3 - 0.15)).to(DEV)     # 15 % margin outside the box
    with torch.no_grad():
        # sample_alpha: trilinear look-up of the mask volume, zero outside
        got = model.alphaMask.sample_alpha(pts)
        ref = R.alpha_lookup(cfg, pts)
        np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-6)
        # compute_alpha: mask test, density, activation, 1 - exp(-sigma * length)   (tensorBase.py:298-318)
        length = float(model.stepSize)
        got = model.compute_alpha(pts, None, length)
        keep = ref > 0
        sigma = torch.zeros(pts.shape[0], device=DEV)
        if keep.any():
            sigma[keep] = R.feature2density(cfg, R.density_feature(cfg, params, R.normalize_coord(cfg, pts[keep])))
        want = 1 - torch.exp(-sigma * length)
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=1e-6)
    if ndc:
        return
    # filtering_rays (tensorBase.py:259-288): bbox slab test, and "any of N eval samples hits the mask"
    idx = torch.arange(rays.shape[0]).float()[:, None]
    frays = rays.cpu()
    _, kept = model.filtering_rays(frays, idx, bbox_only=True)
    o, d = rays[:, :3], rays[:, 3:6]
    vec = torch.where(d == 0, torch.full_like(d, 1e-6), d)
    ra, rb = (model.aabb[1] - o) / vec, (model.aabb[0] - o) / vec
    want = torch.maximum(ra, rb).amin(-1) > torch.minimum(ra, rb).amax(-1)
    assert kept.view(-1).long().tolist() == torch.nonzero(want.cpu()).view(-1).tolist()
    _, kept = model.filtering_rays(frays, idx, N_samples=48)
    with torch.no_grad():
        p, _, _ = R.sample_ray(cfg, o, d, False, 48)
        want = (R.alpha_lookup(cfg, p.reshape(-1, 3)).view(p.shape[:-1]) > 0).any(-1)
    assert kept.view(-1).long().tolist() == torch.nonzero(want.cpu()).view(-1).tolist()
