"""alpha-volume rebuild, ray filtering, shrink, grid up-sampling and the regularisers
against golden outputs of the reference's own methods (tests/golden/lifecycle.npz)."""
import numpy as np
import pytest
import torch

from tests._golden import _npz

ARGS = dict(step_ratio=0.5, fea2denseAct="softplus", density_n_comp=[8, 8, 8], app_n_comp=[16, 16, 16], app_dim=27,
            density_shift=-10.0, distance_scale=25.0, alphaMask_thres=0.001, shadingMode="MLP_Fea", pos_pe=2, view_pe=2,
            fea_pe=2, featureC=64)
CUBE = [[-1.5, -1.5, -1.5], [1.5, 1.5, 1.5]]


def make(recon, z, device, prefix="state0/"):
    model = recon.TensorVMSplit(ARGS, torch.tensor(CUBE, device=device), [32, 32, 32], [2.0, 6.0], device)
    model.load_state_dict({k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)})
    model.alphaMask = recon.AlphaGridMask(device, torch.tensor(CUBE, device=device),
                                          torch.from_numpy(z["alpha0"]).float().to(device))
    return model


def test_shrink_and_upsample_on_cpu(recon):
    """Host-side schedule steps (torch crop / resize on channel-last parameters) reproduce the reference."""
    z = _npz("lifecycle")
    model = make(recon, z, "cpu")
    model.alphaMask = recon.AlphaGridMask("cpu", torch.tensor(CUBE), torch.from_numpy(z["upd/alpha"]).float())
    model.shrink(torch.from_numpy(z["upd/new_aabb"]))
    assert model.gridSize.tolist() == z["shrunk/gridSize"].tolist()
    assert np.allclose(model.aabb.numpy(), z["shrunk/aabb"], atol=0, rtol=0)
    assert float(model.stepSize) == float(z["shrunk/stepSize"]) and model.nSamples == int(z["shrunk/nSamples"])
    for k, p in model.named_parameters():
        assert np.array_equal(p.detach().numpy(), z["shrunk/" + k]), k
        if "_plane." in k or "_line." in k:
            assert recon.is_channel_last(p), k
    model.upsample_volume_grid([36, 40, 30])
    assert float(model.stepSize) == float(z["up/stepSize"]) and model.nSamples == int(z["up/nSamples"])
    for k, p in model.named_parameters():
        if k.startswith("density_") or k.startswith("app_"):
            np.testing.assert_allclose(p.detach().numpy(), z["up/" + k], rtol=1e-6, atol=1e-7, err_msg=k)
            assert recon.is_channel_last(p), k
    groups = model.get_optparam_groups(0.02, 1e-3)
    assert sum(p.numel() for g in groups for p in g["params"]) == sum(p.numel() for p in model.parameters())


def test_regularisers_match_reference(recon):
    z = _npz("lifecycle")
    model = make(recon, z, "cpu")
    assert abs(float(model.vector_comp_diffs()) - float(z["reg/vector_comp_diffs"])) < 1e-6
    assert abs(float(model.density_L1()) - float(z["reg/density_L1"])) < 1e-6
    x = model.density_plane[0]
    h = ((x[:, :, 1:, :] - x[:, :, :-1, :]) ** 2).sum() / (x.shape[1] * (x.shape[2] - 1) * x.shape[3])
    w = ((x[:, :, :, 1:] - x[:, :, :, :-1]) ** 2).sum() / (x.shape[1] * x.shape[2] * (x.shape[3] - 1))
    tv = recon.TVLoss()
    assert abs(float(tv(x)) - float(2 * (h + w))) < 1e-6                      # loss.py:125-141
    assert abs(float(model.TV_loss_density(tv)) - sum(float(tv(p)) * 1e-2 for p in model.density_plane)) < 1e-7
    loss = model.TV_loss_app(tv) + model.density_L1() + model.vector_comp_diffs()
    loss.backward()
    assert all(p.grad is not None for p in list(model.app_plane) + list(model.density_line))


@pytest.mark.gpu
def test_alpha_rebuild_filtering_and_schedule_on_gpu(recon):
    z = _npz("lifecycle")
    dev = "cuda:0"
    model = make(recon, z, dev)
    pts = torch.from_numpy(z["pts"]).to(dev)
    np.testing.assert_allclose(model.alphaMask.sample_alpha(pts).cpu().numpy(), z["sample_alpha"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(model.compute_alpha(pts, None, model.stepSize).cpu().numpy(), z["compute_alpha"],
                               rtol=1e-4, atol=1e-6)
    # filtering_rays: identical survivor sets
    frays = torch.from_numpy(z["frays"])
    idx = torch.arange(frays.shape[0]).float()[:, None]
    _, kept = model.filtering_rays(frays, idx, bbox_only=True)
    assert kept.view(-1).long().tolist() == z["filter_bbox_kept"].tolist()
    _, kept = model.filtering_rays(frays, idx, N_samples=64)
    assert kept.view(-1).long().tolist() == z["filter_alpha_kept"].tolist()
    # alpha-volume rebuild: the reference's occupancy volume bit for bit — except voxels whose pooled alpha sits ON the
    # threshold (the lattice alphas are 1 - exp(-sigma * step): the GPU's exp / softplus differ from the CPU's in the
    # last bits), which are listed with their margins like the ReLU ties of tests/test_full_size.py; the tight box must
    # be the reference's exactly unless such a tie sits on its face
    new_aabb = model.updateAlphaMask((20, 24, 28))
    got = model.alphaMask.alpha_volume[0, 0].cpu().numpy() > 0.5
    ref = z["upd/alpha"] > 0
    assert got.shape == ref.shape
    ties = np.argwhere(got != ref)
    if len(ties):
        mdl = make(recon, z, dev)                                    # the state the rebuild started from (alpha0 mask)
        alpha, _ = mdl.getDenseAlpha((20, 24, 28))                   # (gx, gy, gz), tensorBase.py:215-230
        pooled = torch.nn.functional.max_pool3d(alpha.clamp(0, 1).permute(2, 1, 0)[None, None], 3, 1, 1)[0, 0].cpu().numpy()
        margins = np.abs(pooled[tuple(ties.T)] - ARGS["alphaMask_thres"])
        print("alpha-volume threshold ties (z, y, x) / margin:", [(tuple(t), float(m)) for t, m in zip(ties, margins)])
        assert len(ties) <= 3 and (margins <= 1e-6).all(), (ties, margins)
        voxel = 3.0 / np.array([19, 23, 27])
        assert np.all(np.abs(new_aabb.cpu().numpy() - z["upd/new_aabb"]) <= voxel + 1e-6)
    else:
        assert np.array_equal(new_aabb.cpu().numpy(), z["upd/new_aabb"]), (new_aabb, z["upd/new_aabb"])
    # continue the schedule from the reference's own mask / bbox so later steps compare exactly
    model.alphaMask = recon.AlphaGridMask(dev, torch.tensor(CUBE, device=dev), torch.from_numpy(z["upd/alpha"]).float().to(dev))
    model.shrink(torch.from_numpy(z["upd/new_aabb"]).to(dev))
    assert model.gridSize.tolist() == z["shrunk/gridSize"].tolist()
    model.upsample_volume_grid([36, 40, 30])
    assert model.nSamples == int(z["up/nSamples"])
    rays = torch.from_numpy(z["up/rays"]).to(dev)
    with torch.no_grad():
        rgb, depth, nv = model(rays, None, white_bg=True, is_train=False)
    assert int(nv) == int(z["up/num_valid"])
    np.testing.assert_allclose(rgb.cpu().numpy(), z["up/rgb_map"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(depth.cpu().numpy(), z["up/depth_map"], rtol=1e-4, atol=1e-5)
    # and the model still trains after the schedule steps (new parameters, new workspaces)
    opt = torch.optim.Adam(model.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
    rgb, _, _ = model(rays, None, white_bg=True, is_train=True)
    loss = torch.mean((rgb - 0.5) ** 2) + 1e-3 * model.TV_loss_density(recon.TVLoss()) + 1e-4 * model.density_L1()
    opt.zero_grad()
    loss.backward()
    opt.step()
    assert all(torch.isfinite(p).all() for p in model.parameters())


@pytest.mark.gpu
def test_training_harness_runs_the_intended_schedule(recon):
    """harness.train: MSE + regularisers, alpha-mask update + shrink, up-sampling with optimizer rebuild (row f-2)."""
    from recon_amd import synthetic as S, harness
    dev = "cuda:0"
    torch.manual_seed(0)
    aabb = torch.tensor(S.LEGO_AABB, device=dev)
    args = S.lego_args(density_n_comp=(8, 8, 8), app_n_comp=(16, 16, 16))
    args["featureC"] = 64
    teacher = recon.TensorVMSplit(args, aabb, [32] * 3, S.LEGO_NEAR_FAR, dev)
    S.make_trained_like(teacher, recon.AlphaGridMask, mask_res=32)
    rays = S.blender_rays(3, H=48, W=48, seed=3).to(dev)
    with torch.no_grad():
        gt = recon.OctreeRender_trilinear_fast(rays, teacher, chunk=4096, white_bg=True, device=dev)[0]
    student = recon.TensorVMSplit(args, aabb, [16] * 3, S.LEGO_NEAR_FAR, dev)
    cfg = dict(n_iters=200, batch_size=2048, N_voxel_init=16 ** 3, N_voxel_final=32 ** 3, upsamp_list=[140, 170],
               update_AlphaMask_list=[120], TV_weight_density=0.01, TV_weight_app=0.01, L1_weight_inital=8e-5,
               L1_weight_rest=4e-5, Ortho_weight=0.01)
    hist = harness.train(student, rays, gt, cfg, device=dev, log_every=20, seed=1)
    kinds = [e[1] for e in hist["events"]]
    assert kinds == ["shrink", "upsample", "upsample"], hist["events"]
    assert hist["psnr"][-1][1] > hist["psnr"][0][1] + 3.0, hist["psnr"]
    assert student.alphaMask is not None and all(torch.isfinite(p).all() for p in student.parameters())
    p = harness.evaluate_psnr(student, rays[:2304], gt[:2304], device=dev)
    assert p > 15.0, p


@pytest.mark.gpu
@pytest.mark.parametrize("grid,den,app", [([40, 52, 36], [16, 4, 4], [48, 12, 12]), ([64, 64, 64], [16, 16, 16], [48, 48, 48])])
def test_fused_regularisers_match_the_eager_terms(recon, grid, den, app):
    """tf_regularizers (one pass) against the torch terms of train.py:340-371, which test_regularisers_match_reference
    pins to the reference's values: loss, its parts, and the gradient through both entry points."""
    from recon_amd import synthetic as S
    dev = "cuda:0"
    torch.manual_seed(4)
    aabb = torch.tensor(S.LEGO_AABB, device=dev)
    model = recon.TensorVMSplit(S.lego_args(density_n_comp=den, app_n_comp=app), aabb, grid, S.LEGO_NEAR_FAR, dev)
    with torch.no_grad():
        model.density_plane[1][0, 2, 3, 4] = 0.0           # sign(0) = 0 in the L1 gradient
    w = dict(ortho_weight=0.01, l1_weight=8e-5, tv_weight_density=0.013, tv_weight_app=0.007)
    tv = recon.TVLoss()
    parts = (w["ortho_weight"] * model.vector_comp_diffs(), w["l1_weight"] * model.density_L1(),
             w["tv_weight_density"] * model.TV_loss_density(tv) + w["tv_weight_app"] * model.TV_loss_app(tv))
    ref = parts[0] + parts[1] + parts[2]
    names = [n for n, _ in model.named_parameters() if "_plane." in n or "_line." in n]
    ref_g = dict(zip(names, torch.autograd.grad(ref, [dict(model.named_parameters())[n] for n in names])))
    # (1) differentiable entry point, scaled by an upstream factor
    fused = recon.fused_regularizers(model, **w)
    assert abs(fused.item() - ref.item()) <= 2e-6 * abs(ref.item())
    g1 = dict(zip(names, torch.autograd.grad(3.0 * fused, [dict(model.named_parameters())[n] for n in names])))
    # (2) in-place entry point on top of existing gradients
    model.zero_grad()
    for n, p in model.named_parameters():
        if n in names:
            p.grad = torch.full_like(p, 0.5)
    out = recon.add_regularizer_grads_(model, **w)
    assert abs(out[0].item() - ref.item()) <= 2e-6 * abs(ref.item())
    assert abs(out[3].item() - parts[0].item()) <= 2e-6 * abs(parts[0].item())
    assert abs(out[2].item() - parts[1].item()) <= 2e-6 * abs(parts[1].item())
    assert abs(out[1].item() - parts[2].item()) <= 2e-6 * abs(parts[2].item())
    for n in names:
        scale = ref_g[n].abs().max().item()
        assert (g1[n] / 3.0 - ref_g[n]).abs().max().item() <= 1e-5 * scale, n
        p = dict(model.named_parameters())[n]
        assert p.grad.stride() == p.stride()
        assert (p.grad - 0.5 - ref_g[n]).abs().max().item() <= 1e-5 * scale + 1e-7, n
    # (3) the weights as DEVICE values (host weights 1): what a captured step uses while train.py:336-339 decays them
    for n, p in model.named_parameters():
        if n in names:
            p.grad = torch.full_like(p, 0.5)
    wd = torch.tensor([w["ortho_weight"], w["l1_weight"], w["tv_weight_density"], w["tv_weight_app"]], device=dev)
    out3 = recon.add_regularizer_grads_(model, 1.0, 1.0, 1.0, 1.0, weights_dev=wd)
    assert abs(out3[0].item() - ref.item()) <= 2e-6 * abs(ref.item())
    for n in names:
        scale = ref_g[n].abs().max().item()
        p = dict(model.named_parameters())[n]
        assert (p.grad - 0.5 - ref_g[n]).abs().max().item() <= 1e-5 * scale + 1e-7, n


@pytest.mark.gpu
def test_graphed_step_recaptures_after_schedule_steps(recon):
    """updateAlphaMask / upsample_volume_grid replace the mask and the parameters under a captured step
    (train.py:300-311, 403-425): GraphedTrainStep notices, warms up and captures again instead of replaying stale
    pointers; the caller hands it the rebuilt optimizer."""
    from recon_amd import synthetic as S
    dev = "cuda:0"
    torch.manual_seed(0)
    aabb = torch.tensor(S.LEGO_AABB, device=dev)
    args = S.lego_args(density_n_comp=(8, 8, 8), app_n_comp=(16, 16, 16))
    model = recon.TensorVMSplit(args, aabb, [40, 40, 40], S.LEGO_NEAR_FAR, dev)
    S.make_trained_like(model, recon.AlphaGridMask, mask_res=32, radius=0.7)
    allrays = S.blender_rays(1)
    rays = allrays[torch.randperm(allrays.shape[0], generator=torch.Generator().manual_seed(3))[:2048]].to(dev).contiguous()
    target = torch.rand(2048, 3, generator=torch.Generator().manual_seed(4)).to(dev)
    N = 150
    opt = recon.FusedAdam(model.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
    gs = recon.GraphedTrainStep(model, opt, 2048, N, warmup=1)
    for _ in range(4):
        l0 = float(gs.step(rays, target))
    g0 = gs.graph
    assert g0 is not None
    model.updateAlphaMask((32, 32, 32))                      # new AlphaGridMask object
    l1 = float(gs.step(rays, target))                        # eager warm-up in the new state
    l2 = float(gs.step(rays, target))                        # captured again
    assert gs.graph is not None and gs.graph is not g0
    model.upsample_volume_grid([48, 48, 48])                 # new parameters -> new optimizer (train.py:300-311)
    gs.opt = recon.FusedAdam(model.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
    g1 = gs.graph
    for _ in range(3):
        l3 = float(gs.step(rays, target, None))
    assert gs.graph is not None and gs.graph is not g1
    assert all(np.isfinite(v) for v in (l0, l1, l2, l3))
    assert all(torch.isfinite(p).all() for p in model.parameters())


@pytest.mark.gpu
def test_graphed_harness_follows_the_eager_schedule(recon):
    """harness.train(graphed=True): every iteration replayed from a hipGraph (regularisers with device-resident weights,
    re-capture after alpha-mask update / shrink / up-sampling) ends where the eager loop ends."""
    from recon_amd import synthetic as S, harness
    dev = "cuda:0"
    aabb = torch.tensor(S.LEGO_AABB, device=dev)
    args = S.lego_args(density_n_comp=(8, 8, 8), app_n_comp=(16, 16, 16))
    args["featureC"] = 64
    torch.manual_seed(0)
    teacher = recon.TensorVMSplit(args, aabb, [32] * 3, S.LEGO_NEAR_FAR, dev)
    S.make_trained_like(teacher, recon.AlphaGridMask, mask_res=32)
    rays = S.blender_rays(3, H=48, W=48, seed=3).to(dev)
    with torch.no_grad():
        gt = recon.OctreeRender_trilinear_fast(rays, teacher, chunk=4096, white_bg=True, device=dev)[0]
    cfg = dict(n_iters=200, batch_size=2048, N_voxel_init=16 ** 3, N_voxel_final=32 ** 3, upsamp_list=[140, 170],
               update_AlphaMask_list=[120], TV_weight_density=0.01, TV_weight_app=0.01, L1_weight_inital=8e-5,
               L1_weight_rest=4e-5, Ortho_weight=0.01)
    out = {}
    for graphed in (False, True):
        torch.manual_seed(11)
        student = recon.TensorVMSplit(args, aabb, [16] * 3, S.LEGO_NEAR_FAR, dev)
        torch.manual_seed(12)
        hist = harness.train(student, rays, gt, cfg, device=dev, log_every=20, seed=1, graphed=graphed)
        out[graphed] = (hist, harness.evaluate_psnr(student, rays[:2304], gt[:2304], device=dev), student.gridSize.tolist())
    he, pe, ge = out[False]
    hg, pg, gg = out[True]
    assert [e[:2] for e in hg["events"]] == [e[:2] for e in he["events"]] and ge == gg
    assert hg["n_samples"] == he["n_samples"]
    assert abs(hg["psnr"][0][1] - he["psnr"][0][1]) < 0.05, (hg["psnr"][0], he["psnr"][0])      # same first step
    assert pg > 15.0 and abs(pg - pe) < 1.5, (pg, pe)


@pytest.mark.gpu
def test_eval_between_graph_replays_sees_the_current_weights(recon):
    """A replay updates W1 / W2 / basis without bumping their `_version`, and refreshes the padded copies BEFORE its Adam
    update: an eval after k replays must not run on copies that are one optimizer step old (round-1 advisor finding).
    eval, 3 replays, eval, 2 replays, eval — every eval against the oracle evaluated on the model's CURRENT state."""
    from recon_amd import synthetic as S
    from oracle import ref_torch as R
    from tests.helpers import oracle_of
    dev = "cuda:0"
    torch.manual_seed(0)
    aabb = torch.tensor(S.LEGO_AABB, device=dev)
    args = S.lego_args(density_n_comp=(8, 8, 8), app_n_comp=(16, 16, 16))
    model = recon.TensorVMSplit(args, aabb, [40, 40, 40], S.LEGO_NEAR_FAR, dev)
    S.make_trained_like(model, recon.AlphaGridMask, mask_res=32, radius=0.7)
    allrays = S.blender_rays(1)
    rays = allrays[torch.randperm(allrays.shape[0], generator=torch.Generator().manual_seed(3))[:2048]].to(dev).contiguous()
    target = torch.rand(2048, 3, generator=torch.Generator().manual_seed(4)).to(dev)
    N = 150
    # a large network learning rate makes a one-step-old W1 / W2 / basis visible in the image
    opt = recon.FusedAdam(model.get_optparam_groups(0.02, 2e-2), betas=(0.9, 0.99))
    gs = recon.GraphedTrainStep(model, opt, 2048, N, warmup=1)

    def check():
        with torch.no_grad():
            rgb, _, _ = model(rays[:512], None, white_bg=True, is_train=False, N_samples=N)
        cfg, params = oracle_of(model, "cpu")
        ref, _, _ = R.render_rays(cfg, params, rays[:512].cpu(), None, white_bg=True, is_train=False, n_samples=N)
        err = (rgb.cpu() - ref).abs().max().item()
        assert err < 2e-4, err

    check()
    for _ in range(4):          # warm-up, capture, replays
        gs.step(rays, target)
    assert gs.graph is not None
    check()
    for _ in range(2):
        gs.step(rays, target)
    check()
    feat = model.compute_appfeature(torch.rand(64, 3, device=dev) * 2 - 1)      # same packed basis copy
    cfg, params = oracle_of(model, "cpu")
    assert torch.isfinite(feat).all()


@pytest.mark.gpu
def test_training_workspace_is_reclaimed_when_no_backward_runs(recon):
    """A grad-enabled forward whose result is dropped without a backward must not pin its pooled workspace forever
    (round-1 advisor finding): the pool stays at one entry over many such forwards."""
    from recon_amd import synthetic as S
    dev = "cuda:0"
    torch.manual_seed(0)
    aabb = torch.tensor(S.LEGO_AABB, device=dev)
    model = recon.TensorVMSplit(S.lego_args(density_n_comp=(8, 8, 8), app_n_comp=(16, 16, 16)), aabb, [32] * 3,
                                S.LEGO_NEAR_FAR, dev)
    S.make_trained_like(model, recon.AlphaGridMask, mask_res=32, radius=0.7)
    rays = S.blender_rays(1)[:1024].to(dev).contiguous()
    seen = set()
    for _ in range(8):
        rgb, _, _ = model(rays, None, white_bg=True, is_train=True, N_samples=100)
        seen.add(model.last["ws"].buf.data_ptr())
        del rgb
    assert len(seen) <= 2, seen
    pools = list(model._train_ws.values())
    assert len(pools) == 1 and len(pools[0]) <= 2
