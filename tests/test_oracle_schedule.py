"""The oracle's schedule steps (oracle/ref_torch.py dense_alpha / update_alpha_mask / upsample_params, restating
models/tensorBase.py:215-256 and models/tensoRF.py:268-288) against the reference's own outputs in
tests/golden/lifecycle.npz (written by tests/golden/gen_golden.py from the imported reference on the CPU).  These are
the steps the PSNR-parity run (tests/psnr_parity.py) lets its eager student take, so they are pinned here, on the CPU,
bit for bit."""
import numpy as np
import torch

from oracle import ref_torch as R
from tests._golden import _npz

CUBE = [[-1.5, -1.5, -1.5], [1.5, 1.5, 1.5]]


def _cfg(grid, aabb=CUBE):
    return R.FieldCfg(model="TensorVMSplit", aabb=torch.tensor(aabb), gridSize=list(grid), near_far=[2.0, 6.0],
                      step_ratio=0.5, fea2denseAct="softplus", density_n_comp=[8, 8, 8], app_n_comp=[16, 16, 16],
                      app_dim=27, density_shift=-10.0, distance_scale=25.0, shadingMode="MLP_Fea", pos_pe=2, view_pe=2,
                      fea_pe=2, featureC=64).finalize()


def _state(z, prefix):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files
            if k.startswith(prefix) and z[k].ndim >= 1 and k[len(prefix):] not in ("aabb", "gridSize")}


def test_update_alpha_mask_is_the_reference_volume_and_box():
    """updateAlphaMask((20, 24, 28)) of the reference on the fixture's 32^3 field: same 0/1 volume, same tight box,
    no tolerance (every operation is elementwise or a max; the oracle performs them in the reference's order)."""
    z = _npz("lifecycle")
    cfg = _cfg([32, 32, 32])
    cfg.alpha_volume, cfg.alpha_aabb = torch.from_numpy(z["alpha0"]).float(), torch.tensor(CUBE)
    params = _state(z, "state0/")
    box = R.update_alpha_mask(cfg, params, (20, 24, 28), 0.001)
    got = cfg.alpha_volume.numpy() > 0.5
    assert got.shape == z["upd/alpha"].shape
    assert np.array_equal(got, z["upd/alpha"] > 0), int((got != (z["upd/alpha"] > 0)).sum())
    assert np.array_equal(box.numpy(), z["upd/new_aabb"]), (box, z["upd/new_aabb"])
    assert torch.equal(cfg.alpha_aabb, torch.tensor(CUBE))


def test_upsample_params_is_the_reference_resize():
    """upsample_volume_grid([36, 40, 30]) of the reference on its own shrunk state: every factor tensor bit-equal, same
    stepSize / nSamples; and the up-sampled field renders the reference's rgb / depth on the fixture's rays."""
    z = _npz("lifecycle")
    cfg = _cfg(z["shrunk/gridSize"].tolist(), z["shrunk/aabb"].tolist())
    assert float(cfg.stepSize) == float(z["shrunk/stepSize"]) and cfg.nSamples == int(z["shrunk/nSamples"])
    cfg.alpha_volume, cfg.alpha_aabb = torch.from_numpy(z["upd/alpha"]).float(), torch.tensor(CUBE)
    params = _state(z, "shrunk/")
    up = R.upsample_params(cfg, params, [36, 40, 30])
    assert cfg.gridSize == [36, 40, 30]
    assert float(cfg.stepSize) == float(z["up/stepSize"]) and cfg.nSamples == int(z["up/nSamples"])
    n = 0
    for k, v in up.items():
        if k.startswith("density_") or k.startswith("app_"):
            assert np.array_equal(v.detach().numpy(), z["up/" + k]), k
            n += 1
    assert n == 12
    with torch.no_grad():
        rgb, depth, nv = R.render_rays(cfg, up, torch.from_numpy(z["up/rays"]), None, white_bg=True, is_train=False)
    assert int(nv) == int(z["up/num_valid"])
    np.testing.assert_allclose(rgb.numpy(), z["up/rgb_map"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(depth.numpy(), z["up/depth_map"], rtol=1e-6, atol=1e-6)


def test_first_iterations_of_a_fresh_field_follow_the_reference_adam_trajectory():
    """tests/golden/adam_trajectory.npz: 16 training steps of the reference with torch.optim.Adam from a fresh init.
    For the first steps nothing is shaded, the appearance tensors / basis / MLP get no gradient (None) and Adam skips
    them (tensorBase.py:370, train.py:374-376).  The oracle, being the same eager graph, must show the same None pattern,
    losses, sample counts, per-parameter step counts and final parameters."""
    z = _npz("adam_trajectory")
    names = [str(n) for n in z["names"]]
    cfg = R.FieldCfg(model="TensorVMSplit", aabb=torch.tensor(CUBE), gridSize=[24, 24, 24], near_far=[2.0, 6.0],
                     density_n_comp=[8, 8, 8], app_n_comp=[8, 8, 8], featureC=64).finalize()
    p = {k: torch.from_numpy(z["state0/" + k]).clone().requires_grad_(True) for k in names}
    fast = [v for k, v in p.items() if "_plane." in k or "_line." in k]
    slow = [v for k, v in p.items() if not ("_plane." in k or "_line." in k)]
    opt = torch.optim.Adam([{"params": fast, "lr": 0.02}, {"params": slow, "lr": 1e-3}], betas=(0.9, 0.99))
    rays, target = torch.from_numpy(z["rays"]), torch.from_numpy(z["target"])
    for it in range(len(z["loss"])):
        torch.manual_seed(1000 + it)
        rgb, _, nv = R.render_rays(cfg, p, rays, None, white_bg=True, is_train=True)
        loss = torch.mean((rgb - target) ** 2)
        opt.zero_grad()
        loss.backward()
        assert [int(p[k].grad is None) for k in names] == z["grad_is_none"][it].tolist(), it
        assert int(nv) == int(z["num_valid"][it]), (it, int(nv))
        assert abs(loss.item() - float(z["loss"][it])) <= 1e-6 * max(1.0, abs(float(z["loss"][it]))), it
        opt.step()
    assert int(z["grad_is_none"][:, names.index("basis_mat.weight")].sum()) >= 3      # the fixture has a closed phase
    for k in names:
        assert float(opt.state[p[k]]["step"]) == float(z["adam_step/" + k]), k
        np.testing.assert_allclose(p[k].detach().numpy(), z["final/" + k], rtol=0, atol=2e-5 * np.abs(z["final/" + k]).max(),
                                   err_msg=k)


def test_shrink_params_is_the_reference_crop():
    """shrink(new_aabb) of the reference (tensoRF.py:291-327) on the fixture: every cropped tensor bit-equal, same box,
    grid, stepSize and nSamples."""
    z = _npz("lifecycle")
    cfg = _cfg([32, 32, 32])
    cfg.alpha_volume, cfg.alpha_aabb = torch.from_numpy(z["upd/alpha"]).float(), torch.tensor(CUBE)
    out = R.shrink_params(cfg, _state(z, "state0/"), torch.from_numpy(z["upd/new_aabb"]))
    assert cfg.gridSize == z["shrunk/gridSize"].tolist()
    assert np.array_equal(cfg.aabb.numpy(), z["shrunk/aabb"])
    assert float(cfg.stepSize) == float(z["shrunk/stepSize"]) and cfg.nSamples == int(z["shrunk/nSamples"])
    for k, v in out.items():
        assert np.array_equal(v.detach().numpy(), z["shrunk/" + k]), k


def test_filter_rays_keeps_the_reference_set():
    z = _npz("lifecycle")
    cfg = _cfg([32, 32, 32])
    cfg.alpha_volume, cfg.alpha_aabb = torch.from_numpy(z["alpha0"]).float(), torch.tensor(CUBE)
    rays = torch.from_numpy(z["frays"])
    assert torch.nonzero(R.filter_rays(cfg, rays, bbox_only=True)).view(-1).tolist() == z["filter_bbox_kept"].tolist()
    assert torch.nonzero(R.filter_rays(cfg, rays, n_samples=64)).view(-1).tolist() == z["filter_alpha_kept"].tolist()


def test_regulariser_terms_are_the_reference_values():
    """TVLoss values + gradients (aux_refs.npz: loss.py:120-141 run by the reference) and the line / L1 terms of the
    lifecycle fixture (tensoRF.py:175-195)."""
    a = _npz("aux_refs")
    for tag in ("a", "b", "c"):
        x = torch.from_numpy(a[f"tv/{tag}/x"]).requires_grad_(True)
        y = R.tv_loss(x)
        y.backward()
        assert abs(float(y) - float(a[f"tv/{tag}/loss"])) <= 1e-6 * max(1.0, abs(float(a[f"tv/{tag}/loss"])))
        np.testing.assert_allclose(x.grad.numpy(), a[f"tv/{tag}/grad"], rtol=1e-6, atol=1e-9)
    z = _npz("lifecycle")
    p = _state(z, "state0/")
    assert abs(float(R.vector_comp_diffs(p)) - float(z["reg/vector_comp_diffs"])) < 1e-6
    assert abs(float(R.density_l1(p)) - float(z["reg/density_L1"])) < 1e-6
