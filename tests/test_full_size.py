"""GPU parity at the FULL sizes of BASELINE.json's five configurations (C1..C5), 4096-ray batches.

The committed golden fixtures pin the oracle bit for bit at small grids (tests/test_oracle_golden.py) and the HIP
path against them (tests/test_hip_forward.py).  Here the same oracle runs on the GPU box (device 'cuda', eager
PyTorch-ROCm) on the same seeded inputs as the HIP path at the real sizes: 128^3 / N=443, 300^3 / N=1039, TensorCP
[96]/[288] with the SH head, NDC rays with unequal components, and a 640^3-equivalent non-cubic grid with
near = 0.01.  Bars: bbox / alpha-mask sample masks identical, shaded-sample mask identical up to threshold ties
(eager ROCm transcendental kernels are not bit-identical to the CPU ones the fixtures were made with: at most 3
flips per batch, each within 1e-6 of the 1e-4 threshold), RGB / depth within 1e-4 relative.  Gradients: both sides
sum ~10^5 fp32 terms per entry in different orders, and a hidden unit whose pre-activation is within rounding of 0
passes its gradient on one side only (20 M units per batch: a few such ReLU ties occur; at 300^3 one sample's whole
dL/dV row differed by ~1 % of the tensor's largest entry, identically through the binned and the direct scatter).
The bar is therefore per tensor: relative L2 error <= 2e-3 and largest entry error <= 2e-2 of the tensor's largest
entry (tensors whose gradient is below 1e-6 of the step's largest only get the L2 bar against that global scale).
The small golden cases (tests/test_hip_backward.py) hold the tight bar of 1.5e-4 against the CPU reference."""
import numpy as np
import pytest
import torch

from oracle import ref_torch as R
from tests.helpers import oracle_of
from tests.test_hip_forward import ATOL_RGB, RTOL, bits_to_mask

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TT_AABB = [[-2.4, -1.6, -1.9], [2.2, 1.7, 1.3]]


def _scene(recon, name):
    from recon_amd import synthetic as S
    torch.manual_seed(0)
    ndc, white = False, True
    if name in ("C1_vm128", "C2_vm300"):
        g = 128 if name == "C1_vm128" else 300
        aabb = torch.tensor(S.LEGO_AABB, device=DEV)
        model = recon.TensorVMSplit(S.lego_args(), aabb, recon.N_to_reso(g ** 3, aabb), S.LEGO_NEAR_FAR, DEV)
        rays = S.blender_rays(1)
    elif name.startswith("C3_cp300"):
        aabb = torch.tensor(S.LEGO_AABB, device=DEV)
        head = "SH" if name.endswith("sh") else "MLP_Fea"
        # BASELINE config 3 ([96]/[288], configs/lego.txt:80-83): with the SH head inference only (the reference
        # cannot train that head either); with MLP_Fea also trained — at 288 components the backward's tile does not
        # fit LDS in one piece, which exercises its gather-V-twice layout
        args = S.lego_args(head, density_n_comp=(96,), app_n_comp=(288,))
        model = recon.TensorCP(args, aabb, recon.N_to_reso(300 ** 3, aabb), near_far=S.LEGO_NEAR_FAR, device=DEV)
        rays = S.blender_rays(1)
    elif name == "C4_ndc":
        aabb = torch.tensor(S.LLFF_AABB, device=DEV)
        args = S.lego_args(density_n_comp=(16, 4, 4), app_n_comp=(48, 12, 12))
        model = recon.TensorVMSplit(args, aabb, recon.N_to_reso(300 ** 3, aabb), S.LLFF_NEAR_FAR, DEV)
        rays, ndc, white = S.llff_ndc_rays(1 << 16), True, False
    else:   # C5_tt640
        aabb = torch.tensor(TT_AABB, device=DEV)
        model = recon.TensorVMSplit(S.lego_args(), aabb, recon.N_to_reso(640 ** 3, aabb), S.TT_NEAR_FAR, DEV)
        rays = S.tt_rays(1 << 15, TT_AABB)
    S.make_trained_like(model, recon.AlphaGridMask, radius=0.8 if name != "C4_ndc" else 0.9)
    n_samples = min(int(1e6), recon.cal_n_samples(model.gridSize.tolist(), 0.5))
    perm = torch.randperm(rays.shape[0], generator=torch.Generator().manual_seed(1))[:4096]
    return model, rays[perm].to(DEV).contiguous(), n_samples, ndc, white


@pytest.mark.parametrize("name", ["C1_vm128", "C2_vm300", "C3_cp300_sh", "C3_cp300_mlp", "C4_ndc", "C5_tt640"])
def test_forward_at_baseline_size(recon, name):
    model, rays, N, ndc, white = _scene(recon, name)
    model._debug_masks = True
    with torch.no_grad():
        rgb, depth, nvalid = model(rays, None, white_bg=white, is_train=False, ndc_ray=ndc, N_samples=N)
        cfg, params = oracle_of(model, DEV)
        o_rgb, o_depth, o_n, mid = R.render_rays(cfg, params, rays, None, white_bg=white, is_train=False, ndc_ray=ndc,
                                                 n_samples=N, keep=True)
    torch.cuda.synchronize()
    ws, R_ = model.last["ws"], rays.shape[0]
    assert np.array_equal(bits_to_mask(ws.dbg_bbox, R_, N), mid["bbox_valid"].cpu().numpy()), "bbox mask"
    assert np.array_equal(bits_to_mask(ws.dbg_valid, R_, N), mid["ray_valid"].cpu().numpy()), "alpha-mask / ray_valid mask"
    app, o_app = bits_to_mask(ws.dbg_app, R_, N), mid["app_mask"].cpu().numpy()
    flips = np.argwhere(app != o_app)
    margin = np.abs(mid["weight"].cpu().numpy()[app != o_app] - 1e-4)
    assert len(flips) <= 3 and (margin < 1e-6).all(), (len(flips), margin)
    shaded = int(o_n)
    assert shaded > 4096 and abs(int(nvalid) - shaded) <= 3, (int(nvalid), shaded)
    if len(flips) == 0:     # a flipped sample changes its ray's colour by ~1e-4, above the bar for dark pixels
        np.testing.assert_allclose(rgb.cpu().numpy(), o_rgb.cpu().numpy(), rtol=RTOL, atol=ATOL_RGB)
        np.testing.assert_allclose(depth.cpu().numpy(), o_depth.cpu().numpy(), rtol=RTOL, atol=1e-5)
    else:
        ok = np.ones(R_, bool)
        ok[flips[:, 0]] = False
        np.testing.assert_allclose(rgb.cpu().numpy()[ok], o_rgb.cpu().numpy()[ok], rtol=RTOL, atol=ATOL_RGB)
        np.testing.assert_allclose(depth.cpu().numpy()[ok], o_depth.cpu().numpy()[ok], rtol=RTOL, atol=1e-5)
    print(f"{name}: grid {model.gridSize.tolist()} N={N} shaded/ray={shaded / R_:.1f} flips={len(flips)} "
          f"max|drgb|={(rgb - o_rgb).abs().max().item():.2e} max|ddepth|={(depth - o_depth).abs().max().item():.2e}")


@pytest.mark.parametrize("name", ["C2_vm300", "C3_cp300_mlp", "C4_ndc", "C5_tt640"])
def test_train_gradients_at_baseline_size(recon, name):
    """One training forward/backward (jittered samples, MSE against a random target) against the oracle's autograd."""
    model, rays, N, ndc, white = _scene(recon, name)
    target = torch.rand(rays.shape[0], 3, generator=torch.Generator().manual_seed(5)).to(DEV)
    torch.manual_seed(3)
    if ndc:
        model._jitter_override = torch.rand(1, N)
    model._debug_masks = True
    rgb, _, _ = model(rays, None, white_bg=True, is_train=True, ndc_ray=ndc, N_samples=N)
    app = bits_to_mask(model.last["ws"].dbg_app, rays.shape[0], N)
    loss = torch.mean((rgb - target) ** 2)
    loss.backward()
    cfg, params = oracle_of(model, DEV)
    for p in params.values():
        p.requires_grad_(True)
    torch.manual_seed(3)
    jit = torch.rand(1, N).to(DEV) if ndc else None
    o_rgb, _, _, mid = R.render_rays(cfg, params, rays, None, white_bg=True, is_train=True, ndc_ray=ndc, n_samples=N,
                                     jitter=jit, keep=True)
    o_loss = torch.mean((o_rgb - target) ** 2)
    o_loss.backward()
    assert abs(loss.item() - o_loss.item()) <= 1e-5 * abs(o_loss.item())
    # a shaded-sample threshold tie (see the module docstring) adds or removes one sample's whole contribution on
    # one side: with ties the factor-tensor gradients are compared in the L2 sense only, at one sample's weight
    flips = int((app != mid["app_mask"].cpu().numpy()).sum())
    assert flips <= 3, flips
    l2_bar, max_bar = (2e-3, 2e-2) if flips == 0 else (2e-2, None)
    top = max(params[k].grad.abs().max().item() for k, _ in model.named_parameters())
    worst = {}
    for k, p in model.named_parameters():
        g, og = p.grad, params[k].grad
        scale = og.abs().max().item()
        assert scale > 0, k
        err = (g - og)
        worst[k] = err.abs().max().item() / scale
        if scale >= 1e-6 * top:
            assert max_bar is None or worst[k] <= max_bar, (k, worst[k], scale)
            assert err.norm().item() <= l2_bar * og.norm().item(), (k, err.norm().item(), og.norm().item())
        else:
            assert err.norm().item() <= l2_bar * max(og.norm().item(), 1e-6 * top), (k, err.norm().item(), og.norm().item())
    print(name, f"mask flips {flips}; worst gradient error / max|grad|:", max(worst.values()), max(worst, key=worst.get))
