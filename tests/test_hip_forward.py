"""GPU parity: the HIP path (through the C ABI) against the golden vectors of the reference and against
the oracle.  Sample masks must be bit-exact; RGB / depth within 1e-4 relative fp32 (north_star)."""
import numpy as np
import pytest
import torch

from oracle import ref_torch as R
from tests._golden import CASES, Case
from tests.helpers import build_model

pytestmark = pytest.mark.gpu
RTOL = 1e-4          # north_star tolerance on rendered RGB / depth
ATOL_RGB = 1e-5      # absolute floor for RGB values near 0 (values live in [0,1])


def bits_to_mask(words_i32, R_, N):
    w = words_i32.cpu().numpy().view(np.uint64).reshape(R_, -1)
    bits = np.unpackbits(w.view(np.uint8), axis=1, bitorder="little")
    return bits[:, :N].astype(bool)


def run_hip(recon, c: Case, device="cuda:0"):
    model = build_model(recon, c, device)
    model._debug_masks = True
    call = c.call
    torch.manual_seed(call["seed"])
    if call["ndc_ray"] and call["is_train"]:
        # the reference draws the shared (1,N) jitter with rand_like on the rays' device; replay the CPU stream
        n = call["N_samples"] if call["N_samples"] > 0 else model.nSamples
        model._jitter_override = torch.rand(1, n)
    with torch.no_grad():
        rgb, depth, nvalid = model(c.rays.to(device), c.mask_to(device), white_bg=call["white_bg"],
                                   is_train=call["is_train"], ndc_ray=call["ndc_ray"], N_samples=call["N_samples"])
    torch.cuda.synchronize()
    return model, rgb.cpu(), depth.cpu(), int(nvalid)


def packed_in_ray_order(ws, field):
    off, cnt = ws.app_offset.cpu().numpy(), ws.app_count.cpu().numpy()
    idx = np.concatenate([np.arange(o, o + k) for o, k in zip(off, cnt)]) if cnt.sum() else np.zeros(0, np.int64)
    return field.cpu().numpy()[idx]


@pytest.mark.parametrize("name", CASES)
def test_forward_parity(recon, name):
    c = Case(name)
    model, rgb, depth, nvalid = run_hip(recon, c)
    ws = model.last["ws"]
    R_, N = c.shape
    # --- masks: bit-exact
    assert np.array_equal(bits_to_mask(ws.dbg_bbox, R_, N), c.expect_mask("mid/bbox_valid")), "bbox mask"
    assert np.array_equal(bits_to_mask(ws.dbg_valid, R_, N), c.expect_mask("mid/ray_valid")), "ray_valid mask"
    app = bits_to_mask(ws.dbg_app, R_, N)
    exp_app = c.expect_mask("mid/app_mask")
    flips = int((app != exp_app).sum())
    assert flips == 0, f"app_mask flips={flips}, nearest-threshold margin of the fixture={float(c.expect('mid/app_margin')):.2e}"
    assert nvalid == int(c.expect("out/num_valid_samples"))
    # --- per-sample weights and colours of the shaded samples (ray-major order == reference order)
    w = packed_in_ray_order(ws, ws.app_w)
    np.testing.assert_allclose(w, c.expect("mid/weight")[exp_app], rtol=RTOL, atol=1e-9)
    if "mid/rgb_samples" in c.raw.files:
        s_rgb = packed_in_ray_order(ws, ws.rgb.view(-1, 3))
        np.testing.assert_allclose(s_rgb, c.expect("mid/rgb_samples"), rtol=RTOL, atol=ATOL_RGB)
    # --- outputs
    np.testing.assert_allclose(rgb.numpy(), c.expect("out/rgb_map"), rtol=RTOL, atol=ATOL_RGB)
    np.testing.assert_allclose(depth.numpy(), c.expect("out/depth_map"), rtol=RTOL, atol=1e-5)
    print(f"{name}: max|drgb|={np.abs(rgb.numpy() - c.expect('out/rgb_map')).max():.2e} "
          f"max|ddepth|={np.abs(depth.numpy() - c.expect('out/depth_map')).max():.2e}")


@pytest.mark.parametrize("name", ["vm_cubic_eval", "vm_noncubic_relu", "cp_eval", "vm_cubic_mask_vector"])
def test_feature_hooks(recon, name):
    """compute_densityfeature / compute_appfeature on a point list vs the oracle."""
    c = Case(name)
    model = build_model(recon, c, "cuda:0")
    g = torch.Generator().manual_seed(3)
    pts = torch.rand(1000, 3, generator=g) * 2 - 1
    pts[:8] = torch.tensor([[-1., -1, -1], [1, 1, 1], [1, -1, 1], [0, 0, 0], [-1, 1, 0.5], [0.25, 1, -1],
                            [1, 0, 0], [0, -1, 1]])   # exact lattice / boundary coordinates
    cfg = c.field_cfg()
    den_m = None if c.mask is None else c.mask["decomp"]["den"]
    app_m = None if c.mask is None else c.mask["decomp"]["app"]
    mk = c.mask_to("cuda:0")
    f = model.compute_densityfeature(pts.cuda(), None if mk is None else mk["decomp"]["den"]).cpu()
    a = model.compute_appfeature(pts.cuda(), None if mk is None else mk["decomp"]["app"]).cpu()
    f_ref = R.density_feature(cfg, c.state, pts, den_m)
    a_ref = R.app_feature(cfg, c.state, pts, app_m)
    np.testing.assert_allclose(f.numpy(), f_ref.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(a.numpy(), a_ref.numpy(), rtol=1e-5, atol=1e-5)


def test_renderer_matches_per_chunk_calls(recon):
    """OctreeRender_trilinear_fast: 6-tuple, chunking does not change results (renderer.py:13-26)."""
    c = Case("vm_cubic_eval")
    model = build_model(recon, c, "cuda:0")
    rays = c.rays
    with torch.no_grad():
        out = recon.OctreeRender_trilinear_fast(rays, model, chunk=64, N_samples=-1, white_bg=True, device="cuda:0")
        rgb1, depth1, n1 = model(rays.cuda(), None)
    assert out[1] is None and out[3] is None and out[4] is None and isinstance(out[5], float)
    assert torch.equal(out[0], rgb1) and torch.equal(out[2], depth1) and out[5] == float(n1)
    model.super_chunk = 64
    with torch.no_grad():
        out2 = recon.OctreeRender_trilinear_fast(rays, model, chunk=64, N_samples=-1, white_bg=True, device="cuda:0")
    assert torch.equal(out2[0], rgb1) and out2[5] == float(n1)
