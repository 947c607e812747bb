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
    # The fixtures come from the reference on the CPU.  update_stepSize (tensorBase.py:104-116) is the same torch
    # expression here, but torch.mean over the three axis units rounds differently on the GPU (1 ulp on cubic grids:
    # sum * (1/3) against sum / 3) — the reference itself would step differently on the two devices.  For the bit-exact
    # z comparison the kernels get the step the fixture's run used.
    step_cpu = float(c.cfg_d["stepSize"])
    if abs(float(model.stepSize) - step_cpu) <= 2e-7 * step_cpu:
        model.stepSize = torch.tensor(step_cpu, dtype=torch.float32, device=device)
        model._geom = None
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
    # --- sample positions along the rays: bit-exact (tensorBase.py:198-203, :181-183)
    z = ws.dbg_z.view(R_, N).cpu().numpy()
    assert np.array_equal(z.view(np.uint32), c.expect("mid/z").astype(np.float32).view(np.uint32)), \
        f"z differs in {int((z != c.expect('mid/z')).sum())} of {z.size} samples"
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


@pytest.mark.parametrize("name", ["vm_cubic_eval", "vm_head_MLP", "vm_head_MLP_PE", "vm_cubic_mask_vector", "vm_ndc_eval"])
def test_render_module_alone(recon, name):
    """renderModule(pts, viewdirs, features, mask) as a stand-alone call (mlp.py:41-69, 84-107, 126-155): the fixture holds
    the reference head's output on the shaded samples of the case (gen_golden.py run_case: `mid/rgb_samples` =
    model.renderModule(xyz_n[app], viewdirs[app], app_features, mask=enc_mask))."""
    c = Case(name)
    if "mid/rgb_samples" not in c.raw.files:
        pytest.skip("no shaded samples in this fixture")
    dev = "cuda:0"
    model = build_model(recon, c, dev)
    R_, N = c.shape
    app = torch.from_numpy(c.expect_mask("mid/app_mask"))
    rays = c.rays
    z = torch.from_numpy(c.expect("mid/z").astype(np.float32))
    pts = rays[:, None, :3] + rays[:, None, 3:6] * z[..., None]                    # tensorBase.py:205 / :185
    aabb = torch.tensor(c.cfg_d["aabb"])
    xyz_n = (pts - aabb[0]) * (2.0 / (aabb[1] - aabb[0])) - 1                       # tensorBase.py:130-131
    vd = rays[:, 3:6]
    if c.call["ndc_ray"]:
        vd = vd / torch.norm(vd, dim=-1, keepdim=True)
    vd = vd.view(-1, 1, 3).expand(pts.shape)
    feats = torch.from_numpy(c.expect("mid/app_features"))
    mk = c.mask_to(dev)
    enc = None if mk is None else mk["encoding"]
    out = model.renderModule(xyz_n[app].to(dev), vd[app].to(dev), feats.to(dev), enc).cpu().numpy()
    np.testing.assert_allclose(out, c.expect("mid/rgb_samples"), rtol=RTOL, atol=ATOL_RGB)


@pytest.mark.parametrize("name", ["vm_cubic_eval", "vm_cubic_train", "vm_tnt_inside", "cp_eval"])
def test_early_ray_termination_is_parity_safe(recon, name):
    """north_star: wavefront early termination.  Samples behind T < t_stop are not evaluated; they are unshaded (w <= T <
    1e-7 << rayMarch_weight_thres) but would still add w to acc_map / depth_map (tensorBase.py:377,387), so the cut is only
    parity-safe far below fp32 resolution of those sums: at t_stop = 1e-7 every output stays inside the 1e-4
    bar, the shaded set is unchanged, and (on the opaque fixtures) fewer density samples are evaluated."""
    c = Case(name)
    dev = "cuda:0"
    outs = {}
    for t_stop in (0.0, 1e-7):
        model = build_model(recon, c, dev)
        model.t_stop = t_stop
        call = c.call
        torch.manual_seed(call["seed"])
        with torch.no_grad():
            rgb, depth, nvalid = model(c.rays.to(dev), c.mask_to(dev), white_bg=call["white_bg"], is_train=call["is_train"],
                                       ndc_ray=call["ndc_ray"], N_samples=call["N_samples"])
        evaluated = int(model.last["ws"].counters2d[:, 1].sum())
        outs[t_stop] = (rgb.cpu().numpy(), depth.cpu().numpy(), int(nvalid), evaluated)
    full, cut = outs[0.0], outs[1e-7]
    assert cut[2] == full[2] == int(c.expect("out/num_valid_samples"))
    np.testing.assert_allclose(cut[0], c.expect("out/rgb_map"), rtol=RTOL, atol=ATOL_RGB)
    np.testing.assert_allclose(cut[1], c.expect("out/depth_map"), rtol=RTOL, atol=1e-5)
    assert cut[3] <= full[3]
    print(f"{name}: density samples evaluated {full[3]} -> {cut[3]} with t_stop = 1e-7")


def test_early_ray_termination_skips_work_on_an_opaque_scene(recon):
    """On a dense synthetic field the cut must actually bite (and still render the same image)."""
    from recon_amd import synthetic as S
    dev = "cuda:0"
    torch.manual_seed(0)
    aabb = torch.tensor(S.LEGO_AABB, device=dev)
    model = recon.TensorVMSplit(S.lego_args(), aabb, [64] * 3, S.LEGO_NEAR_FAR, dev)
    S.make_trained_like(model, recon.AlphaGridMask, mask_res=48)
    with torch.no_grad():
        model.density_plane[0][:, 0] = 14.0                 # opaque ball
    allr = S.blender_rays(1)
    dist = torch.linalg.cross(allr[:, :3], allr[:, 3:6]).norm(dim=-1) / allr[:, 3:6].norm(dim=-1)   # ray - origin distance
    rays = allr[dist < 0.5][::7][:4096].to(dev).contiguous()       # rays through the ball
    res = {}
    for t_stop in (0.0, 1e-7):
        model.t_stop = t_stop
        with torch.no_grad():
            rgb, depth, n = model(rays, None, white_bg=True, is_train=False, N_samples=-1)
        res[t_stop] = (rgb.cpu(), depth.cpu(), int(n), int(model.last["ws"].counters2d[:, 1].sum()))
    assert res[1e-7][2] == res[0.0][2]
    assert res[1e-7][3] < res[0.0][3], (res[1e-7][3], res[0.0][3])      # (the cut acts per block of 64 density samples)
    assert (res[1e-7][0] - res[0.0][0]).abs().max().item() < 1e-5
    assert ((res[1e-7][1] - res[0.0][1]).abs() / res[0.0][1].abs().clamp_min(1e-3)).max().item() < 1e-4


@pytest.mark.gpu
def test_cached_launch_descriptors_are_reused_and_follow_the_parameters(recon):
    """The forward's TfField / TfShade / pack job are built once (TensorBase._plan) and reused while nothing they hold by
    value changed: a second call must not rebuild them, and a parameter that is re-homed (`p.data = ...`),
    updated in place, or a changed scalar setting must be seen by the next call."""
    c = Case("vm_head_MLP")
    model = build_model(recon, c, "cuda:0")
    rays = c.rays.cuda()
    with torch.no_grad():
        rgb0, depth0, _ = model(rays, None)
        plan = model._plans[False]
        rgb1, _, _ = model(rays, None)
        assert model._plans[False] is plan and torch.equal(rgb0, rgb1)
        # in-place update of a shading weight (what an optimiser step does): same plan, new packed copy
        w = model.renderModule.mlp[0].weight
        w.mul_(0.5)
        twin = build_model(recon, c, "cuda:0")
        twin.renderModule.mlp[0].weight.mul_(0.5)
        want, _, _ = twin(rays, None)
        got, _, _ = model(rays, None)
        assert model._plans[False] is plan and torch.equal(got, want)
        # re-homed factor: same Parameter object, new storage with new values
        p = model.density_plane[0]
        p.data = (p.data * 1.25).clone()
        twin.density_plane[0].mul_(1.25)
        want, want_d, _ = twin(rays, None)
        got, got_d, _ = model(rays, None)
        assert model._plans[False] is not plan
        assert torch.equal(got, want) and torch.equal(got_d, want_d)
        # a scalar the descriptors hold by value
        plan = model._plans[False]
        model.distance_scale = twin.distance_scale = 10.0
        want, _, _ = twin(rays, None)
        got, _, _ = model(rays, None)
        assert model._plans[False] is not plan and torch.equal(got, want)


def test_sh_and_rgb_heads_alone_against_the_reference(recon):
    """SHRender / RGBRender as stand-alone calls (models/mlp.py:15-25) through tf_shade_points, directly against the
    reference functions' outputs (tests/golden/sh_head.npz: 300 feature rows and view directions)."""
    from recon_amd import synthetic as S
    from tests._golden import _npz
    z = _npz("sh_head")
    dev = "cuda:0"
    feats, dirs = torch.from_numpy(z["feats"]).to(dev), torch.from_numpy(z["dirs"]).to(dev)
    pts = torch.zeros_like(dirs)
    aabb = torch.tensor(S.LEGO_AABB, device=dev)
    sh = recon.TensorVMSplit(S.lego_args("SH", density_n_comp=(4, 4, 4), app_n_comp=(4, 4, 4)), aabb, [8, 8, 8],
                             S.LEGO_NEAR_FAR, dev)
    assert sh.renderModule == "SH"
    np.testing.assert_allclose(sh.renderModule(pts, dirs, feats).cpu().numpy(), z["rgb_sh"], rtol=RTOL, atol=ATOL_RGB)
    a = S.lego_args("RGB", density_n_comp=(4, 4, 4), app_n_comp=(4, 4, 4))
    a["app_dim"] = 3
    rgb = recon.TensorVMSplit(a, aabb, [8, 8, 8], S.LEGO_NEAR_FAR, dev)
    np.testing.assert_allclose(rgb.renderModule(pts, dirs, feats[:, :3].contiguous()).cpu().numpy(), z["rgb_passthrough"],
                               rtol=RTOL, atol=ATOL_RGB)


@pytest.mark.parametrize("scene", ["lego_like", "ndc", "tiny_lists"])
def test_pipelined_and_classic_forward_kernels_agree(recon, scene):
    """tf_shade_forward has two kernels (csrc/shade.hip): the pipelined one-workgroup-per-CU kernel — hidden layers on
    the bf16 pipe with three-piece operands — wherever its shape conditions hold, and the fp32 two-workgroups-per-CU
    kernel elsewhere (tf_shade_forward_variant switches).  Same batch, same jitter, training mode (every row the
    backward reads is written): sample colours, rendered colours and the saved V / X / H1 / H2 rows of both within
    fp32 rounding of each other, and the SAME ReLU pattern — a unit may differ only where its activation is a tie
    (|h| <= 1e-6 in the kernel that has it on; the reference's own GPU run moves such units too)."""
    from recon_amd import _hip as H
    from recon_amd import synthetic as S
    dev = "cuda:0"
    lib = H.lib()
    torch.manual_seed(3)
    ndc, white = False, True
    if scene == "ndc":
        aabb = torch.tensor(S.LLFF_AABB, device=dev)
        model = recon.TensorVMSplit(S.lego_args(density_n_comp=(16, 4, 4), app_n_comp=(48, 12, 12)), aabb, [80, 88, 56], S.LLFF_NEAR_FAR, dev)
        rays, ndc, white, N = S.llff_ndc_rays(3000).to(dev), True, False, 96
        S.make_trained_like(model, recon.AlphaGridMask, mask_res=48, radius=0.9)
    else:
        aabb = torch.tensor(S.LEGO_AABB, device=dev)
        model = recon.TensorVMSplit(S.lego_args(), aabb, [96, 96, 96], S.LEGO_NEAR_FAR, dev)
        S.make_trained_like(model, recon.AlphaGridMask, mask_res=64, radius=0.8)
        n_rays = 4096 if scene == "lego_like" else 150      # tiny_lists: a few samples per shard — chunks span several shards
        allrays = S.blender_rays(1)
        rays = allrays[torch.randperm(allrays.shape[0], generator=torch.Generator().manual_seed(5))[:n_rays]].to(dev).contiguous()
        N = 333
    jit = torch.rand(1 if ndc else rays.shape[0], N if ndc else 1, generator=torch.Generator().manual_seed(7))
    out = {}
    try:
        for variant in (1, 0):
            assert lib.tf_shade_forward_variant(variant) == 0
            model._jitter_override = jit.clone()
            model._bg_override = False
            model.zero_grad(set_to_none=True)
            rgb, _, nv = model(rays, None, white_bg=white, is_train=True, ndc_ray=ndc, N_samples=N)
            torch.cuda.synchronize()
            ws = model.last["ws"]
            n_app = int(nv)
            fc, kp = 128, (int(model.last["shade"].in_c) + 15) // 16 * 16
            order = lambda t, w: packed_rows(ws, t, w)
            out[variant] = dict(rgb_map=rgb.detach().cpu().numpy(), n=n_app,
                                rgb=order(ws.rgb, 3), v=order(ws.dv, int(model.last["shade"].n_app_total)),
                                x=order(ws.xs, kp), h1=order(ws.h1s, fc), h2=order(ws.h2s, fc))
            del rgb
    finally:
        lib.tf_shade_forward_variant(0)
        model._bg_override = None
    a, b = out[1], out[0]          # classic, pipelined
    assert a["n"] == b["n"] and a["n"] > (50 if scene == "tiny_lists" else 2000)
    assert np.array_equal(a["v"], b["v"])                                    # the same gather arithmetic
    np.testing.assert_allclose(b["x"], a["x"], rtol=0, atol=2e-6)            # basis product: other summation order
    for k in ("h1", "h2"):
        np.testing.assert_allclose(b[k], a[k], rtol=0, atol=2e-5 * max(1.0, float(np.abs(a[k]).max())))
        flips = (a[k] > 0) != (b[k] > 0)
        assert np.maximum(np.abs(a[k]), np.abs(b[k]))[flips].max(initial=0.0) <= 1e-6, (k, int(flips.sum()))
    np.testing.assert_allclose(b["rgb"], a["rgb"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(b["rgb_map"], a["rgb_map"], rtol=0, atol=2e-6)
    print({k: float(np.abs(a[k] - b[k]).max()) for k in ("x", "h1", "h2", "rgb", "rgb_map")},
          {k: int(((a[k] > 0) != (b[k] > 0)).sum()) for k in ("h1", "h2")}, "units:", a["h1"].size)


def packed_rows(ws, flat, width):
    """rows (width floats each) of a packed per-sample buffer in ray-major order"""
    off, cnt = ws.app_offset.cpu().numpy(), ws.app_count.cpu().numpy()
    idx = np.concatenate([np.arange(o, o + k) for o, k in zip(off, cnt)]) if cnt.sum() else np.zeros(0, np.int64)
    return flat.detach().view(-1, width).cpu().numpy()[idx]
