"""Synthetic workload of SURVEY §8(d): Blender-Lego camera geometry and a 'trained-like' field state.

No dataset exists in the container or on the GPU box, so the benchmark and the harness use rays generated
from the Blender camera model the reference's loader implements (dataLoader/blender.py:30-60,
dataLoader/ray_utils.py:24-42, 66-87): H = W = 800, camera_angle_x = 0.6911112070083618, cameras on the
radius-4.0311 upper hemisphere looking at the origin, unit-norm directions, aabb = +-1.5, near_far = [2, 6],
white background."""
import math

import numpy as np
import torch

LEGO_AABB = [[-1.5, -1.5, -1.5], [1.5, 1.5, 1.5]]
LEGO_NEAR_FAR = [2.0, 6.0]


def lego_args(shadingMode="MLP_Fea", density_n_comp=(16, 16, 16), app_n_comp=(48, 48, 48)):
    """The 13 constructor keys of train.py:228-242 with configs/config.yaml values (MLP_Fea per north_star)."""
    return dict(step_ratio=0.5, fea2denseAct="softplus", density_n_comp=list(density_n_comp),
                app_n_comp=list(app_n_comp), app_dim=27, density_shift=-10.0, distance_scale=25.0,
                alphaMask_thres=0.001, shadingMode=shadingMode, pos_pe=2, view_pe=2, fea_pe=2, featureC=128)


def blender_rays(n_views=1, H=800, W=800, seed=20211202, radius=4.0311, angle_x=0.6911112070083618):
    """(n_views*H*W, 6) fp32 rays [o, d], one camera per view drawn on the upper hemisphere."""
    rng = np.random.default_rng(seed)
    focal = 0.5 * W / math.tan(0.5 * angle_x)
    j, i = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    dirs = torch.stack([(i + 0.5 - W / 2) / focal, (j + 0.5 - H / 2) / focal, torch.ones_like(i)], -1)
    dirs = dirs / torch.norm(dirs, dim=-1, keepdim=True)
    out = []
    for _ in range(n_views):
        th = rng.uniform(0, 2 * math.pi)
        ph = rng.uniform(math.radians(10), math.radians(80))
        cam = np.array([math.cos(th) * math.cos(ph), math.sin(th) * math.cos(ph), math.sin(ph)]) * radius
        fwd = -cam / np.linalg.norm(cam)
        right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        c2w = torch.tensor(np.stack([right, down, fwd, cam], axis=1), dtype=torch.float32)  # OpenCV-style
        d = dirs.view(-1, 3) @ c2w[:, :3].T
        out.append(torch.cat([c2w[:, 3].expand_as(d), d], 1))
    return torch.cat(out, 0)


def ball_alpha_volume(res=128, aabb=LEGO_AABB, radius=0.8):
    """(res,res,res) 0/1 indicator of the ball |p| < radius on the aabb lattice, layout [z][y][x]."""
    lo, hi = torch.tensor(aabb[0]), torch.tensor(aabb[1])
    lin = [torch.linspace(float(lo[k]), float(hi[k]), res) for k in range(3)]
    zz, yy, xx = torch.meshgrid(lin[2], lin[1], lin[0], indexing="ij")
    return ((xx ** 2 + yy ** 2 + zz ** 2) < radius ** 2).float()


@torch.no_grad()
def make_trained_like(model, alpha_mask_cls, mask_res=128, radius=0.8):
    """Boost one density component and install a ball alpha mask: a fresh init has zero shaded samples
    (density_shift = -10), so the shading stages would never run (SURVEY §9 'Init state')."""
    if hasattr(model, "density_plane"):
        model.density_plane[0][:, 0] = 10.0
        model.density_line[0][:, 0] = 1.0
    else:
        for i in range(3):
            model.density_line[i][:, 0] = 2.2
    vol = ball_alpha_volume(mask_res, model.aabb.tolist(), radius).to(model.aabb.device)
    model.alphaMask = alpha_mask_cls(model.device, model.aabb, vol)
    return model


def bbox_hit_mask(rays, aabb):
    """filtering_rays(bbox_only=True) predicate (models/tensorBase.py:271-277)."""
    o, d = rays[:, :3], rays[:, 3:6]
    vec = torch.where(d == 0, torch.full_like(d, 1e-6), d)
    ra, rb = (aabb[1] - o) / vec, (aabb[0] - o) / vec
    return torch.maximum(ra, rb).amin(-1) > torch.minimum(ra, rb).amax(-1)


# ---- the other BASELINE configurations (C4 forward-facing NDC, C5 Tanks&Temples-like) ----------
LLFF_AABB = [[-1.5, -1.67, -1.0], [1.5, 1.67, 1.0]]      # dataLoader/llff.py:143
LLFF_NEAR_FAR = [0.0, 1.0]                                # dataLoader/llff.py:142
TT_NEAR_FAR = [0.01, 6.0]                                 # dataLoader/tankstemple.py:97


def ndc_project(H, W, focal, near, rays_o, rays_d):
    """Camera-space rays of a forward-facing scene -> NDC rays (the projection dataLoader/ray_utils.py:90-107
    applies): origins are moved onto the near plane, then x, y are divided by depth and scaled by the image
    half-extent over the focal length; z maps [near, inf) to [-1, 1)."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    o = rays_o + t[..., None] * rays_d
    sx, sy = -2.0 * focal / W, -2.0 * focal / H
    ox, oy, oz = o[..., 0] / o[..., 2], o[..., 1] / o[..., 2], o[..., 2]
    dx, dy = rays_d[..., 0] / rays_d[..., 2], rays_d[..., 1] / rays_d[..., 2]
    new_o = torch.stack([sx * ox, sy * oy, 1.0 + 2.0 * near / oz], -1)
    new_d = torch.stack([sx * (dx - ox), sy * (dy - oy), -2.0 * near / oz], -1)
    return new_o, new_d


def llff_ndc_rays(n_rays, H=756, W=1008, focal=815.0, seed=7):
    """`n_rays` NDC rays of a forward-facing capture: a 1008x756 pinhole looking down -z (OpenGL convention, as the
    LLFF poses are) from small random offsets around the origin, projected with ndc_project(near=1)."""
    g = torch.Generator().manual_seed(seed)
    px = torch.rand(n_rays, generator=g) * W
    py = torch.rand(n_rays, generator=g) * H
    dirs = torch.stack([(px - W / 2) / focal, -(py - H / 2) / focal, -torch.ones(n_rays)], -1)
    origins = (torch.rand(n_rays, 3, generator=g) - 0.5) * torch.tensor([0.6, 0.4, 0.1])
    o, d = ndc_project(H, W, focal, 1.0, origins, dirs)
    return torch.cat([o, d], 1).float()


def tt_rays(n_rays, aabb, seed=11):
    """`n_rays` rays of an inward-facing capture whose cameras stand INSIDE the (enlarged) scene box, as in Tanks&Temples
    (near = 0.01): origins on a ring at ~60 % of the box half-extent, unit directions towards jittered points near the
    centre; one in 64 directions has an exactly zero component (the reference's `where(d == 0, 1e-6, d)` branch)."""
    g = torch.Generator().manual_seed(seed)
    lo, hi = torch.tensor(aabb[0]), torch.tensor(aabb[1])
    ctr, half = (lo + hi) / 2, (hi - lo) / 2
    th = torch.rand(n_rays, generator=g) * 2 * math.pi
    ring = torch.stack([torch.cos(th), torch.sin(th), 0.3 * (torch.rand(n_rays, generator=g) - 0.5)], -1)
    o = ctr + 0.6 * half * ring
    target = ctr + 0.25 * half * (torch.rand(n_rays, 3, generator=g) - 0.5)
    d = target - o
    d = d / d.norm(dim=-1, keepdim=True)
    z = torch.arange(n_rays) % 64 == 0
    d[z, 2] = 0.0
    return torch.cat([o, d], 1).float()


# ---- BASELINE.json's five configurations as seeded synthetic scenes (bench.py --config, tests/test_full_size.py)
TT_AABB = [[-2.4, -1.6, -1.9], [2.2, 1.7, 1.3]]
BASELINE_SCENES = ("C1_vm128", "C2_vm300", "C3_cp300_sh", "C3_cp300_mlp", "C4_ndc", "C5_tt640")


def baseline_scene(name, device, n_rays=None, views=1):
    """(model, rays (on the CPU), n_samples, ndc_ray, white_bg) of one BASELINE configuration at its full size, in the
    'trained-like' state:
      C1_vm128      configs/lego.txt, TensorVMSplit 128^3, N = 443 (the reference's CPU-runnable case)
      C2_vm300      TensorVMSplit 300^3, MLP_Fea, N = 1039 (the headline)
      C3_cp300_*    TensorCP [96] / [288] at 300^3 with the SH head (inference only, as in the reference) / MLP_Fea
      C4_ndc        forward-facing NDC rays (llff.py:141-143: no white background), VM [16,4,4] / [48,12,12]
      C5_tt640      Tanks&Temples-like: cameras inside a non-cubic box, near = 0.01, 640^3-equivalent grid
    `views` Blender views (C1-C3) or `n_rays` generated rays (C4, C5)."""
    from .field import AlphaGridMask, TensorCP, TensorVMSplit
    from .utils import N_to_reso, cal_n_samples
    torch.manual_seed(0)
    ndc, white = False, True
    if name in ("C1_vm128", "C2_vm300"):
        g = 128 if name == "C1_vm128" else 300
        aabb = torch.tensor(LEGO_AABB, device=device)
        model = TensorVMSplit(lego_args(), aabb, N_to_reso(g ** 3, aabb), LEGO_NEAR_FAR, device)
        rays = blender_rays(views)
    elif name.startswith("C3_cp300"):
        aabb = torch.tensor(LEGO_AABB, device=device)
        head = "SH" if name.endswith("sh") else "MLP_Fea"
        # BASELINE config 3 ([96]/[288], configs/lego.txt:80-83): with the SH head inference only (the reference
        # cannot train that head either); with MLP_Fea also trained — at 288 components the backward's tile does not
        # fit LDS in one piece, which exercises its gather-V-twice layout
        args = lego_args(head, density_n_comp=(96,), app_n_comp=(288,))
        model = TensorCP(args, aabb, N_to_reso(300 ** 3, aabb), near_far=LEGO_NEAR_FAR, device=device)
        rays = blender_rays(views)
    elif name == "C4_ndc":
        aabb = torch.tensor(LLFF_AABB, device=device)
        args = lego_args(density_n_comp=(16, 4, 4), app_n_comp=(48, 12, 12))
        model = TensorVMSplit(args, aabb, N_to_reso(300 ** 3, aabb), LLFF_NEAR_FAR, device)
        rays, ndc, white = llff_ndc_rays(n_rays or (1 << 16)), True, False
    elif name == "C5_tt640":
        aabb = torch.tensor(TT_AABB, device=device)
        model = TensorVMSplit(lego_args(), aabb, N_to_reso(640 ** 3, aabb), TT_NEAR_FAR, device)
        rays = tt_rays(n_rays or (1 << 15), TT_AABB)
    else:
        raise ValueError(f"unknown scene {name!r}: one of {BASELINE_SCENES}")
    make_trained_like(model, AlphaGridMask, radius=0.8 if name != "C4_ndc" else 0.9)
    n_samples = min(int(1e6), cal_n_samples(model.gridSize.tolist(), 0.5))
    return model, rays, n_samples, ndc, white
