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
