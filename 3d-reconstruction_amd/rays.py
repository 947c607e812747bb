"""On-device ray generation: the camera rays of dataLoader/ray_utils.py without building the
(H, W, 3) direction image on the host and copying (R, 6) rays per chunk (renderer.py:18).

`generate_rays(H, W, focal, c2w, ...)` returns the (n, 6) fp32 rays `[o, d]` of the requested pixels on the GPU,
computed by `tf_generate_rays` with the arithmetic of `get_ray_directions[_blender]` + `get_rays`
(+ `ndc_rays_blender`).  Pinned by tests/golden/aux_refs.npz: rays the reference's own `get_rays` /
`ndc_rays_blender` produced (the module's `kornia` import is stubbed in the generator; neither function uses it)."""
import ctypes as C

import torch

from . import _hip as H
from .field import _stream


def generate_rays(H_img, W_img, focal, c2w, pixel_ids=None, first_pixel=0, n=None, center=None, normalize=True,
                  opengl=False, ndc_near=None, device="cuda"):
    """focal: float or (fx, fy); c2w: (3,4) or (4,4) camera-to-world; pixel_ids: int64 tensor of row-major pixel numbers
    (None: the `n` pixels from `first_pixel`, default the whole image); center: (cx, cy), default (W/2, H/2);
    normalize: unit-length camera directions (the Blender loader's convention); opengl: -y / -z camera axes
    (`get_ray_directions_blender`); ndc_near: project to NDC with this near plane (`ndc_rays_blender`)."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise H.HipError("generate_rays runs on the GPU only")
    cam = H.TfCamera()
    cam.height, cam.width = int(H_img), int(W_img)
    fx, fy = (focal, focal) if not hasattr(focal, "__len__") else focal
    cam.fx, cam.fy = float(fx), float(fy)
    cx, cy = (W_img / 2, H_img / 2) if center is None else center
    cam.cx, cam.cy = float(cx), float(cy)
    m = torch.as_tensor(c2w, dtype=torch.float32).cpu()[:3, :4].reshape(-1).tolist()
    for k in range(12):
        cam.c2w[k] = m[k]
    cam.opengl, cam.normalize = int(bool(opengl)), int(bool(normalize))
    cam.ndc, cam.ndc_near = int(ndc_near is not None), float(ndc_near or 0.0)
    ids = None
    if pixel_ids is not None:
        ids = pixel_ids.to(device=dev, dtype=torch.int64).contiguous()
        n = ids.numel()
    elif n is None:
        n = H_img * W_img - first_pixel
    out = torch.empty(n, 6, dtype=torch.float32, device=dev)
    H.check(H.lib().tf_generate_rays(C.byref(cam), ids.data_ptr() if ids is not None else None, int(first_pixel), int(n),
                                     out.data_ptr(), _stream()), "tf_generate_rays")
    return out
