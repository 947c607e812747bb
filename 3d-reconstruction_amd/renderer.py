"""Chunked ray-batch renderer with the reference's call signature (renderer.py:13-26)."""
import torch


class LazySampleCount:
    """Deferred `num_samples` (6th return value).  The reference converts the per-chunk count with
    `float(num_valid_samples)` (renderer.py:24), a device sync per chunk that stalls the launch pipeline; no
    caller in the reference uses the value for control flow (train.py:323 ignores it).  When the model sets
    `lazy_sample_count = True` the renderer returns this object instead: it behaves like the float in
    `float()`, arithmetic, comparisons and formatting, and syncs only when first read."""

    def __init__(self, tensors):
        self._t, self._v = tensors, None

    def __float__(self):
        if self._v is None:
            self._v = float(torch.stack([c.reshape(()) for c in self._t]).sum()) if self._t else 0.0
            self._t = None
        return self._v

    def __int__(self):
        return int(float(self))

    def __add__(self, o):
        return float(self) + float(o)

    __radd__ = __add__

    def __eq__(self, o):
        return float(self) == float(o)

    def __lt__(self, o):
        return float(self) < float(o)

    def __format__(self, spec):
        return format(float(self), spec)

    def __repr__(self):
        return repr(float(self))


def OctreeRender_trilinear_fast(rays, tensorf, mask=None, chunk=4096, N_samples=-1, ndc_ray=False, white_bg=True,
                                is_train=False, device='cuda'):
    """Returns `(rgbs, None, depth_maps, None, None, num_samples: float)` like renderer.py:26.

    Training keeps the reference's per-chunk call order exactly (one jitter draw and, without white
    background, one background draw per chunk).  In evaluation rays are independent and no random number
    is consumed, so several chunks are marched per kernel launch (`super_chunk` rays) and the per-chunk
    host sync `float(num_valid_samples)` (renderer.py:24) collapses into one sync at the end; results are
    identical to chunk-by-chunk calls."""
    n_all = rays.shape[0]
    step = chunk
    if not is_train:
        step = max(chunk, (getattr(tensorf, 'super_chunk', 32768) // chunk) * chunk)
    rgbs, depth_maps, counts = [], [], []
    for start in range(0, n_all, step):
        rays_chunk = rays[start:start + step].to(device)
        rgb_map, depth_map, num_valid_samples = tensorf(rays_chunk, mask, is_train=is_train, white_bg=white_bg,
                                                        ndc_ray=ndc_ray, N_samples=N_samples)
        rgbs.append(rgb_map)
        depth_maps.append(depth_map)
        counts.append(num_valid_samples)
    if not rgbs:
        return torch.empty(0, 3), None, torch.empty(0), None, None, 0.0
    total = LazySampleCount(counts)
    if not getattr(tensorf, 'lazy_sample_count', False):
        total = float(total)
    if len(rgbs) == 1:      # one chunk (the training call): hand the kernel outputs over without a copy
        return rgbs[0], None, depth_maps[0], None, None, total
    return torch.cat(rgbs), None, torch.cat(depth_maps), None, None, total
