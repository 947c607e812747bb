"""Chunked ray-batch renderer with the reference's call signature (renderer.py:13-26)."""
import torch


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
    total = float(torch.stack([c.reshape(()) for c in counts]).sum())
    return torch.cat(rgbs), None, torch.cat(depth_maps), None, None, total
