"""Ray-sharded data parallelism: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The path shards by rays: every rank holds a replica of the field, renders a disjoint slice of
the global batch and, once per step, the parameter gradients are summed across ranks.  The HIP backward
already writes every gradient of a step into ONE contiguous fp32 buffer (`model.grad_flat`, 69.5 MB at
300^3), so the exchange is a single all-reduce on that buffer — no per-tensor bucketing, no copies.  On the
fully connected 8-GPU xGMI mesh one large all-reduce lets RCCL use all 7 links of every GPU at once.
The reference has no multi-GPU code; nothing here mirrors a reference call pattern."""
import math

import torch
import torch.distributed as dist


def shard_ids(ids: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Rank r takes ids[r::W] of the same seeded permutation, so the union over ranks equals the batch a
    single process would draw (train.py:44-56 SimpleSampler)."""
    return ids[rank::world]


def gradient_support(model):
    """Segments [(start, stop)] (floats) of `model.grad_flat` outside which every rank's data gradient is exactly
    zero, or None for "everything".

    The alpha mask is replicated and a sample only exists where the mask's trilinear look-up is positive, i.e. within
    one mask voxel of a non-zero voxel.  Along each axis that bounds the samples' coordinates, hence (bilinear taps:
    floor and floor + 1) the ROWS of every factor plane that can receive gradient.  Rows are contiguous in the
    channel-last planes, so the support is one row block per plane plus the (small) line tensors, basis and MLP.
    On the synthetic Lego scene (ball of radius 0.8 in a +-1.5 box) that is 53 % of the 69.5 MB.  The result is cached
    per (alpha mask, grid, buffer layout); computing it costs one `nonzero` over the mask volume."""
    layout = getattr(model, "grad_layout", None)
    mask = getattr(model, "alphaMask", None)
    if layout is None or mask is None or not hasattr(model, "density_plane"):
        return None
    offs, grad_len = layout
    geom = getattr(model, "_geom", None)          # host copy of the geometry (field.py); replaced when the grid changes
    cached = getattr(model, "_support_cache", None)
    if cached is not None and cached[2] is mask and cached[3] is offs and cached[4] is geom and geom is not None:
        return cached[1]                          # per-step path: no device access
    grid = [int(g) for g in model.gridSize.tolist()]
    key = (tuple(grid), grad_len)
    vol = mask.alpha_volume[0, 0]                                    # (Gz, Gy, Gx)
    nz = torch.nonzero(vol > 0)
    if nz.numel() == 0:
        return None
    lo_idx, hi_idx = nz.amin(0).tolist()[::-1], nz.amax(0).tolist()[::-1]      # per axis x, y, z
    m_lo, m_hi = mask.aabb[0].tolist(), mask.aabb[1].tolist()
    a_lo, a_hi = model.aabb[0].tolist(), model.aabb[1].tolist()
    mg = [int(g) for g in mask.gridSize.tolist()]
    rows = []
    for ax in range(3):
        cell = (m_hi[ax] - m_lo[ax]) / max(mg[ax] - 1, 1)
        w_lo = m_lo[ax] + (lo_idx[ax] - 1) * cell                    # one voxel of slack: cells next to a non-zero corner
        w_hi = m_lo[ax] + (hi_idx[ax] + 1) * cell
        x_lo = (w_lo - a_lo[ax]) / (a_hi[ax] - a_lo[ax]) * (grid[ax] - 1)
        x_hi = (w_hi - a_lo[ax]) / (a_hi[ax] - a_lo[ax]) * (grid[ax] - 1)
        r0 = max(0, int(math.floor(x_lo)) - 1)                       # floor tap, one row of rounding slack
        r1 = min(grid[ax], int(math.floor(x_hi)) + 3)                # floor + 1 tap, exclusive end, one row of slack
        rows.append((r0, max(r1, r0)))
    mat1 = (1, 2, 2)                                                 # row axis (H) of plane i (tensorBase.py:60)
    segs, covered = [], set()
    for kind, planes in (("density", model.density_plane), ("app", model.app_plane)):
        for i, p in enumerate(planes):
            name = f"{kind}_plane.{i}"
            _, c, h, w = p.shape
            r0, r1 = rows[mat1[i]]
            segs.append((offs[name] + r0 * w * c, offs[name] + r1 * w * c))
            covered.add(name)
    for name, o in offs.items():                                     # everything else in full
        if name not in covered:
            segs.append((o, o + dict(model.named_parameters())[name].numel()))
    segs = sorted(s for s in segs if s[1] > s[0])
    merged = []
    for a, b in segs:                                                # merge neighbours (offsets are 64-float aligned)
        if merged and a <= merged[-1][1] + 64:
            merged[-1] = (merged[-1][0], max(merged[-1][1], b))
        else:
            merged.append((a, b))
    total = sum(b - a for a, b in merged)
    result = merged if total < 0.9 * grad_len else None
    model._support_cache = (key, result, mask, offs, geom)
    return result


def allreduce_gradients(model, group=None, average=True, use_support=True):
    """Sums (averages) the step's gradients across ranks in one collective.  Falls back to a flattened copy
    when the gradients do not come from the HIP backward's contiguous buffer (e.g. CPU tests).  With an alpha mask
    only the part of the buffer that can be non-zero is exchanged (`gradient_support`)."""
    if not dist.is_available() or not dist.is_initialized():
        return
    world = dist.get_world_size(group)
    if world == 1:
        return
    flat = getattr(model, "grad_flat", None)
    params = [p for p in model.parameters() if p.grad is not None]
    owned = flat is not None and all(p.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr()
                                     for p in params)
    if owned:
        segs = gradient_support(model) if use_support else None
        if segs is None:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            if average:
                flat.mul_(1.0 / world)
            return
        pieces = [flat[a:b] for a, b in segs]
        buf = torch.cat(pieces)                                      # one gather launch
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        if average:
            buf.mul_(1.0 / world)
        torch._foreach_copy_(pieces, list(buf.split([b - a for a, b in segs])))
        return
    buf = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    if average:
        buf.mul_(1.0 / world)
    off = 0
    for p in params:
        n = p.grad.numel()
        p.grad.copy_(buf[off:off + n].view(p.grad.shape))   # logical order; copy_ handles channel-last strides
        off += n


def allreduce_scalar(value: torch.Tensor, group=None, average=True):
    """Loss / PSNR logging across ranks."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return value
    v = value.detach().clone()
    dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
    return v / dist.get_world_size(group) if average else v
