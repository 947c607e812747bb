"""Ray-sharded data parallelism: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The path shards by rays: every rank holds a replica of the field, renders a disjoint slice of
the global batch and, once per step, the parameter gradients are summed across ranks.  The HIP backward
already writes every gradient of a step into ONE contiguous fp32 buffer (`model.grad_flat`, 69.5 MB at
300^3), so the exchange is a single all-reduce on that buffer — no per-tensor bucketing, no copies.  On the
fully connected 8-GPU xGMI mesh one large all-reduce lets RCCL use all 7 links of every GPU at once.
The reference has no multi-GPU code; nothing here mirrors a reference call pattern."""
import math
import os

import torch
import torch.distributed as dist


# TF_DP_FORCE_EXCHANGE=1: run the gradient exchange (and the split-graph step built around it) even in a one-rank
# group — a rehearsal of the N > 1 code path, RCCL included, on a single GPU
FORCE_EXCHANGE = bool(int(os.environ.get("TF_DP_FORCE_EXCHANGE", "0")))


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


def gradient_support_rows(model):
    """(w, idx) — `model.grad_flat.view(-1, w)[idx]` are the only rows of the gradient buffer that can be non-zero on
    any rank — or None.  The cell-level refinement of `gradient_support`:

    a sample exists only where the (replicated) alpha mask's trilinear look-up is positive, i.e. within one mask voxel
    of a non-zero voxel in all three axes; projected along the axis a factor plane does not span, a plane cell (r, c)
    can receive gradient only if the projected occupancy has a non-zero voxel within that distance of the grid
    interval (c - 1, c + 1) x (r - 1, r + 1) its bilinear taps cover.  The test is one rectangle query per cell on
    the integral image of the projection (torch ops, once per alpha-mask update).  Cells are contiguous C-float rows
    of the channel-last planes, so the support is a row list: the exchange gathers those rows, all-reduces them and
    writes them back (`allreduce_gradients`).  On the synthetic Lego ball this is ~25 % of the buffer (the row blocks
    of `gradient_support`: 53 %); after `shrink` the row blocks cover everything, the silhouettes still do not."""
    layout = getattr(model, "grad_layout", None)
    mask = getattr(model, "alphaMask", None)
    if layout is None or mask is None or not hasattr(model, "density_plane") or len(model.density_plane) != 3:
        return None
    offs, grad_len = layout
    geom = getattr(model, "_geom", None)
    cached = getattr(model, "_rows_cache", None)
    if cached is not None and cached[1] is mask and cached[2] is offs and cached[3] is geom and geom is not None:
        return cached[0]
    result = None
    comps = [int(p.shape[1]) for p in list(model.density_plane) + list(model.app_plane)]
    w = math.gcd(16, *comps)
    vol = mask.alpha_volume[0, 0] > 0                                # (Gz, Gy, Gx)
    if w >= 4 and grad_len % w == 0 and bool(vol.any()):
        dev = vol.device
        occ3 = vol.permute(2, 1, 0)                                  # indexed [x, y, z]
        grid = [int(g) for g in model.gridSize.tolist()]
        mg = [int(g) for g in mask.gridSize.tolist()]
        m_lo, m_hi = mask.aabb[0].tolist(), mask.aabb[1].tolist()
        a_lo, a_hi = model.aabb[0].tolist(), model.aabb[1].tolist()

        def voxel_range(ax):
            """per grid index g of axis ax: [lo, hi] mask voxels that can put a sample into a cell whose taps are g"""
            g = torch.arange(grid[ax], dtype=torch.float64)
            span = (a_hi[ax] - a_lo[ax]) / max(grid[ax] - 1, 1)
            cell = (m_hi[ax] - m_lo[ax]) / max(mg[ax] - 1, 1)
            # samples with grid coordinate in (g - 1, g + 1) touch row g; one more row of slack for rounding
            m0 = (a_lo[ax] + (g - 2) * span - m_lo[ax]) / cell - 1.0
            m1 = (a_lo[ax] + (g + 2) * span - m_lo[ax]) / cell + 1.0
            lo = (torch.floor(m0) + 1).clamp(0, mg[ax]).long()       # voxels v with m0 < v < m1; lo == mg: empty
            hi = torch.ceil(m1).clamp(0, mg[ax]).long()              # exclusive
            return lo.to(dev), torch.maximum(hi.to(dev), lo.to(dev))

        ranges = [voxel_range(ax) for ax in range(3)]
        mat0, mat1 = (0, 0, 1), (1, 2, 2)                            # W and H axis of plane i (tensorBase.py:60)
        pieces = []
        covered = set()
        for i in range(3):
            aw, ah = mat0[i], mat1[i]
            occ = occ3.any(dim=3 - aw - ah).to(torch.int32)          # [aw index, ah index]  (aw < ah)
            integ = torch.zeros(occ.shape[0] + 1, occ.shape[1] + 1, dtype=torch.int32, device=dev)
            integ[1:, 1:] = occ.cumsum(0).cumsum(1)
            (lw, hw), (lh, hh) = ranges[aw], ranges[ah]
            cnt = integ[hw[None, :], hh[:, None]] - integ[lw[None, :], hh[:, None]] \
                - integ[hw[None, :], lh[:, None]] + integ[lw[None, :], lh[:, None]]          # (H, W)
            cells = torch.nonzero((cnt > 0).reshape(-1)).view(-1)                            # r * W + c
            for kind, planes in (("density", model.density_plane), ("app", model.app_plane)):
                name = f"{kind}_plane.{i}"
                _, c, h, wd = planes[i].shape
                assert (h, wd) == (grid[ah], grid[aw]) and offs[name] % w == 0
                k = c // w
                rows = offs[name] // w + cells[:, None] * k + torch.arange(k, device=dev)[None, :]
                pieces.append(rows.reshape(-1))
                covered.add(name)
        named = dict(model.named_parameters())
        for name, o in offs.items():                                 # everything else in full (offsets are 64-aligned)
            if name not in covered:
                pieces.append(torch.arange(o // w, (o + named[name].numel() + w - 1) // w, device=dev))
        idx = torch.sort(torch.cat(pieces))[0]
        if idx.numel() * w < 0.9 * grad_len:
            result = (w, idx)
    model._rows_cache = (result, mask, offs, geom, None if result is None else result[1].to(torch.int32), None)
    return result


def allreduce_gradients(model, group=None, average=True, use_support=True):
    """Sums (averages) the step's gradients across ranks in one collective.  Falls back to a flattened copy
    when the gradients do not come from the HIP backward's contiguous buffer (e.g. CPU tests).  With an alpha mask
    only the part of the buffer that can be non-zero is exchanged: the touched cells of the factor planes
    (`gradient_support_rows`) or, when that list is not available (or `use_support="blocks"`), their row blocks
    (`gradient_support`); `use_support=False` exchanges the whole buffer."""
    if not dist.is_available() or not dist.is_initialized():
        return
    world = dist.get_world_size(group)
    if world == 1 and not FORCE_EXCHANGE:
        return
    _allreduce_live(model, group)
    flat = getattr(model, "grad_flat", None)
    params = [p for p in model.parameters() if p.grad is not None]
    owned = flat is not None and all(p.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr()
                                     for p in params)
    if owned:
        rows = gradient_support_rows(model) if use_support and use_support != "blocks" else None
        if rows is not None:
            w, idx = rows
            table = flat.view(-1, w)
            if flat.is_cuda:                                             # tf_gather_rows / tf_scatter_rows: float4 moves
                from . import _hip as H
                from .field import _stream
                idx32 = model._rows_cache[4]
                buf = torch.empty(idx.numel(), w, dtype=torch.float32, device=flat.device)
                H.check(H.lib().tf_gather_rows(flat.data_ptr(), idx32.data_ptr(), idx.numel(), w, buf.data_ptr(),
                                               _stream()), "tf_gather_rows")
            else:
                buf = table.index_select(0, idx)
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
            if average:
                buf.mul_(1.0 / world)
            if flat.is_cuda:
                H.check(H.lib().tf_scatter_rows(flat.data_ptr(), idx32.data_ptr(), idx.numel(), w, buf.data_ptr(),
                                                _stream()), "tf_scatter_rows")
            else:
                table.index_copy_(0, idx, buf)
            return
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


def _allreduce_live(model, group=None):
    """The step's sample counts (field.TensorBase._live: density / shaded samples, the optimizer's gates) summed like the
    gradients: a tensor group is updated when ANY rank had samples for it — what a single process with the whole batch
    would do — and every rank takes the same decision."""
    live = getattr(model, "_live", None)
    if live is not None:
        dist.all_reduce(live, op=dist.ReduceOp.SUM, group=group)


# ---- bucketed, overlapped exchange -------------------------------------------------------------------
# The backward finishes the density factors' gradients (tf_march_backward + density scatter) before it starts the
# shading backward and the appearance scatter — 0.33 ms of the 0.45 ms backward at config 2.  The exchange therefore
# runs in two buckets: the density rows travel while the shading half computes (the process group's collective stream
# runs beside the compute stream), the rest follows at the end.  No bucket crosses a tensor whose gradient is not final.
def _bucket_segments(model):
    """[(start, stop)] in floats of model.grad_flat for the two buckets: density lines + planes | everything else."""
    offs, grad_len = model.grad_layout
    order = sorted(offs.items(), key=lambda kv: kv[1])
    ends = [order[i + 1][1] if i + 1 < len(order) else grad_len for i in range(len(order))]
    dens, rest = [], []
    for (name, start), stop in zip(order, ends):
        (dens if name.startswith("density_") else rest).append((start, stop))

    def merge(segs):
        out = []
        for a, b in segs:
            if out and out[-1][1] == a:
                out[-1] = (out[-1][0], b)
            else:
                out.append((a, b))
        return out
    return merge(dens), merge(rest)


def _bucket_rows(model):
    """(w, idx32 density rows, idx32 other rows) from gradient_support_rows, or None."""
    rows = gradient_support_rows(model)
    if rows is None:
        return None
    cache = model._rows_cache
    if len(cache) > 5 and cache[5] is not None:
        return cache[5]
    w, idx = rows
    dens, _ = _bucket_segments(model)
    in_d = torch.zeros_like(idx, dtype=torch.bool)
    for a, b in dens:
        in_d |= (idx * w >= a) & (idx * w < b)
    res = (w, idx[in_d].to(torch.int32).contiguous(), idx[~in_d].to(torch.int32).contiguous())
    model._rows_cache = tuple(cache[:5]) + (res,)
    return res


def bucket_gather(model, part):
    """First third of a bucket's exchange, launches only (capturable in a hipGraph): the pieces of model.grad_flat that
    travel for part "density" / "rest" / "all" — with an alpha mask the rows that can be non-zero, gathered into a packed
    buffer (tf_gather_rows), else slices of the buffer itself.  Returns [(tensor to all-reduce, row index | None, row width)].
    ("all": one bucket after the whole backward — the direct-scatter mode, whose line gradients are only folded out of
    their replicas at the very end, tf_reduce_replicas.)"""
    flat = model.grad_flat
    br = _bucket_rows(model)
    items = []
    if br is not None and flat.is_cuda:
        from . import _hip as H
        from .field import _stream
        w, idx_d, idx_r = br
        for idx in ((idx_d,) if part == "density" else (idx_r,) if part == "rest" else (idx_d, idx_r)):
            if idx.numel():
                buf = torch.empty(idx.numel(), w, dtype=torch.float32, device=flat.device)
                H.check(H.lib().tf_gather_rows(flat.data_ptr(), idx.data_ptr(), idx.numel(), w, buf.data_ptr(), _stream()),
                        "tf_gather_rows")
                items.append((buf, idx, w))
    else:
        dens, rest = _bucket_segments(model)
        for a, b in (dens if part == "density" else rest if part == "rest" else sorted(dens + rest)):
            items.append((flat[a:b], None, 0))
    if part != "density" and getattr(model, "_live", None) is not None:
        items.append((model._live, None, 0))      # the sample counts behind the optimizer's gates travel with the gradients
    return items


def bucket_reduce(items, group=None):
    """The collective itself (never inside a capture): one asynchronous all-reduce (sum) per piece; with RCCL it runs on
    the process group's stream, ordered behind the kernels enqueued so far and beside those enqueued next."""
    return [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True) for t, _, _ in items]


def bucket_writeback(model, items, scale=None):
    """Last third, launches only (capturable): gathered rows go back into model.grad_flat (tf_scatter_rows)."""
    flat = model.grad_flat
    for buf, idx, w in items:
        if scale is not None:
            buf.mul_(scale)
        if idx is not None:
            from . import _hip as H
            from .field import _stream
            H.check(H.lib().tf_scatter_rows(flat.data_ptr(), idx.data_ptr(), idx.numel(), w, buf.data_ptr(), _stream()),
                    "tf_scatter_rows")


def exchange_begin(model, part, group=None):
    """Starts the all-reduce (sum) of one bucket of model.grad_flat — part "density" or "rest" — and returns what
    exchange_end needs (bucket_gather + bucket_reduce)."""
    if not dist.is_available() or not dist.is_initialized():      # no process group: a one-process run, nothing travels
        return []
    items = bucket_gather(model, part)
    return list(zip(bucket_reduce(items, group), items))


def exchange_end(model, pending, average=True, group=None):
    """Waits for the bucket (the compute stream waits, not the host, under RCCL) and writes gathered rows back."""
    if not pending:
        return
    for work, _ in pending:
        work.wait()
    bucket_writeback(model, [it for _, it in pending], 1.0 / dist.get_world_size(group) if average else None)


def enable_overlapped_exchange(model, group=None):
    """Eager data-parallel loops: the backward itself starts the density bucket as soon as those gradients are final
    (autograd.backward_launches calls model._density_grads_ready); finish_gradient_exchange then sends the rest and
    waits for both.  Without a process group (or with one rank) nothing is installed."""
    if not dist.is_available() or not dist.is_initialized():
        return False
    if dist.get_world_size(group) == 1 and not FORCE_EXCHANGE:
        return False

    def ready(m):
        m._pending_density = exchange_begin(m, "density", group)
    model._density_grads_ready = ready
    return True


def finish_gradient_exchange(model, group=None, average=True):
    """After loss.backward(): the remaining bucket, then wait for both.  Falls back to the one-shot
    allreduce_gradients when the backward did not start a bucket (no hook installed, direct-scatter mode, CPU)."""
    pend_d = getattr(model, "_pending_density", None)
    model._pending_density = None
    if pend_d is None:
        return allreduce_gradients(model, group=group, average=average)
    pend_r = exchange_begin(model, "rest", group)
    exchange_end(model, pend_d, average, group)
    exchange_end(model, pend_r, average, group)


def allreduce_scalar(value: torch.Tensor, group=None, average=True):
    """Loss / PSNR logging across ranks."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return value
    v = value.detach().clone()
    dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
    return v / dist.get_world_size(group) if average else v


def eval_blocks(n_rays: int, world: int):
    """Contiguous ray blocks per rank (row blocks of an image): [(start, stop)] of length `world`; the
    first `n_rays % world` ranks take one ray more."""
    base, extra = divmod(n_rays, world)
    out, start = [], 0
    for r in range(world):
        stop = start + base + (1 if r < extra else 0)
        out.append((start, stop))
        start = stop
    return out


def render_sharded(render_fn, rays: torch.Tensor, group=None, gather: bool = True):
    """Evaluation across ranks: rank r renders its contiguous block of `rays` with `render_fn(rays_block) -> (rgb (n,3),
    depth (n,))` — e.g. a closure over `OctreeRender_trilinear_fast` — and, with `gather`, every rank receives the whole
    image through one all-gather of (n, 4) tiles (7.7 MB for 800 x 800); without it each rank keeps its block.
    No collective is needed for correctness: rays are independent and the field is replicated."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return render_fn(rays)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    blocks = eval_blocks(rays.shape[0], world)
    a, b = blocks[rank]
    rgb, depth = render_fn(rays[a:b])
    if not gather:
        return rgb, depth
    longest = max(e - s for s, e in blocks)
    tile = torch.zeros(longest, 4, dtype=torch.float32, device=rgb.device)
    tile[: b - a, :3] = rgb
    tile[: b - a, 3] = depth
    tiles = [torch.empty_like(tile) for _ in range(world)]
    dist.all_gather(tiles, tile, group=group)
    full = torch.cat([t[: e - s] for t, (s, e) in zip(tiles, blocks)])
    return full[:, :3].contiguous(), full[:, 3].contiguous()
