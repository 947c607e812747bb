"""Differentiable entry of the HIP ray-marching path.

`TensorBase.forward` routes here whenever gradients are enabled.  The forward launches the same three
kernels as inference (keeping the per-ray valid-sample lists); the backward launches
`tf_march_backward` (compositing + density factors) and `tf_shade_backward` (shading MLP,
basis matrix, appearance factors) and hands PyTorch one gradient tensor per parameter, laid out like the
parameter (channel-last for the factor tensors), so `torch.optim.Adam` and the reference training loop
(train.py:374-376) work unchanged."""
import ctypes as C
import weakref

import torch

from . import _hip as H
from .field import _stream, is_channel_last


N_REP = 64   # replicas of the line-gradient tensors (TfFactorGrads.n_rep)


_LAYOUTS = {}     # tuple of (name, shape, stride) -> (offs, total, line_len, [(name, size, stride, offset)])


def _grad_layout(named):
    """Where every parameter's gradient sits in the step's flat buffer: line tensors first (the direct-scatter
    replicas follow the buffer), everything 64-float aligned; gradients are laid out like their parameters (channel-last
    for the factor tensors).  Cached per parameter set: building 19 views costs ~0.1 ms of host time per step otherwise."""
    key = tuple((n, tuple(p.shape), tuple(p.stride())) for n, p in named)
    hit = _LAYOUTS.get(key)
    if hit is not None:
        return hit
    order = sorted(named, key=lambda kv: 0 if '_line.' in kv[0] else 1)
    offs, total, line_len, views = {}, 0, 0, []
    for name, p in order:
        offs[name] = total
        if p.dim() == 4 and is_channel_last(p):
            b, c, h, w = p.shape
            stride = (h * w * c, 1, w * c, c)
        else:
            stride = tuple(torch.empty(p.shape, device='meta').stride())
        views.append((name, tuple(p.shape), stride, total))
        total += (p.numel() + 63) // 64 * 64
        if '_line.' in name:
            line_len = total
    if len(_LAYOUTS) > 64:
        _LAYOUTS.clear()
    _LAYOUTS[key] = (offs, total, line_len, views)
    return _LAYOUTS[key]


def _grad_buffers(named, n_rep=N_REP, store=None):
    """One zero-filled allocation: [line gradients | all other gradients | n_rep replicas of the line block].
    The direct-scatter kernels spread line-gradient atomics over the replicas; tf_reduce_replicas folds them
    into the head of the buffer, so flat[:grad_len] is every parameter gradient of the step, contiguous (one
    all-reduce).  The binned scatter flushes each line bucket once per work item and needs no replicas.
    `store` (graph.GraphedTrainStep, binned scatter only): {'flat', 'clean'} — a buffer that lives across steps; its
    owner's optimizer returns the gradients it consumed to zero (FusedAdam.consume_grads), so a clean buffer is handed out
    as it is: no 70 MB fill per step at config 2."""
    offs, total, line_len, layout = _grad_layout(named)
    dev = named[0][1].device
    if store is not None and n_rep == 0:
        flat = store.get('flat')
        if flat is None or flat.numel() != total or flat.device != dev:
            flat = store['flat'] = torch.zeros(total, dtype=torch.float32, device=dev)
        elif not store.get('clean', False):      # (a step that never reached its optimizer: an exception, a dropped loss)
            flat.zero_()
        store['clean'] = False
    else:
        flat = torch.zeros(total + n_rep * line_len, dtype=torch.float32, device=dev)
    views = {name: torch.as_strided(flat, size, stride, off) for name, size, stride, off in layout}
    return views, flat, offs, total, line_len


def _bin_job(model, ws, part, factors, grid, stage, fgrads=None, grad=None, grad_ld=0):
    """TfBinJob of the density / appearance gradient scatter; every job owns its sort workspace."""
    app = part == "app"
    # sorted early (stage 1, then 2), beside tf_shade_forward: one key per 16-component group — the cheaper shared-key sort
    # slows the shading kernel it runs beside by more than it saves; sorted in the backward (stage 0): shared keys
    share = 0 if stage in (1, 2) else 1
    nkeys = ws.binned_cfg[(5 if app else 4) if share else (1 if app else 0)]
    nmax = max(ws.binned_cfg[0], ws.binned_cfg[1])
    j = H.TfBinJob()
    j.share_groups = share
    j.model = H.MODEL_CP if model._is_cp() else H.MODEL_VM
    j.factors = factors
    if fgrads is not None:
        j.grads = fgrads
    j.grid = grid
    j.counters, j.slot, j.seg_cap = ws.counters.data_ptr(), (0 if app else 3), (ws.seg_cap if app else ws.ent_seg_cap)
    j.xyz = (ws.app_xyz if app else ws.ent_xyz).data_ptr()
    j.grad, j.grad_ld = (grad.data_ptr() if grad is not None else None), grad_ld
    j.tile, j.bucket, j.chunk = ws.binned_cfg[3], model.bin_bucket, (model.bin_chunk if share else model.bin_chunk_early)
    ints = (ws.bin_ints_app if app else ws.bin_ints).data_ptr()
    j.hist = (ws.hist_app if app else ws.hist_density).data_ptr()
    j.hist_zeroed = 1                     # zeroed with the shard counters at the start of the forward
    j.offsets = ints
    j.cursor, j.chunk_off = ints + 4 * (nmax + 8), ints + 8 * (nmax + 8)
    j.binned, j.nkeys = (ws.binned_app if app else ws.binned).data_ptr(), nkeys
    j.stage = stage
    # capacities the kernels check every position against (TfBinJob.status collects violations)
    j.binned_cap = ws.binned_app_len if app else ws.binned_len
    j.items_cap = (ws.bin_ints_len - (2 * (nmax + 8) + nkeys + 1) - 4) // 4      # int4 items behind chunk_off[nkeys + 1] (16-B aligned)
    j.status = ws.bin_status.data_ptr()
    return j


def _early_sort(model, ws, field, shade, named):
    """Fork: both entry lists of the backward's binned scatter are complete once tf_march_forward has run (the
    appearance samples' coordinates, and — TfMarchIO.ent_xyz — the density samples'), so their counting sorts
    (three small, latency-bound kernels each) are issued on a second stream here and run next to the shading
    kernels.  The backward joins before its first scatter.  Works the same inside a hipGraph capture."""
    main = torch.cuda.current_stream()
    if getattr(model, "_sort_inline", False):     # bench.py's per-kernel timing pass: same launches, no overlap
        lib = H.lib()
        model._timed("tf_binned_sort_pair", lib.tf_binned_sort_pair,
                     C.byref(_bin_job(model, ws, "density", field.density, field.grid, 1)),
                     C.byref(_bin_job(model, ws, "app", shade.app, field.grid, 1)), _stream())
        return main, _grad_buffers(named, 0, getattr(model, "_grad_store", None))
    if model._sort_stream is None:
        model._sort_stream = torch.cuda.Stream(device=main.device, priority=-1)   # its few workgroups go first
    side = model._sort_stream
    side.wait_stream(main)
    lib = H.lib()
    with torch.cuda.stream(side):
        st = _stream()
        model._timed("tf_binned_sort_pair", lib.tf_binned_sort_pair,
                     C.byref(_bin_job(model, ws, "density", field.density, field.grid, 1)),
                     C.byref(_bin_job(model, ws, "app", shade.app, field.grid, 1)), st)
        # the step's (zero-filled) gradient buffer is produced here as well: 70 MB of fill off the main stream
        bufs = _grad_buffers(named, 0, getattr(model, "_grad_store", None))
        bufs[1].record_stream(main)
    return side, bufs


class _RenderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, rays, mask, white_bg, is_train, ndc_ray, N_samples, names, *params):
        ctx.set_materialize_grads(False)     # no zero-filled gradients for the two non-differentiable results
        named_fwd = list(zip(names, params))
        c = model._run_forward(rays, mask, white_bg, is_train, ndc_ray, N_samples, save_valid=True,
                               after_march=lambda ws, field, shade: _early_sort(model, ws, field, shade, named_fwd))
        ws = c['ws']
        ws.owner = weakref.ref(ctx)          # lets the pool reclaim the workspace if this graph is dropped un-backwarded
        ctx.early_bufs = None
        if c.get('sorted_on') is not None:      # (second stream, gradient buffers): the buffers travel on ctx only —
            c['sorted_on'], ctx.early_bufs = c['sorted_on']   # a second owner (model.last) would make autograd copy them
        ctx.model, ctx.c, ctx.names = model, c, names
        ctx.versions = [p._version for p in params]
        ctx.params = params
        rgb_map, depth = c.pop('rgb_map'), c.pop('depth')     # fresh tensors written by the kernels
        # the third result (`app_mask.sum()`, tensorBase.py:390) costs a reduction launch; callers that drop it
        # (graph.GraphedTrainStep) switch it off and get the un-summed per-shard counters' first entry instead
        nvalid = c['n_shaded'] if c['n_shaded'] is not None else ws.counters2d[0, 0]
        ctx.mark_non_differentiable(depth, nvalid)
        return rgb_map, depth, nvalid

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_n):
        model, c, names = ctx.model, ctx.c, ctx.names
        for p, v in zip(ctx.params, ctx.versions):
            if p._version != v:
                raise RuntimeError("a parameter was modified in place between forward and backward")
        named = list(zip(names, ctx.params))
        if g_rgb is None:
            g_rgb = torch.zeros(c['ws'].R, 3, dtype=torch.float32, device=ctx.params[0].device)
        early, ctx.early_bufs = ctx.early_bufs, None
        slot = c.get('live_slot')
        grads = backward_launches(model, c, named, g_rgb, early, "all")
        n_density = n_shaded = 1
        if slot is not None:
            # The parameters the reference's autograd graph would not contain get no gradient here either: density factors
            # without a valid sample (tensorBase.py:359), appearance factors / basis / MLP without a shaded one (:370).  The
            # counts were written to pinned memory by the forward's compositing launch; its event has usually fired by now
            # (the wait is on the forward only, the backward launches above are already queued).
            slot[1].synchronize()
            n_density, n_shaded, overflow, _ = slot[0].tolist()
            model._last_sample_counts = (n_density, n_shaded)      # early_sort = 'auto' decides the next step from them
            ws = c['ws']
            if ws.right_sized:
                if overflow:      # this batch did not fit its workspace: the gradients above are incomplete
                    ctr = ws.counters2d[:, :2].cpu()
                    model._grow_caps(ws.R, ws.N, int(ctr[:, 0].max()), int(ctr[:, 1].max()))
                    ctx.c = None
                    raise H.WorkspaceOverflow(
                        f"training batch needs {int(ctr[:, 0].max())} shaded / {int(ctr[:, 1].max())} density entries per shard, "
                        f"the workspace holds {ws.seg_cap} / {ws.ent_seg_cap}: it has been enlarged — run the step again "
                        f"(model.retry_on_overflow(step_fn) repeats it with the same random draws)",
                        jitter=c['keep'][3], z_table=c['keep'][4], use_bg=c['use_bg'])
                # Room is added BEFORE it runs out (shards fill unevenly: 1.15 x the mean is taken for the fullest).  While a
                # young field's sample counts multiply from step to step the headroom follows their growth rate r (room
                # for r^2 x the present demand, at most 4 x); once they settle it is 1.3 x.
                per = 1.15 / H.N_SHARDS
                need = (n_shaded * per, n_density * per)
                prev = model._need_prev.get((ws.R, ws.N), (0.0, 0.0))
                model._need_prev[(ws.R, ws.N)] = need
                f = [min(4.0, max(1.3, (need[i] / prev[i]) ** 2 if prev[i] > 0 else 1.3)) for i in (0, 1)]
                if need[0] * f[0] > ws.seg_cap or need[1] * f[1] > ws.ent_seg_cap:
                    # (a quarter more than the trigger level: batch-to-batch fluctuations of a few per cent must not grow —
                    #  and re-validate — the workspace every other step)
                    model._grow_caps(ws.R, ws.N, need[0] * f[0] * 1.25, need[1] * f[1] * 1.25, factor=1.0)
            if not model.reference_none_grads:
                n_density = n_shaded = 1
        out = tuple((grads[n] if (n_density if n.startswith('density_') else n_shaded) else None) if p.requires_grad else None
                    for n, p in named)
        ctx.c = None
        return (None,) * 8 + out


def backward_launches(model, c, named, g_rgb, early_bufs=None, stage="all"):
    """The backward's kernel launches for the forward context `c` (TensorBase._run_forward(save_valid=True)).
    stage "all": everything (autograd path).  The data-parallel graphed step cuts it in two so that the exchange of the
    density gradients can run beside the shading backward (graph.py):
      "density": gradient buffer, tf_march_backward, density scatter  -> the density factors' gradients are final;
      "shade"  : tf_shade_backward, appearance scatter                -> everything else.
    Returns {parameter name: gradient view} (views of ONE zero-filled buffer, model.grad_flat)."""
    lib = H.lib()
    ws = c['ws']
    st = _stream()
    binned = ws.binned_cfg is not None
    n_rep = 0 if binned else N_REP
    presorted = c.get('sorted_on') is not None
    cp = model._is_cp()
    if stage in ("all", "density"):
        if presorted and early_bufs is not None:   # made on the second stream during the forward (_early_sort)
            grads, flat, offs, grad_len, line_len = early_bufs
        else:
            grads, flat, offs, grad_len, line_len = _grad_buffers(named, n_rep, getattr(model, "_grad_store", None))
        model.grad_flat = flat[:grad_len]   # every gradient of this step, one contiguous buffer (parallel.py)
        if getattr(model, "grad_layout", None) is None or model.grad_layout[0] is not offs:
            model.grad_layout = (offs, grad_len)     # name -> offset (floats): parallel.gradient_support
        g = g_rgb.detach().to(torch.float32).contiguous()
        rep0 = flat.data_ptr() + 4 * (grad_len if n_rep else 0)   # replica 0 of the line block (binned: the head itself)
        dg = H.TfFactorGrads()
        ag = H.TfFactorGrads()
        for fg, kind in ((dg, 'density'), (ag, 'app')):
            fg.n_rep, fg.rep_stride = max(n_rep, 1), line_len
            for i in range(3):
                if not cp:
                    fg.plane[i] = grads[f'{kind}_plane.{i}'].data_ptr()
                fg.line[i] = rep0 + 4 * offs[f'{kind}_line.{i}']
        c['_bwd'] = (grads, flat, offs, grad_len, line_len, rep0, dg, ag, g)
        model._timed("tf_march_backward", lib.tf_march_backward, C.byref(c['field']), C.byref(c['io']), g.data_ptr(),
                     ws.rgb_pre.data_ptr(), int(c['use_bg']), ws.rgb.data_ptr(), ws.grad_rgb.data_ptr(), C.byref(dg),
                     ws.ent_xyz.data_ptr() if binned else None, ws.ent_df.data_ptr() if binned else None, st)
        if binned:
            if presorted:       # join the second stream.  One cross-stream wait costs ~10 us on the queue that waits, wherever
                # it stands (placed behind tf_shade_backward, long after the sorts had finished, it cost the same), so
                # there is only this one
                torch.cuda.current_stream().wait_stream(c['sorted_on'])
            j = _bin_job(model, ws, "density", c['field'].density, c['field'].grid, 2 if presorted else 0, dg, ws.ent_df, 0)
            model._timed("tf_binned_scatter_density", lib.tf_binned_scatter, C.byref(j), st)
        hook = getattr(model, "_density_grads_ready", None)
        if hook is not None and binned:     # eager data parallel: the density gradients can travel now (parallel.py)
            hook(model)
        if stage == "density":
            return grads
    grads, flat, offs, grad_len, line_len, rep0, dg, ag, g = c['_bwd']
    sg = H.TfShadeGrads()
    sg.w1, sg.b1 = grads['renderModule.mlp.0.weight'].data_ptr(), grads['renderModule.mlp.0.bias'].data_ptr()
    sg.w2, sg.b2 = grads['renderModule.mlp.2.weight'].data_ptr(), grads['renderModule.mlp.2.bias'].data_ptr()
    sg.w3, sg.b3 = grads['renderModule.mlp.4.weight'].data_ptr(), grads['renderModule.mlp.4.bias'].data_ptr()
    sg.basis = grads['basis_mat.weight'].data_ptr()
    sg.app = ag
    sg.dv_out, sg.wslab = ws.dv.data_ptr(), ws.wslab.data_ptr()
    sg.x_saved, sg.rgb_fwd = ws.xs.data_ptr(), ws.rgb.data_ptr()
    sg.h1_saved, sg.h2_saved = ws.h1s.data_ptr(), ws.h2s.data_ptr()
    sg.direct_scatter = 0 if binned else 1
    model._timed("tf_shade_backward", lib.tf_shade_backward, C.byref(c['shade']), c['rays'].data_ptr(),
                 int(c['ndc']), ws.counters.data_ptr(), ws.seg_cap, ws.app_ray.data_ptr(), ws.app_xyz.data_ptr(),
                 ws.grad_rgb.data_ptr(), C.byref(sg), st)
    if binned:
        j = _bin_job(model, ws, "app", c['shade'].app, c['field'].grid, 2 if presorted else 0, ag, ws.dv, model._n_app_total())
        model._timed("tf_binned_scatter_app", lib.tf_binned_scatter, C.byref(j), st)
    if n_rep:
        model._timed("tf_reduce_replicas", lib.tf_reduce_replicas, rep0, n_rep, line_len, line_len, flat.data_ptr(), st)
    ws.busy, ws.owner = False, None     # stream order: the next forward that takes this workspace runs after these kernels
    c.pop('_bwd', None)
    return grads


def render_with_grad(model, rays, mask, white_bg, is_train, ndc_ray, N_samples):
    if model.shadingMode in ('SH', 'RGB'):
        raise H.HipError("training with the SH / RGB heads is not implemented in the HIP backward "
                         "(the reference cannot construct them either, models/tensorBase.py:89-98)")
    # the (name, parameter) list is cached; only the factor tensors are ever replaced (shrink / upsample_volume_grid /
    # load), so their identities are the cache key — walking model.parameters() every call costs 0.1 ms of host time
    sig = tuple(id(p) for n in ("density_plane", "density_line", "app_plane", "app_line") if hasattr(model, n)
                for p in getattr(model, n)) + (id(model.basis_mat.weight),) + \
        tuple(id(p) for p in model.renderModule.parameters())
    cache = model._named_cache
    if cache is None or cache[2] != sig:
        named = list(model.named_parameters())
        model._tag_parameters(named)
        cache = model._named_cache = (tuple(n for n, _ in named), [p for _, p in named], sig)
    return _RenderFn.apply(model, rays, mask, white_bg, is_train, ndc_ray, N_samples, cache[0], *cache[1])
