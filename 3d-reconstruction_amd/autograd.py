"""Differentiable entry of the HIP ray-marching path.

`TensorBase.forward` routes here whenever gradients are enabled.  The forward launches the same three
kernels as inference (keeping the per-ray valid-sample lists); the backward launches
`tf_march_backward` (compositing + density factors) and `tf_shade_backward` (shading MLP,
basis matrix, appearance factors) and hands PyTorch one gradient tensor per parameter, laid out like the
parameter (channel-last for the factor tensors), so `torch.optim.Adam` and the reference training loop
(train.py:374-376) work unchanged."""
import ctypes as C

import torch

from . import _hip as H
from .field import _stream, is_channel_last


def _grad_buffers(named):
    """One zero-filled allocation carved into per-parameter gradient views (factor tensors channel-last)."""
    sizes = [p.numel() for _, p in named]
    offs, total = [], 0
    for n in sizes:
        offs.append(total)
        total += (n + 63) // 64 * 64
    dev = named[0][1].device
    flat = torch.zeros(total, dtype=torch.float32, device=dev)
    out = {}
    for (name, p), off, n in zip(named, offs, sizes):
        chunk = flat[off:off + n]
        if p.dim() == 4 and is_channel_last(p):
            b, c, h, w = p.shape
            out[name] = chunk.view(b, h, w, c).permute(0, 3, 1, 2)
        else:
            out[name] = chunk.view(p.shape)
    return out, flat


class _RenderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, rays, mask, white_bg, is_train, ndc_ray, N_samples, names, *params):
        c = model._run_forward(rays, mask, white_bg, is_train, ndc_ray, N_samples, save_valid=True)
        ws = c['ws']
        ctx.model, ctx.c, ctx.names = model, c, names
        ctx.versions = [p._version for p in params]
        ctx.params = params
        rgb_map = ws.rgb_map.view(ws.R, 3).clone()
        depth = ws.depth.clone()
        nvalid = ws.counters2d[:, 0].sum()
        ctx.mark_non_differentiable(depth, nvalid)
        return rgb_map, depth, nvalid

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_n):
        model, c, names = ctx.model, ctx.c, ctx.names
        for p, v in zip(ctx.params, ctx.versions):
            if p._version != v:
                raise RuntimeError("a parameter was modified in place between forward and backward")
        lib = H.lib()
        ws = c['ws']
        st = _stream()
        named = list(zip(names, ctx.params))
        grads, flat = _grad_buffers(named)
        model.grad_flat = flat      # all parameter gradients of this step, one contiguous buffer (parallel.py)
        g = g_rgb.detach().to(torch.float32).contiguous()
        cp = model._is_cp()

        dg = H.TfFactorGrads()
        ag = H.TfFactorGrads()
        for i in range(3):
            if not cp:
                dg.plane[i] = grads[f'density_plane.{i}'].data_ptr()
                ag.plane[i] = grads[f'app_plane.{i}'].data_ptr()
            dg.line[i] = grads[f'density_line.{i}'].data_ptr()
            ag.line[i] = grads[f'app_line.{i}'].data_ptr()
        model._timed("tf_march_backward", lib.tf_march_backward, C.byref(c['field']), C.byref(c['io']), g.data_ptr(),
                     ws.rgb_pre.data_ptr(), int(c['use_bg']), ws.rgb.data_ptr(), ws.grad_rgb.data_ptr(), C.byref(dg), st)
        sg = H.TfShadeGrads()
        sg.w1, sg.b1 = grads['renderModule.mlp.0.weight'].data_ptr(), grads['renderModule.mlp.0.bias'].data_ptr()
        sg.w2, sg.b2 = grads['renderModule.mlp.2.weight'].data_ptr(), grads['renderModule.mlp.2.bias'].data_ptr()
        sg.w3, sg.b3 = grads['renderModule.mlp.4.weight'].data_ptr(), grads['renderModule.mlp.4.bias'].data_ptr()
        sg.basis = grads['basis_mat.weight'].data_ptr()
        sg.app = ag
        model._timed("tf_shade_backward", lib.tf_shade_backward, C.byref(c['shade']), c['rays'].data_ptr(),
                     int(c['ndc']), ws.counters.data_ptr(), ws.seg_cap, ws.app_ray.data_ptr(), ws.app_xyz.data_ptr(),
                     ws.grad_rgb.data_ptr(), C.byref(sg), st)
        out = tuple(grads[n] if p.requires_grad else None for n, p in named)
        ctx.c = None
        return (None,) * 8 + out


def render_with_grad(model, rays, mask, white_bg, is_train, ndc_ray, N_samples):
    if model.shadingMode in ('SH', 'RGB'):
        raise H.HipError("training with the SH / RGB heads is not implemented in the HIP backward "
                         "(the reference cannot construct them either, models/tensorBase.py:89-98)")
    named = [(n, p) for n, p in model.named_parameters()]
    names = tuple(n for n, _ in named)
    return _RenderFn.apply(model, rays, mask, white_bg, is_train, ndc_ray, N_samples, names, *[p for _, p in named])
