"""Regularisers on the factor tensors that the reference's training loop adds to the MSE (train.py:340-371).

`TVLoss` keeps the reference's call interface (`tvreg = TVLoss(); tensorf.TV_loss_density(tvreg)`, loss.py:120-141):
for a plane x of shape (B, C, H, W) it returns  weight * 2 * (sum (d_h x)^2 / (C (H-1) W) + sum (d_w x)^2 / (C H (W-1))) / B.
SURVEY row f-3: these run as plain torch ops on the channel-last parameters for now (not part of the
ray-marching kernels)."""
import torch


def total_variation(x: torch.Tensor) -> torch.Tensor:
    b, c, h, w = x.shape
    dh = torch.diff(x, dim=2)
    dw = torch.diff(x, dim=3)
    return 2.0 * (dh.square().sum() / (c * (h - 1) * w) + dw.square().sum() / (c * h * (w - 1))) / b


class TVLoss(torch.nn.Module):
    def __init__(self, TVLoss_weight=1):
        super().__init__()
        self.TVLoss_weight = TVLoss_weight

    def forward(self, x):
        return self.TVLoss_weight * total_variation(x)


# ---------------------------------------------------------------------------------------------------------
# One-pass HIP version of the four terms for TensorVMSplit (tf_regularizers, csrc/reg.hip)
def _reg_job(model, w_ortho, w_l1, w_tv_density, w_tv_app, loss, grads, scale, weights_dev=None):
    import ctypes as C
    from . import _hip as H
    from .field import is_channel_last
    job = H.TfRegJob()
    for part, planes, lines, comps in (("density", model.density_plane, model.density_line, model.density_n_comp),
                                       ("app", model.app_plane, model.app_line, model.app_n_comp)):
        fac, fg = getattr(job, part), getattr(job, part + "_grad")
        for i in range(3):
            p, l = planes[i], lines[i]
            if not (p.is_cuda and is_channel_last(p) and is_channel_last(l)):
                raise H.HipError("fused regularisers need the channel-last CUDA factor tensors of TensorVMSplit")
            fac.plane[i], fac.line[i], fac.n_comp[i] = p.data_ptr(), l.data_ptr(), int(comps[i])
            if grads is not None:
                gp, gl = grads[id(p)], grads[id(l)]
                if gp.stride() != p.stride() or gl.stride() != l.stride() or gp.dtype != torch.float32:
                    raise H.HipError("fused regularisers: a gradient is not laid out like its parameter")
                fg.plane[i], fg.line[i] = gp.data_ptr(), gl.data_ptr()
        fg.n_rep, fg.rep_stride = 1, 0
    geom = getattr(model, "_geom", None)        # host copy of the geometry (no device read: usable inside a capture)
    grid = [int(g) for g in geom["grid"]] if geom is not None else [int(g) for g in model.gridSize.tolist()]
    for k in range(3):
        job.grid[k] = grid[k]
    job.w_ortho, job.w_l1, job.w_tv_density, job.w_tv_app = float(w_ortho), float(w_l1), float(w_tv_density), float(w_tv_app)
    job.loss = loss.data_ptr()
    job.scale = scale.data_ptr() if scale is not None else None
    job.want_grad = int(grads is not None)
    job.weights_dev = weights_dev.data_ptr() if weights_dev is not None else None
    return job


def fused_supported(model):
    """VM decomposition with whole channel quads of at most 64 components per tensor, parameters on the GPU."""
    comps = list(getattr(model, "density_n_comp", [])) + list(getattr(model, "app_n_comp", []))
    return (hasattr(model, "density_plane") and len(comps) == 6 and all(c % 4 == 0 and 0 < c <= 64 for c in comps)
            and model.density_plane[0].is_cuda)


def _factor_params(model):
    return list(model.density_plane) + list(model.density_line) + list(model.app_plane) + list(model.app_line)


def _launch(model, weights, grads, scale, weights_dev=None):
    import ctypes as C
    from . import _hip as H
    from .field import _stream
    loss = torch.zeros(4, dtype=torch.float32, device=model.density_plane[0].device)
    job = _reg_job(model, *weights, loss, grads, scale, weights_dev)
    H.check(H.lib().tf_regularizers(C.byref(job), _stream()), "tf_regularizers")
    return loss


class _FusedRegFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, weights, *params):
        ctx.model, ctx.weights, ctx.params = model, weights, params
        return _launch(model, weights, None, None)[0].clone()

    @staticmethod
    def backward(ctx, gout):
        params = ctx.params
        total = sum((p.numel() + 63) // 64 * 64 for p in params)
        flat = torch.zeros(total, dtype=torch.float32, device=params[0].device)
        grads, off = {}, 0
        for p in params:
            grads[id(p)] = torch.as_strided(flat, p.size(), p.stride(), off)
            off += (p.numel() + 63) // 64 * 64
        _launch(ctx.model, ctx.weights, grads, gout.detach().to(torch.float32).contiguous())
        return (None, None) + tuple(grads[id(p)] for p in params)


def fused_regularizers(model, ortho_weight=0.0, l1_weight=0.0, tv_weight_density=0.0, tv_weight_app=0.0):
    """Differentiable scalar
        ortho_weight * vector_comp_diffs() + l1_weight * density_L1() + tv_weight_density * TV_loss_density(TVLoss())
        + tv_weight_app * TV_loss_app(TVLoss())
    of a TensorVMSplit, computed by two HIP launches (and two more in backward) instead of ~200 eager ones:
    `total_loss = loss + fused_regularizers(tensorf, ...)` replaces train.py:340-371."""
    w = (ortho_weight, l1_weight, tv_weight_density, tv_weight_app)
    return _FusedRegFn.apply(model, w, *_factor_params(model))


@torch.no_grad()
def add_regularizer_grads_(model, ortho_weight=0.0, l1_weight=0.0, tv_weight_density=0.0, tv_weight_app=0.0,
                           weights_dev=None):
    """The same sum, with its gradient ADDED to the existing `.grad` of the factor tensors in the same pass (call it
    between `loss.backward()` and `optimizer.step()`); returns the 4 device floats [total, TV, L1, ortho].
    `weights_dev`: 4 device floats [ortho, l1, tv_density, tv_app] multiplying the host weights (pass 1.0 for those) —
    for captured steps, where the per-iteration decay of the TV weights (train.py:336-339) must stay adjustable."""
    grads = {}
    for p in _factor_params(model):
        if p.grad is None:
            p.grad = torch.zeros_like(p)      # zeros_like keeps the channel-last strides
        grads[id(p)] = p.grad
    return _launch(model, (ortho_weight, l1_weight, tv_weight_density, tv_weight_app), grads, None, weights_dev)
