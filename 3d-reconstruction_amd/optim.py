"""One-launch Adam for the field's parameters.

`FusedAdam` is a drop-in for the reference's `torch.optim.Adam(grad_vars, betas=(0.9, 0.99))`
(train.py:272-273, stepped at train.py:374-376; rebuilt after upsampling / shrinking at train.py:300-311): same
constructor arguments, same `param_groups` (so `param_group['lr'] *= lr_factor`, train.py:391-392, keeps
working), same update rule.  `step()` is ONE `tf_adam_step` launch over all parameters in storage order
(channel-last planes included) instead of torch's one multi-tensor launch per parameter group and list
chunk; the learning rates and the step count live on the device, so the launch can be captured in a hipGraph
(`graph.GraphedTrainStep`) and still follow a per-step learning-rate schedule.

There is no CPU path: the parameters, gradients and moments must be CUDA fp32 tensors (HipError otherwise)."""
import ctypes as C

import torch

from . import _hip as H
from .field import _stream


def _dense(p):
    from torch._prims_common import is_non_overlapping_and_dense
    return is_non_overlapping_and_dense(p)


def _dense_like(p, g):
    return g is not None and g.dtype == torch.float32 and g.device == p.device and g.stride() == p.stride()


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))
        if len({(g['betas'], g['eps']) for g in self.param_groups}) != 1:
            raise ValueError("FusedAdam: betas / eps must be the same for every parameter group")
        self._flat = None        # (m_flat, v_flat, offsets)
        self._jobs = None        # cached TfAdamJob structs (only the gradient pointers change from step to step)
        self._lr_host = None
        self._lr_dev = self._step_dev = None
        # True: step() also returns every gradient it consumed to zero (TfAdamJob.clear_grads) — zero_grad() folded into the
        # update for callers that accumulate the next step's gradients into the same buffer (graph.GraphedTrainStep)
        self.consume_grads = False

    def _params(self):
        return [(gi, p) for gi, g in enumerate(self.param_groups) for p in g['params']]

    def _init_state(self, ps):
        dev = ps[0][1].device
        offs, total = [], 0
        for _, p in ps:
            if p.dtype != torch.float32 or p.device.type != 'cuda':
                raise H.HipError("FusedAdam needs CUDA fp32 parameters (no CPU path)")
            if not _dense(p):
                raise H.HipError("FusedAdam: parameter is not dense in its storage")
            offs.append(total)
            total += (p.numel() + 63) // 64 * 64
        m = torch.zeros(total, device=dev)
        v = torch.zeros(total, device=dev)
        self._step_dev = torch.zeros((), device=dev)           # completed updates
        self._arrivals = torch.zeros(1, dtype=torch.int32, device=dev)
        # one word per kernel workgroup: which 256-float pieces have ever had a non-zero gradient (TfAdamJob.touched);
        # the moments of the others are still zero and are not read
        n_chunks = sum((p.numel() + H.ADAM_CHUNK - 1) // H.ADAM_CHUNK for _, p in ps)
        self._touched = torch.zeros(n_chunks, dtype=torch.int32, device=dev)
        for (gi, p), o in zip(ps, offs):
            st = self.state[p]
            loaded = st.get('exp_avg'), st.get('exp_avg_sq'), st.get('step')      # state restored by load_state_dict
            st['exp_avg'] = torch.as_strided(m, p.size(), p.stride(), o)
            st['exp_avg_sq'] = torch.as_strided(v, p.size(), p.stride(), o)
            if loaded[0] is not None and loaded[1] is not None:
                st['exp_avg'].copy_(loaded[0])
                st['exp_avg_sq'].copy_(loaded[1])
                self._touched.fill_(-1)                        # moments of unknown history: read everything
                if loaded[2] is not None:
                    self._step_dev.fill_(float(loaded[2]))
            st['step'] = self._step_dev
        self._flat = (m, v, offs, [id(p) for _, p in ps])
        self._jobs = None
        self._lr_dev = torch.zeros(len(self.param_groups), device=dev)
        self._lr_host = None

    def load_state_dict(self, state_dict):
        """Moments restored from a checkpoint replace the flat buffers' views; fold them back in on the next step."""
        super().load_state_dict(state_dict)
        self._flat = None

    def sync_lr(self):
        """Uploads the groups' learning rates when they changed on the host (train.py:391-392 decays them every
        iteration).  Called by step(); GraphedTrainStep calls it before each replay."""
        lrs = [float(g['lr']) for g in self.param_groups]
        if lrs != self._lr_host:
            if torch.cuda.is_current_stream_capturing():
                raise H.HipError("FusedAdam: learning rates changed inside a graph capture")
            # a FRESH pinned tensor per upload (the caching host allocator keeps it alive until the copy has run): with
            # graph replays the host runs many steps ahead of the GPU, and rewriting one staging buffer would let the
            # copy of step k read the rates of a later step
            self._lr_dev.copy_(torch.tensor(lrs, dtype=torch.float32).pin_memory(), non_blocking=True)
            self._lr_host = lrs

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        ps = [(gi, p) for gi, p in self._params() if p.grad is not None]
        if not ps:
            return loss
        if self._flat is None or self._flat[3] != [id(p) for _, p in ps]:
            if self._flat is not None:
                raise H.HipError("FusedAdam: the set of parameters with gradients changed; build a new optimizer "
                                 "(train.py:300-311 does after upsampling / shrinking)")
            self._init_state(ps)
        self.sync_lr()
        m, v, offs, _ = self._flat
        one_launch = len(ps) <= H.ADAM_MAX_SEG       # then the kernel advances the step count itself
        lib, st = H.lib(), _stream()
        if self._jobs is None:                       # everything but the gradient pointers is fixed for this optimizer
            g0 = self.param_groups[0]
            self._jobs, chunk0 = [], 0
            for s0 in range(0, len(ps), H.ADAM_MAX_SEG):
                job = H.TfAdamJob()
                part = ps[s0:s0 + H.ADAM_MAX_SEG]
                run = 0
                for i, (gi, p) in enumerate(part):
                    sg = job.seg[i]
                    sg.p = p.data_ptr()
                    sg.m, sg.v = m.data_ptr() + 4 * offs[s0 + i], v.data_ptr() + 4 * offs[s0 + i]
                    sg.n, sg.group = p.numel(), gi
                    run += (p.numel() + H.ADAM_CHUNK - 1) // H.ADAM_CHUNK
                    job.chunk_end[i] = run
                job.n_seg = len(part)
                job.lrs, job.step = self._lr_dev.data_ptr(), self._step_dev.data_ptr()
                job.beta1, job.beta2, job.eps = g0['betas'][0], g0['betas'][1], g0['eps']
                if one_launch:
                    job.step_rw, job.arrivals = self._step_dev.data_ptr(), self._arrivals.data_ptr()
                job.touched = self._touched.data_ptr() + 4 * chunk0
                chunk0 += run
                self._jobs.append((job, part, [p.data_ptr() for _, p in part]))
        for job, part, ptrs in self._jobs:
            # the HIP backward hands out views of ONE buffer with a fixed layout: when the first and the last gradient sit
            # where they sat relative to each other last step, every gradient does, and only the base moved
            g0, g1, gm = part[0][1].grad, part[-1][1].grad, part[len(part) // 2][1].grad
            if g0 is None or g1 is None or gm is None:
                raise H.HipError("FusedAdam: a parameter lost its gradient between steps")
            base = g0.data_ptr()
            span = (g1.data_ptr() - base, gm.data_ptr() - base)
            memo = getattr(job, "_memo", None)
            if memo is not None and memo[1] == span and g0.stride() == memo[2] and g1.stride() == memo[3]:
                if base != memo[0]:
                    delta = base - memo[0]
                    for i in range(len(part)):
                        job.seg[i].g = job.seg[i].g + delta
                    job._memo = (base, span, memo[2], memo[3])
            else:
                for i, (gi, p) in enumerate(part):
                    g = p.grad
                    if not _dense_like(p, g):
                        raise H.HipError("FusedAdam: a gradient is not laid out like its parameter (expected the HIP "
                                         "backward's gradient views)")
                    if p.data_ptr() != ptrs[i]:
                        raise H.HipError("FusedAdam: a parameter's storage was replaced; build a new optimizer")
                    job.seg[i].g = g.data_ptr()
                job._memo = (base, span, g0.stride(), g1.stride()) if len(part) > 1 else None
            job.clear_grads = int(bool(self.consume_grads))
            H.check(lib.tf_adam_step(C.byref(job), st), "tf_adam_step")
        if not one_launch:
            self._step_dev += 1
        # the kernel wrote the parameters behind autograd's back: caches keyed on ._version (packed weight copies,
        # field.py) must see the change
        torch.autograd.graph.increment_version([p for _, p in ps])
        return loss
