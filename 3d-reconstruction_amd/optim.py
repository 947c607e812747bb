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
        self._flat = None        # (m_flat, v_flat, offsets, parameter ids)
        self._jobs = None        # cached TfAdamJob structs (only the gradient pointers change from step to step)
        self._lr_host = None
        self._lr_dev = self._step_dev = None
        self._reg_active, self._reg_host = None, [0.0, 0.0, 0.0, 0.0]
        # True: step() also returns every gradient it consumed to zero (TfAdamJob.clear_grads) — zero_grad() folded into the
        # update for callers that accumulate the next step's gradients into the same buffer (graph.GraphedTrainStep)
        self.consume_grads = False
        self.kernel_events = None   # bench.py: dict name -> [(start, end)] HIP events around the launch

    def _params(self):
        return [(gi, p) for gi, g in enumerate(self.param_groups) for p in g['params'] if p.requires_grad]

    def _uploads(self):
        if getattr(self, "_ring", None) is None:
            self._ring = H.PinnedRing(max(8, len(self.param_groups)))
        return self._ring

    def set_regularizer_activity(self, ortho=False, l1=False, tv_density=False, tv_app=False):
        """Which regulariser terms are part of the loss (train.py:340-371: `if Ortho_reg_weight > 0 ...`).  They give the
        factor tensors a gradient whether or not the batch produced samples, so they open those tensors' gates
        (TfAdamJob.reg_active; see _init_state).  Uploaded only when the on / off pattern changes."""
        flags = [float(bool(ortho)), float(bool(l1)), float(bool(tv_density)), float(bool(tv_app))]
        if flags != self._reg_host:
            self._reg_host = flags
            if self._reg_active is not None:
                if torch.cuda.is_current_stream_capturing():
                    raise H.HipError("FusedAdam: regulariser activity changed inside a graph capture")
                self._uploads().upload(self._reg_active, flags)

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
        # completed updates PER PARAMETER (torch.optim.Adam's state['step']): a parameter that has no gradient in a step —
        # the reference's appearance tensors and MLP while a fresh field has no shaded sample yet (tensorBase.py:370),
        # host side `p.grad is None`, device side a closed gate (TfAdamSeg.gate) — is left alone and keeps its count
        self._step_dev = torch.zeros(len(ps), device=dev)
        n_jobs = (len(ps) + H.ADAM_MAX_SEG - 1) // H.ADAM_MAX_SEG
        self._arrivals = torch.zeros(n_jobs, dtype=torch.int32, device=dev)
        self._reg_active = torch.tensor(self._reg_host, dtype=torch.float32).to(dev)
        # one word per kernel workgroup: which 256-float pieces have ever had a non-zero gradient (TfAdamJob.touched);
        # the moments of the others are still zero and are not read
        n_chunks = sum((p.numel() + H.ADAM_CHUNK - 1) // H.ADAM_CHUNK for _, p in ps)
        self._touched = torch.zeros(n_chunks, dtype=torch.int32, device=dev)
        for i, ((gi, p), o) in enumerate(zip(ps, offs)):
            st = self.state[p]
            loaded = st.get('exp_avg'), st.get('exp_avg_sq'), st.get('step')      # state restored by load_state_dict
            st['exp_avg'] = torch.as_strided(m, p.size(), p.stride(), o)
            st['exp_avg_sq'] = torch.as_strided(v, p.size(), p.stride(), o)
            if loaded[0] is not None and loaded[1] is not None:
                st['exp_avg'].copy_(loaded[0])
                st['exp_avg_sq'].copy_(loaded[1])
                self._touched.fill_(-1)                        # moments of unknown history: read everything
                if loaded[2] is not None:
                    self._step_dev[i] = float(loaded[2])
            st['step'] = self._step_dev[i]
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
            # through a ring of pinned staging buffers (H.PinnedRing): with graph replays the host runs many steps ahead of
            # the GPU, and rewriting ONE staging buffer would let the copy of step k read the rates of a later step
            self._uploads().upload(self._lr_dev, lrs)
            self._lr_host = lrs

    def _build_jobs(self, ps):
        m, v, offs, _ = self._flat
        g0 = self.param_groups[0]
        self._jobs, chunk0 = [], 0
        live = None
        for _, p in ps:
            tag = getattr(p, "_tf_gate", None)
            if tag is not None:
                if live is not None and tag[0] is not live:
                    raise H.HipError("FusedAdam: the parameters belong to more than one field model")
                live = tag[0]
        for ji, s0 in enumerate(range(0, len(ps), H.ADAM_MAX_SEG)):
            job = H.TfAdamJob()
            part = ps[s0:s0 + H.ADAM_MAX_SEG]
            run = 0
            for i, (gi, p) in enumerate(part):
                sg = job.seg[i]
                sg.p = p.data_ptr()
                sg.m, sg.v = m.data_ptr() + 4 * offs[s0 + i], v.data_ptr() + 4 * offs[s0 + i]
                sg.n, sg.group = p.numel(), gi
                tag = getattr(p, "_tf_gate", None)
                sg.gate = int(tag[1]) if tag is not None else 0
                run += (p.numel() + H.ADAM_CHUNK - 1) // H.ADAM_CHUNK
                job.chunk_end[i] = run
            job.n_seg = len(part)
            job.lrs, job.step = self._lr_dev.data_ptr(), self._step_dev.data_ptr() + 4 * s0
            job.beta1, job.beta2, job.eps = g0['betas'][0], g0['betas'][1], g0['eps']
            job.arrivals = self._arrivals.data_ptr() + 4 * ji
            job.touched = self._touched.data_ptr() + 4 * chunk0
            job.live = live.data_ptr() if live is not None else None
            job.reg_active = self._reg_active.data_ptr()
            chunk0 += run
            self._jobs.append([job, part, [p.data_ptr() for _, p in part], None, live])

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        ps = self._params()
        if not ps or all(p.grad is None for _, p in ps):
            return loss
        if self._flat is None or self._flat[3] != [id(p) for _, p in ps]:
            if self._flat is not None:
                raise H.HipError("FusedAdam: the set of parameters changed; build a new optimizer "
                                 "(train.py:300-311 does after upsampling / shrinking)")
            self._init_state(ps)
        self.sync_lr()
        lib, st = H.lib(), _stream()
        if self._jobs is None:                       # everything but the gradient pointers is fixed for this optimizer
            self._build_jobs(ps)
        for rec in self._jobs:
            job, part, ptrs, memo, _live = rec
            # where every gradient sits this step (None: the parameter has none, torch.optim.Adam would skip it).  The HIP
            # backward hands out views of ONE buffer with a fixed layout: when every gradient sits where it sat relative to
            # the first one last step, layouts and strides are the ones validated then, and only the base may have moved
            gp, skip, base = [], 0, 0
            for i, (gi, p) in enumerate(part):
                g = p.grad
                if g is None:
                    skip |= 1 << i
                    gp.append(None)
                else:
                    a = g.data_ptr()
                    if not base:
                        base = a
                    gp.append(a - base)
                if p.data_ptr() != ptrs[i]:
                    raise H.HipError("FusedAdam: a parameter's storage was replaced; build a new optimizer")
            if not base:
                continue                              # no gradient in this job at all: nothing is updated or counted
            if memo is not None and memo[1] == gp:
                if base != memo[0]:
                    delta = base - memo[0]
                    for i in range(len(part)):
                        if gp[i] is not None:
                            job.seg[i].g = job.seg[i].g + delta
                    memo[0] = base
            else:
                for i, (gi, p) in enumerate(part):
                    g = p.grad
                    if g is None:
                        job.seg[i].g = ptrs[i]        # (never read: the segment is skipped)
                        continue
                    if not _dense_like(p, g):
                        raise H.HipError("FusedAdam: a gradient is not laid out like its parameter (expected the HIP "
                                         "backward's gradient views)")
                    job.seg[i].g = g.data_ptr()
                rec[3] = [base, gp]
            job.skip_mask = skip
            job.clear_grads = int(bool(self.consume_grads))
            ev = self.kernel_events
            if ev is not None and not torch.cuda.is_current_stream_capturing():     # bench: HIP events around the launch
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                H.check(lib.tf_adam_step(C.byref(job), st), "tf_adam_step")
                b.record()
                ev.setdefault("tf_adam_step", []).append((a, b))
            else:
                H.check(lib.tf_adam_step(C.byref(job), st), "tf_adam_step")
        # the kernel wrote the parameters behind autograd's back: caches keyed on ._version (packed weight copies,
        # field.py) must see the change
        torch.autograd.graph.increment_version([p for _, p in ps])
        return loss
