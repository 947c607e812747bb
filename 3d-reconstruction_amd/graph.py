"""hipGraph-captured training step.

At MI355X speeds one train step of this path is ~1.5 ms of GPU work launched as ~40 kernels; driving it from
Python costs about as much host time (autograd dispatch, optimizer, ctypes), so the eager loop is launch-bound.
`GraphedTrainStep` captures forward -> MSE -> backward -> Adam into ONE hipGraph (torch.cuda.CUDAGraph on the
stream the C-ABI launches use) and replays it per step.  Everything data-dependent already lives on the
device (sample counts, packed lists, bins), so the captured launch sequence is step-invariant.  Host work per
step is: draw the stratified jitter from the CPU generator exactly like the reference (tensorBase.py:201),
stage the batch into the static input buffers, replay.

Data parallel (`world_size > 1`): the step is captured as THREE graphs — (a) forward / loss / backward up to the density
gradients, (b) shading backward + appearance scatter, (c) regularisers + optimizer — with the gradient exchange
(RCCL all-reduces of the two buckets' packed rows: `parallel.bucket_gather` closes graphs (a) and (b), `bucket_writeback`
opens (c), only `bucket_reduce` runs eagerly between the replays): the density bucket travels while (b) replays, so no
collective is ever inside a capture and only the second bucket is exposed.

When the schedule replaces the alpha mask or the parameters (updateAlphaMask, shrink, upsample_volume_grid — the
caller then assigns the rebuilt optimizer to `.opt`, as train.py:300-311 rebuilds it), the next `step` notices,
runs one eager step and captures again.

Restrictions (else use the eager path): fixed batch size / N_samples.  With `white_bg=False` the random background
draw of tensorBase.py:380 stays a host decision per step: the step is captured once per outcome (data parallel: one
draw per global batch, shared by the ranks — see `bg_seed`).  Results of EARLIER eager training forwards of the same
model (`rgb`, the loss) must not be alive when the step is captured: they keep autograd's AccumulateGrad nodes bound
to the stream they ran on, the capture would record a dependency on that stream and HIP fails in
`hipStreamEndCapture` (torch warns "AccumulateGrad node's stream does not match")."""
import collections
import ctypes as C

import torch
import torch.distributed as dist

from . import _hip as H
from . import parallel
from .field import _stream


class GraphedTrainStep:
    # early_sort='auto': the step's two counting sorts run beside tf_shade_forward on a second stream when the
    # warm-up step produced at most this many (density, shaded) samples — measured: config 2 (236 k / 82 k) gains 8 %,
    # C4 (1.76 M / 470 k) loses 20 % because the sorts then outlast the shading kernel they hide behind
    EARLY_SORT_LIMITS = (500_000, 170_000)

    def __init__(self, model, optimizer, batch, n_samples, mask=None, ndc_ray=False, warmup=3, split=None,
                 early_sort='auto', white_bg=True, regularizers=False, bg_seed=20211202):
        self.model, self.opt = model, optimizer
        # split: capture backward and optimizer separately with the gradient all-reduce in between
        self.split = (dist.is_available() and dist.is_initialized() and
                      (dist.get_world_size() > 1 or parallel.FORCE_EXCHANGE)) if split is None else bool(split)
        self.graph_opt = None
        self._captured_for = None
        self._early = early_sort
        # data parallel: d loss / d rgb is pre-divided by the world size and the ranks' gradients are summed, which
        # equals averaging them without a second pass over the gradient buffer (1 / 2^k scales exactly)
        self._world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        dev = next(model.parameters()).device
        self.rays = torch.zeros(batch, 6, device=dev)
        self.target = torch.zeros(batch, 3, device=dev)
        # [0, batch): the step's sampling jitter; [batch]: the slot of the pinned report ring this step writes its sample
        # counts / overflow flag to (TfLive.slot) — both staged by one launch in front of every replay
        self._jitter_all = torch.zeros(batch + 1, device=dev)
        self.jitter = self._jitter_all[:batch]
        self._n_slots = 32
        self._ring = torch.zeros(self._n_slots, 4, dtype=torch.int32).pin_memory()
        self._ring_np = self._ring.numpy()
        self._pending = collections.deque()     # steps whose report has not been read yet
        self._step_no = 0
        self.loss = torch.zeros((), device=dev)
        self._grad_rgb = torch.zeros(batch, 3, device=dev)
        self._loss_state = torch.zeros(2, device=dev)      # tf_composite_forward_loss: running sum + arrival counter
        # the step's gradient buffer, kept across steps when the optimizer is a FusedAdam: its launch returns every gradient
        # it consumed to zero (consume_grads), so the backward accumulates into the same buffer again without a fill
        self._gstore = {'flat': None, 'clean': True}
        self._jit_ring, self._jit_i = None, 0              # pinned jitter buffers of _stage (ring)
        self._packjobs = {}                                # use_bg -> the captured forward's TfPackJob (run by _stage)
        self._dist = dist.is_available() and dist.is_initialized()
        self._items, self._items_d, self._items_r = {}, [], []      # data parallel: the buckets' (buffer, rows, width) pieces
        self.n_samples, self.ndc = n_samples, ndc_ray
        # FreeNeRF masks (train.py:303-318: a new mask dict every iteration with `free_reg`): the step keeps every mask
        # vector in ONE static device buffer the captured kernels read; set_mask() refreshes its values before a replay
        self.mask, self._mask_buf, self._mask_sig = None, None, None
        self._mask_ring, self._mask_ring_n, self._regw_ring = None, 0, None
        # white_bg=False (datasets without a white background, e.g. llff.py:141): the reference adds the white
        # background to a training batch with probability 1/2 (tensorBase.py:380) — a host decision per step, so the
        # step is captured once per outcome and the draw (same generator, same position in the stream) picks the graph
        self.white_bg = bool(white_bg)
        # ... data parallel: the reference has one process and one draw per batch.  Rule chosen here: one draw per GLOBAL
        # batch — every rank reads it from a generator all ranks seed alike (`bg_seed`), so the ranks replay the same
        # variant and their shards together are the batch a single process would have rendered; each rank still takes
        # the reference's draw from its default generator (discarded), which keeps that stream where train.py's loop has it
        self._bg_gen = torch.Generator().manual_seed(int(bg_seed)) if (self.split and not self.white_bg) else None
        self._graphs = {}            # use_bg -> (graph, graph_opt)
        # regularizers=True: tf_regularizers (train.py:340-371 in one pass) runs between the backward and the optimizer
        # with its four weights read from device memory — set_regularizer_weights() before each step follows the
        # schedule (the TV weights decay every iteration, train.py:336-339) without a new capture
        self._regw = torch.zeros(4, device=next(model.parameters()).device) if regularizers else None
        self._regw_host = None
        self._bg = True
        self.graph = None
        self._warm = max(1, warmup)   # >= 1: the first eager step also caches host copies of the geometry
        # PyTorch's whole-network capture recipe: warm-up iterations and the capture run on the same side
        # stream, so the autograd AccumulateGrad nodes are bound to the stream that is later captured
        self._side = torch.cuda.Stream(device=dev)
        if mask is not None:
            self.set_mask(mask)

    def _signature(self):
        """What a captured graph depends on besides the static buffers: the alpha mask object and the parameters'
        storage (both are replaced, not updated in place, by the schedule steps of train.py:300-311, 403-425)."""
        m = self.model
        lists = [getattr(m, n) for n in ("density_plane", "density_line", "app_plane", "app_line") if hasattr(m, n)]
        # the objects themselves, not their id(): CPython hands the address of a freed mask / optimizer to the next
        # allocation, and a stale graph replayed on a coincidence would read freed tables.  Compared with `is`.
        return (m.alphaMask, self.opt, self.n_samples) + tuple(p for lst in lists for p in lst)     # cheap: runs every step

    @staticmethod
    def _same(a, b):
        return a is not None and b is not None and len(a) == len(b) and all(
            (x == y) if isinstance(x, int) else (x is y) for x, y in zip(a, b))

    def _fwd_bwd(self):
        model = self.model
        keep, model.count_samples = model.count_samples, False     # nobody reads num_valid_samples here
        # loss = mean((rgb - target)^2) (train.py:334) and d loss / d rgb are formed by the compositing launch itself
        # (tf_composite_forward_loss): no launch of their own between the forward and the backward
        model._loss_fuse = self._fuse(1.0 / self._world if self.split else 1.0)
        model._grad_store = self._store()
        model._live_host_override = self._live_override()
        try:
            rgb, _, _ = model(self.rays, self.mask, white_bg=self.white_bg, is_train=True, ndc_ray=self.ndc,
                              N_samples=self.n_samples)
            model._loss_fuse = None
            self.opt.zero_grad(set_to_none=True)
            rgb.backward(self._grad_rgb)
        finally:
            model.count_samples = keep
            model._loss_fuse = model._grad_store = model._live_host_override = None

    def _store(self):
        return self._gstore if hasattr(self.opt, "consume_grads") else None

    def _live_override(self):
        """Where this step's compositing launch reports its sample counts and overflow flag: slot `jitter[batch]` of the
        pinned ring (a captured launch cannot change its arguments; the slot number is staged with the jitter)."""
        return (self._ring, self._jitter_all.data_ptr() + 4 * self.rays.shape[0], self._n_slots)

    def _fuse(self, grad_scale):
        """TfLossFuse of this step's static buffers (target, gradient, loss, the kernel's two state words)."""
        f = H.TfLossFuse()
        f.target, f.grad_scale = self.target.data_ptr(), float(grad_scale)
        f.grad, f.loss, f.state = self._grad_rgb.data_ptr(), self.loss.data_ptr(), self._loss_state.data_ptr()
        return f

    def _mask_rows(self, mask):
        """[(group, key, plane index | None, length, value)] for every mask vector of `mask`, in a fixed order; the value is
        whatever the reference indexes (`mask['decomp']['den'][i]`: a scalar or a (C_i,) vector; an encoding mask: a
        scalar or a vector over the encoding's columns) — broadcast to `length` when it is uploaded."""
        m = self.model
        cp = m._is_cp()
        rows = []
        for key, comps in (("den", m.density_n_comp), ("app", m.app_n_comp)):
            v = mask["decomp"][key]
            if v is not None:
                for i in range(1 if cp else 3):
                    rows.append(("decomp", key, i, int(comps[0] if cp else comps[i]), v[i]))
        enc_len = {"pos": 2 * 3 * m.pos_pe, "view": 2 * 3 * m.view_pe, "fea": 2 * m.app_dim * m.fea_pe}
        for key in ("pos", "view", "fea"):
            v = mask["encoding"].get(key)
            if v is not None:
                rows.append(("encoding", key, None, enc_len[key], v))
        return rows

    def set_mask(self, mask):
        """The FreeNeRF mask dict of the next step(s) (utils.get_free_mask, train.py:303-318), or None.  The values go
        into the step's static mask buffer (one small H2D copy from a fresh pinned tensor, ordered before the next
        replay); only a change of the dict's STRUCTURE (which entries are None) makes the step capture again."""
        if mask is None:
            if self.mask is not None:
                self.mask, self._mask_buf, self._mask_sig = None, None, None
                self._graphs, self._packjobs, self._items = {}, {}, {}
                self.graph = self.graph_opt = None
                self._warm = max(self._warm, 1)
            return
        rows = self._mask_rows(mask)
        sig = tuple(r[:4] for r in rows)
        dev = self.rays.device
        if sig != self._mask_sig:
            total = sum(r[3] for r in rows)
            self._mask_buf = torch.ones(max(total, 1), device=dev)
            static = {"encoding": {"pos": None, "view": None, "fea": None}, "decomp": {"den": None, "app": None}}
            off = 0
            for grp, key, i, n, _ in rows:
                view = self._mask_buf[off:off + n]
                off += n
                if grp == "decomp":
                    if static[grp][key] is None:
                        static[grp][key] = []
                    static[grp][key].append(view)
                else:
                    static[grp][key] = view
            had = self._mask_sig is not None or bool(self._graphs)
            self.mask, self._mask_sig = static, sig
            if had:     # other mask pointers than the captured ones: capture again
                self._graphs, self._packjobs, self._items = {}, {}, {}
                self.graph = self.graph_opt = None
                self._warm = max(self._warm, 1)
        if rows:
            host = torch.cat([torch.broadcast_to(torch.as_tensor(v, dtype=torch.float32).detach().cpu().reshape(-1), (n,))
                              for _, _, _, n, v in rows])
            if self._mask_ring is None or self._mask_ring_n < host.numel():
                self._mask_ring, self._mask_ring_n = H.PinnedRing(host.numel()), host.numel()
            self._mask_ring.upload(self._mask_buf, host)

    def set_regularizer_weights(self, ortho=0.0, l1=0.0, tv_density=0.0, tv_app=0.0):
        """Weights of the four regulariser terms for the next step(s) (needs regularizers=True)."""
        vals = [float(ortho), float(l1), float(tv_density), float(tv_app)]
        if hasattr(self.opt, "set_regularizer_activity"):     # terms that are on give their tensors a gradient in every step
            self.opt.set_regularizer_activity(*[v > 0 for v in vals])
        if vals != self._regw_host:
            if self._regw_ring is None:
                self._regw_ring = H.PinnedRing(4)
            self._regw_ring.upload(self._regw, vals)
            self._regw_host = vals

    # ---- data-parallel pieces (no autograd: the launches are issued directly, so the backward can be cut in two) -----
    def _named(self):
        named = list(self.model.named_parameters())
        return named

    def _fwd_density(self):
        """forward + loss gradient + the backward up to the point where the density gradients are final"""
        from .autograd import _early_sort, backward_launches
        model = self.model
        named = self._named()
        keep, model.count_samples = model.count_samples, False
        hook, model._density_grads_ready = getattr(model, "_density_grads_ready", None), None   # no collective in a capture
        model._loss_fuse = self._fuse(1.0 / self._world)
        model._grad_store = self._store()
        model._live_host_override = self._live_override()
        try:
            with torch.no_grad():
                c = model._run_forward(self.rays, self.mask, self.white_bg, True, self.ndc, self.n_samples, save_valid=True,
                                       after_march=lambda ws, field, shade: _early_sort(model, ws, field, shade, named))
            early = None
            if c.get('sorted_on') is not None:
                c['sorted_on'], early = c['sorted_on']
            grads = backward_launches(model, c, named, self._grad_rgb, early, "density")
        finally:
            model.count_samples = keep
            model._density_grads_ready = hook
            model._loss_fuse = model._grad_store = model._live_host_override = None
        for n, p in named:
            p.grad = grads[n]
        self._ctx = (c, named)

    def _shade_half(self):
        from .autograd import backward_launches
        c, named = self._ctx
        backward_launches(self.model, c, named, self._grad_rgb, None, "shade")

    def _regs_and_opt(self):
        if self._regw is not None:      # rank-invariant terms: added after the data gradients have been reduced
            from .regularizers import add_regularizer_grads_
            add_regularizer_grads_(self.model, 1.0, 1.0, 1.0, 1.0, weights_dev=self._regw)
        # (the buffer was handed out by this step's backward: not clean, and p.grad are views of it)
        consume = self._store() is not None and self._gstore['flat'] is not None and not self._gstore['clean']
        if consume:      # ... provided autograd kept the views it was handed (a cloned gradient would leave the buffer dirty:
            flat = self._gstore['flat']      # then it is simply zero-filled at the next hand-out)
            g = next((p.grad for p in self.model.parameters() if p.grad is not None), None)
            consume = g is not None and flat.data_ptr() <= g.data_ptr() < flat.data_ptr() + 4 * flat.numel()
        if not consume:
            self.opt.step()
            return
        keep, self.opt.consume_grads = self.opt.consume_grads, True
        try:
            self.opt.step()
        finally:
            self.opt.consume_grads = keep
        self._gstore['clean'] = True

    def _body(self):
        if self.split:      # the eager warm-up runs the very sequence the three graphs will replay
            self._part_a()
            work_d = parallel.bucket_reduce(self._items_d)
            self._part_b()
            work_r = parallel.bucket_reduce(self._items_r)
            for w in work_d + work_r:
                w.wait()
            self._part_c()
        else:
            self._fwd_bwd()
            self._regs_and_opt()

    # the three captured pieces of the data-parallel step: only the collectives themselves run between the replays — the
    # packing of the rows that travel (tf_gather_rows) closes graphs (a) and (b), their write-back opens graph (c)
    def _part_a(self):
        self._fwd_density()
        # direct-scatter mode (binned_scatter off, or more keys than the sort's tables hold): the density LINE gradients
        # still sit in their replicas when the density stage ends — tf_reduce_replicas folds them into the buffer only at
        # the end of the shading stage — so nothing may travel yet: one bucket ("all") after the whole backward
        self._one_bucket = self._ctx[0]['ws'].binned_cfg is None
        self._items_d = parallel.bucket_gather(self.model, "density") if (self._dist and not self._one_bucket) else []

    def _part_b(self):
        self._shade_half()
        self._items_r = parallel.bucket_gather(self.model, "all" if self._one_bucket else "rest") if self._dist else []

    def _part_c(self):
        parallel.bucket_writeback(self.model, self._items_d + self._items_r)    # (gradients are pre-divided by the world size)
        self._regs_and_opt()

    def _draw_jitter(self):
        """The step's sampling jitter, drawn from the CPU generator like the reference does (tensorBase.py:198-203), into a
        pinned buffer the staging launch reads in place.  The buffers go round a ring: a slot is rewritten only after the
        launch that read it has run (the host issues replays far ahead of the GPU)."""
        R = self.rays.shape[0]
        if self._jit_ring is None:      # R jitter values + the report slot of the step (_live_override)
            self._jit_ring = [[torch.zeros(R + 1).pin_memory(), None] for _ in range(32)]
        slot = self._jit_ring[self._jit_i % len(self._jit_ring)]
        self._jit_i += 1
        if slot[1] is not None:
            slot[1].synchronize()
        torch.rand(R, 1, out=slot[0][:R].view(R, 1))
        return slot

    def _stage(self, rays, target, ids, slot):
        """Batch, jitter and — once the step is captured — the forward's weight-pack job, in one launch where possible."""
        job = self._packjobs.get(self._bg) if self._graphs.get(self._bg) is not None else None
        R = self.rays.shape[0]
        if (ids is not None and rays.is_cuda and target.is_cuda and ids.is_cuda and ids.dtype == torch.int64
                and rays.dtype == torch.float32 and target.dtype == torch.float32 and rays.is_contiguous()
                and target.is_contiguous() and rays.shape[1:] == (6,) and target.shape[1:] == (3,)
                and target.shape[0] == rays.shape[0] and ids.numel() == R):
            # allrays[ray_idx], allrgbs[ray_idx] (train.py:297-298) straight into the static buffers
            H.check(H.lib().tf_gather_batch_staged(rays.data_ptr(), target.data_ptr(), rays.shape[0], ids.contiguous().data_ptr(),
                                                   R, self.rays.data_ptr(), self.target.data_ptr(), slot[0].data_ptr(),
                                                   self._jitter_all.data_ptr(), R + 1, C.byref(job) if job is not None else None,
                                                   _stream()), "tf_gather_batch_staged")
        else:
            if ids is None:
                self.rays.copy_(rays, non_blocking=True)
                self.target.copy_(target, non_blocking=True)
            else:
                torch.index_select(rays, 0, ids, out=self.rays)
                torch.index_select(target, 0, ids, out=self.target)
            H.check(H.lib().tf_gather_batch_staged(None, None, 0, None, 0, None, None, slot[0].data_ptr(), self._jitter_all.data_ptr(),
                                                   R + 1, C.byref(job) if job is not None else None, _stream()),
                    "tf_gather_batch_staged")
        if slot[1] is None:
            slot[1] = torch.cuda.Event()
        slot[1].record()

    def step(self, rays, target, ids=None):
        """One optimisation step on (rays, target) — or on rows `ids` of them; returns the (device) loss tensor."""
        lost = self._poll()
        if lost:
            self._recover(lost)
        if self._graphs and not self._same(self._signature(), self._captured_for):
            # the model changed under the graph (updateAlphaMask / shrink / upsample_volume_grid replace the mask and the
            # parameters, train.py:300-311): the captured pointers are stale -> warm up and capture again.  (Checked before
            # the staging launch, which runs the captured forward's weight-pack job.)
            self._drop_graphs()
        slot = self._draw_jitter()
        # random-background draw of tensorBase.py:380, taken after the jitter draw like the reference's forward does
        bg = True if self.white_bg else bool(torch.rand((1,)) < 0.5)
        if self._bg_gen is not None:
            bg = bool(torch.rand((1,), generator=self._bg_gen) < 0.5)
        return self._submit(rays, target, ids, slot, bg)

    def _drop_graphs(self):
        self._graphs = {}
        self._packjobs = {}
        self._items = {}
        self.graph = self.graph_opt = None
        self._warm = 1

    def _submit(self, rays, target, ids, slot, bg):
        """Stages and runs one step whose random draws (`slot`: pinned jitter, `bg`) are already made."""
        if len(self._pending) >= self._n_slots - 2:        # a report slot is reused only after its report has been read
            self._pending[0][6].synchronize()
            lost = self._poll()
            if lost:
                self._recover(lost)
        report = self._step_no % self._n_slots
        self._step_no += 1
        slot[0][self.rays.shape[0]] = float(report)
        seen = int(self._ring_np[report, 3])
        self._bg = bg
        self._stage(rays, target, ids, slot)
        self.model._bg_override = self._bg
        try:
            loss = self._step()
        finally:
            self.model._bg_override = None
        ev = torch.cuda.Event()
        ev.record()
        self._pending.append((report, seen, rays, target, ids, (slot, bg), ev))
        return loss

    def _poll(self):
        """Reads the reports that have come in (pinned memory, no synchronisation): returns the steps whose right-sized
        workspace overflowed — their gradients were incomplete and FusedAdam's gate left the parameters alone."""
        lost = []
        while self._pending:
            rec = self._pending[0]
            h = self._ring_np[rec[0]]
            if int(h[3]) == rec[1]:
                break                                   # this step has not reported yet (nor have the later ones)
            self._pending.popleft()
            if int(h[2]) != 0:
                lost.append((rec, int(h[0]), int(h[1])))
        return lost

    def _recover(self, lost):
        """Steps that overflowed their workspace are run again, in order, after the workspace has grown (steps that were
        enqueued behind them and fitted have been applied meanwhile: the batches commute up to that reordering)."""
        torch.cuda.synchronize()
        lost += self._poll()
        assert not self._pending
        ws = self.model.last['ws'] if self.model.last is not None else None
        if ws is not None:
            per = 1.15 / H.N_SHARDS
            self.model._grow_caps(ws.R, ws.N, max(l[2] for l in lost) * per, max(l[1] for l in lost) * per)
        self._drop_graphs()
        self.overflow_reruns = getattr(self, "overflow_reruns", 0) + len(lost)
        for rec, _, _ in lost:
            _, _, rays, target, ids, (slot, bg), _ = rec
            self._submit(rays, target, ids, slot, bg)

    def _step(self):
        if hasattr(self.opt, "sync_lr") and self.opt._lr_dev is not None:
            self.opt.sync_lr()                                    # FusedAdam: lr schedule follows the host values
        hit = self._graphs.get(self._bg)
        if hit is not None:
            if not self.split:
                hit[0].replay()
            else:
                self._replay_split(hit)
            self._after_replay()
            return self.loss
        self.model.static_jitter = self.jitter
        cur = torch.cuda.current_stream()
        if self._warm > 0:                                        # eager warm-up steps (also sizes the pools)
            self._warm -= 1
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side):
                self._body()
            cur.wait_stream(self._side)
            if self._early is not None:                           # decide once, from what this step actually sampled
                if self._early == 'auto':
                    torch.cuda.synchronize()
                    ws = self.model.last['ws']
                    n_app, n_den = (int(v) for v in ws.counters2d[:, :2].sum(0).tolist())
                    self._early = ws.binned_cfg is not None and n_den <= self.EARLY_SORT_LIMITS[0] \
                        and n_app <= self.EARLY_SORT_LIMITS[1]
                # the captured step cannot decide per replay: the mode is fixed here (a model left on 'auto' would follow
                # the host's last counts, which a replay does not update)
                want = bool(self._early)
                self.model.early_sort = want
                if want != (self.model.last.get('sorted_on') is not None):
                    self._warm = max(self._warm, 1)               # one eager step in the new mode before the capture
                self._early = None
            return self.loss
        torch.cuda.synchronize()
        self._captured_for = self._signature()
        pool = next(iter(self._graphs.values()))[0].pool() if self._graphs else None     # one pool for all variants
        g = torch.cuda.CUDAGraph()
        ext = self.model._pack_external = {}     # the forward hands its weight-pack job over instead of launching it
        try:
            if not self.split:
                with torch.cuda.graph(g, stream=self._side, pool=pool):
                    self._body()
        except BaseException:
            self.model._pack_external = None
            raise
        if not self.split:
            self.model._pack_external = None
            self._packjobs[self._bg] = ext.get('job')
            self._graphs[self._bg] = (g, None)
            self.graph = g
            self._run_pack_job()                                  # (this step was staged before its graph existed)
            g.replay()                                            # capture only records; run this step now
            self._after_replay()
            return self.loss
        # Three graphs: (a) forward + loss + backward until the density gradients are final, (b) shading backward +
        # appearance scatter, (c) regularisers + Adam — with the two buckets of the gradient exchange issued eagerly in
        # between, the density bucket travelling while (b) replays.  Thread-local capture mode: the process group's
        # helper threads (RCCL proxy / watchdog, gloo workers) may make HIP calls of their own while this thread
        # captures; they never touch the captured stream.
        try:
            with torch.cuda.graph(g, stream=self._side, pool=pool, capture_error_mode="thread_local"):
                self._part_a()
        finally:
            self.model._pack_external = None
        self._packjobs[self._bg] = ext.get('job')
        gb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gb, stream=self._side, pool=g.pool(), capture_error_mode="thread_local"):
            self._part_b()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=self._side, pool=g.pool(), capture_error_mode="thread_local"):
            self._part_c()
        self._items[self._bg] = (self._items_d, self._items_r)
        self._graphs[self._bg] = (g, g2, gb)
        self.graph, self.graph_opt = g, g2
        self._run_pack_job()
        self._replay_split(self._graphs[self._bg])
        self._after_replay()
        return self.loss

    def _run_pack_job(self):
        job = self._packjobs.get(self._bg)
        if job is not None:
            H.check(H.lib().tf_pack_matrices(C.byref(job), _stream()), "tf_pack_matrices")

    def _replay_split(self, graphs):
        ga, gopt, gb = graphs
        items_d, items_r = self._items[self._bg]
        ga.replay()
        work_d = parallel.bucket_reduce(items_d)                # on buffers of the graphs' pool: static addresses
        gb.replay()                                             # ... beside the density bucket's collective
        work_r = parallel.bucket_reduce(items_r)
        for w in work_d + work_r:
            w.wait()                                            # (the compute stream waits, not the host, under RCCL)
        gopt.replay()

    def _after_replay(self):
        """A replay runs Adam behind autograd's back: the parameters' `_version` counters (bumped once, at capture) no
        longer say that the weights changed, and the replay itself refreshed the padded weight copies BEFORE its Adam
        update — so they are one step behind the weights now.  Drop the cache tags: the next eager forward (validation
        between replays, compute_appfeature, the warm-up step after a schedule event) packs again."""
        self.model.invalidate_packed_weights()
