"""Training loop with the reference's INTENDED schedule (train.py:296-483 minus the stray `return` at :447 and
with the undefined names bound to their config keys).

    for it in range(n_iters):
        batch  <- SimpleSampler permutation                                train.py:44-56, 297-298
        render <- renderer(rays, tensorf, mask, chunk=batch, N_samples=nSamples, is_train=True)   :323-333
        loss   <- MSE (+ ortho / L1 / TV regularisers with decaying weights)               :338-371
        backward, Adam step, lr *= lr_factor                                                :374-392
        it in update_AlphaMask_list -> updateAlphaMask; first time shrink, second time re-filter rays   :450-465
        it in upsamp_list -> N_to_reso(next N_voxel), upsample_volume_grid, rebuild Adam (lr reset)    :468-481

Host-side Python on the product API only (no oracle).  Data-parallel runs pass `rank` / `world` and get the
gradient all-reduce of parallel.py."""
import gc
import math

import numpy as np
import torch

from . import parallel
from .optim import FusedAdam
from .regularizers import TVLoss, add_regularizer_grads_, fused_supported
from .renderer import OctreeRender_trilinear_fast
from .utils import N_to_reso, cal_n_samples, get_free_mask


class SimpleSampler:
    """train.py:44-56."""

    def __init__(self, total, batch, seed=None, device=None):
        self.total, self.batch, self.curr, self.ids = total, batch, total, None
        self.rng = np.random.default_rng(seed) if seed is not None else None
        # device: the permutation is uploaded ONCE per epoch and sliced there — a 4096-index H2D copy per step (pageable
        # memory: a stream synchronisation) would tie the host to the GPU in every iteration of the captured loop
        self.device = device

    def nextids(self):
        self.curr += self.batch
        if self.curr + self.batch > self.total:
            perm = self.rng.permutation(self.total) if self.rng is not None else np.random.permutation(self.total)
            self.ids = torch.LongTensor(perm)
            if self.device is not None:
                self.ids = self.ids.to(self.device)
            self.curr = 0
        return self.ids[self.curr:self.curr + self.batch]


DEFAULTS = dict(n_iters=3000, batch_size=4096, lr_init=0.02, lr_basis=1e-3, lr_decay_iters=-1,
                lr_decay_target_ratio=0.1, lr_upsample_reset=1, N_voxel_init=128 ** 3, N_voxel_final=300 ** 3,
                upsamp_list=[2000, 3000, 4000, 5500, 7000], update_AlphaMask_list=[2000, 4000], step_ratio=0.5,
                alpha_mask_reso=None, Ortho_weight=0.0, L1_weight_inital=0.0, L1_weight_rest=0.0,
                TV_weight_density=0.0, TV_weight_app=0.0, free_reg=False, white_bg=True, ndc_ray=False,
                # samples per ray after an up-sampling: "reference" = train.py:472 `min(nSamples, cal_n_samples(...))`
                # (this fork: N never grows past the initial grid's count, 443 for 128^3); "upstream" = TensoRF's
                # `min(args.nSamples = 1e6, cal_n_samples(...))` (N follows the grid: 1039 at 300^3)
                n_samples_rule="reference")


def psnr(mse):
    return -10.0 * math.log(max(float(mse), 1e-12)) / math.log(10.0)     # loss.py:46-47


def train(tensorf, allrays, allrgbs, cfg=None, device="cuda", rank=0, world=1, log_every=0, seed=20211202, graphed=False):
    """Runs the schedule on `tensorf`; `allrays` (N,6) / `allrgbs` (N,3) may live on the CPU (as in the reference)
    or on the GPU.  Returns a history dict (loss / PSNR per step, events).

    graphed=True drives every iteration through `GraphedTrainStep` (forward, MSE, backward, regularisers, Adam replayed
    from a hipGraph; re-captured after each schedule event): same schedule, same host RNG stream, the rays are kept on
    the GPU; with `free_reg` the iteration's FreeNeRF masks are uploaded into the step's static mask buffer
    (GraphedTrainStep.set_mask)."""
    c = dict(DEFAULTS)
    c.update(cfg or {})
    aabb = tensorf.aabb
    n_iters, batch = c["n_iters"], c["batch_size"]
    upsamp_list, mask_list = list(c["upsamp_list"]), list(c["update_AlphaMask_list"])
    n_voxel_list = (torch.round(torch.exp(torch.linspace(math.log(c["N_voxel_init"]), math.log(c["N_voxel_final"]),
                                                        len(upsamp_list) + 1))).long()).tolist()[1:]   # train.py:209-215
    nSamples = min(int(1e6), cal_n_samples(tensorf.gridSize.tolist(), c["step_ratio"]))
    lr_factor = c["lr_decay_target_ratio"] ** (1 / (c["lr_decay_iters"] if c["lr_decay_iters"] > 0 else n_iters))
    def make_opt(lr_xyz, lr_net):     # train.py:272-273; "torch" selects torch.optim.Adam instead of tf_adam_step
        groups = tensorf.get_optparam_groups(lr_xyz, lr_net)
        if c.get("optimizer", "fused") == "torch":
            return torch.optim.Adam(groups, betas=(0.9, 0.99))
        return FusedAdam(groups, betas=(0.9, 0.99))

    opt = make_opt(c["lr_init"], c["lr_basis"])
    tvreg = TVLoss()
    ortho_w, l1_w = c["Ortho_weight"], c["L1_weight_inital"]
    tv_d, tv_a = c["TV_weight_density"], c["TV_weight_app"]
    if not c["ndc_ray"]:
        allrays, allrgbs = tensorf.filtering_rays(allrays, allrgbs, bbox_only=True)      # train.py:291
    hist = dict(loss=[], psnr=[], events=[], n_samples=[])
    # park the objects that exist now in the collector's permanent generation: at ~1 ms per step a full cyclic
    # collection over the set-up's long-lived objects (a few ms) would otherwise recur every handful of steps
    gs = None
    if graphed:
        from .graph import GraphedTrainStep
        if not fused_supported(tensorf) or c.get("optimizer", "fused") == "torch":
            raise ValueError("harness.train(graphed=True) needs the fused regularisers / FusedAdam")
        allrays, allrgbs = allrays.to(device).float().contiguous(), allrgbs.to(device).float().contiguous()
        gs = GraphedTrainStep(tensorf, opt, batch, nSamples, ndc_ray=c["ndc_ray"], white_bg=c["white_bg"], warmup=1,
                              regularizers=True)
    if c.get("fused_regularizers", True) and fused_supported(tensorf) and gs is None:
        # eager data parallel: the density gradients leave during the backward (only where nothing else touches .grad
        # before the exchange: the autograd regularisers of the other branch accumulate into it asynchronously)
        parallel.enable_overlapped_exchange(tensorf)
    # (rays on the GPU: the batch indices live there too, one upload per epoch)
    sampler_dev = allrays.device if allrays.is_cuda else None
    sampler = SimpleSampler(allrays.shape[0], batch * world, seed, device=sampler_dev)
    gc.collect()
    gc.freeze()
    mask = None
    for it in range(n_iters):
        ids = parallel.shard_ids(sampler.nextids(), rank, world).to(allrays.device)
        if tv_d > 0:
            tv_d *= lr_factor
        if tv_a > 0:
            tv_a *= lr_factor
        use_ortho = ortho_w if hasattr(tensorf, "vector_comp_diffs") else 0.0
        if c["free_reg"]:
            mask = get_free_mask(pos_bl=tensorf.pos_bit_length, view_bl=tensorf.view_bit_length,
                                 fea_bl=tensorf.fea_bit_length, den_bl=tensorf.density_n_comp,
                                 app_bl=tensorf.app_n_comp, step=it, total_step=n_iters,
                                 device="cpu" if gs is not None else device)           # train.py:303-318
        if gs is not None:       # the iteration below, replayed from a hipGraph (re-captured after schedule events)
            gs.opt, gs.n_samples = opt, nSamples
            if mask is not None:
                gs.set_mask(mask)
            gs.set_regularizer_weights(use_ortho, l1_w, max(tv_d, 0.0), max(tv_a, 0.0))
            loss = gs.step(allrays, allrgbs, ids)
        else:
            rays_train, rgb_train = allrays[ids], allrgbs[ids].to(device)

            def one_step():
                rgb_map, _, depth_map, _, _, n = OctreeRender_trilinear_fast(
                    rays_train, tensorf, mask, chunk=batch, N_samples=nSamples, white_bg=c["white_bg"], ndc_ray=c["ndc_ray"],
                    device=device, is_train=True)
                loss = torch.mean((rgb_map - rgb_train) ** 2)
                if hasattr(opt, "set_regularizer_activity"):    # (FusedAdam: these terms open the factor tensors' gates)
                    opt.set_regularizer_activity(use_ortho > 0, l1_w > 0, tv_d > 0, tv_a > 0)
                if c.get("fused_regularizers", True) and fused_supported(tensorf):
                    # train.py:340-371 in one pass over the factor tensors (tf_regularizers): the terms do not depend on the
                    # rays, so their gradient is added after the data gradients have been reduced across ranks
                    opt.zero_grad()
                    loss.backward()
                    parallel.finish_gradient_exchange(tensorf)
                    if max(use_ortho, l1_w, tv_d, tv_a) > 0:
                        add_regularizer_grads_(tensorf, use_ortho, l1_w, max(tv_d, 0.0), max(tv_a, 0.0))
                else:
                    total = loss
                    if use_ortho > 0:
                        total = total + use_ortho * tensorf.vector_comp_diffs()
                    if l1_w > 0:
                        total = total + l1_w * tensorf.density_L1()
                    if tv_d > 0:
                        total = total + tensorf.TV_loss_density(tvreg) * tv_d
                    if tv_a > 0:
                        total = total + tensorf.TV_loss_app(tvreg) * tv_a
                    opt.zero_grad()
                    total.backward()
                    parallel.allreduce_gradients(tensorf)
                return loss

            # (a batch that outgrows the right-sized workspace is repeated with its own jitter: field.retry_on_overflow)
            loss = tensorf.retry_on_overflow(one_step) if hasattr(tensorf, "retry_on_overflow") else one_step()
            opt.step()
        for g in opt.param_groups:
            g["lr"] = g["lr"] * lr_factor
        if log_every and (it % log_every == 0 or it == n_iters - 1):
            mse = float(parallel.allreduce_scalar(loss.detach()))
            hist["loss"].append((it, mse))
            hist["psnr"].append((it, psnr(mse)))
        if it in mask_list:                                                                # train.py:450-465
            g3 = tensorf.gridSize.tolist()
            reso_mask = c["alpha_mask_reso"] or (g3 if g3[0] * g3[1] * g3[2] < 256 ** 3 else [256, 256, 256])
            # train.py:456 hands the FreeNeRF density mask of this iteration to the alpha-volume rebuild
            new_aabb = tensorf.updateAlphaMask(tuple(reso_mask), mask['decomp']['den'] if mask is not None else None)
            if it == mask_list[0]:
                tensorf.shrink(new_aabb)
                l1_w = c["L1_weight_rest"]
                hist["events"].append((it, "shrink", tensorf.gridSize.tolist()))
            if not c["ndc_ray"] and it == mask_list[-1] and len(mask_list) > 1:
                allrays, allrgbs = tensorf.filtering_rays(allrays, allrgbs)
                sampler = SimpleSampler(allrays.shape[0], batch * world, seed + it, device=sampler_dev)
            # shrink replaced the parameters: the optimizer is rebuilt at the CURRENT learning rates (decayed it + 1 times)
            opt = make_opt(c["lr_init"] * lr_factor ** (it + 1), c["lr_basis"] * lr_factor ** (it + 1))
        if it in upsamp_list:                                                              # train.py:468-481
            n_voxels = n_voxel_list.pop(0)
            reso_cur = N_to_reso(n_voxels, tensorf.aabb)
            nSamples = min(nSamples if c["n_samples_rule"] == "reference" else int(1e6),
                           cal_n_samples(reso_cur, c["step_ratio"]))                      # train.py:472
            tensorf.upsample_volume_grid(reso_cur)
            scale = 1.0 if c["lr_upsample_reset"] else c["lr_decay_target_ratio"] ** (it / n_iters)
            opt = make_opt(c["lr_init"] * scale, c["lr_basis"] * scale)
            hist["events"].append((it, "upsample", reso_cur, nSamples))
        hist["n_samples"].append(nSamples)
    gc.unfreeze()
    if hasattr(tensorf, "check_scatter_status"):
        tensorf.check_scatter_status()     # the binned scatter's sticky error word: nothing was refused during the run
    return hist


@torch.no_grad()
def evaluate_psnr(tensorf, rays, rgbs, n_samples=-1, chunk=4096, white_bg=True, ndc_ray=False, device="cuda"):
    """loss.py:11-57 for one ray set: render, clamp, MSE -> PSNR."""
    rgb, _, _, _, _, _ = OctreeRender_trilinear_fast(rays, tensorf, chunk=chunk, N_samples=n_samples, ndc_ray=ndc_ray,
                                                     white_bg=white_bg, device=device)
    mse = torch.mean((rgb.clamp(0, 1) - rgbs.to(rgb.device)) ** 2)
    return psnr(mse)
