"""GPU parity of the HIP backward: d loss / d parameter for loss = mean((rgb_map - target)^2) against the
gradients autograd produced on the reference itself (golden fixtures)."""
import numpy as np
import pytest
import torch

from tests._golden import GRAD_CASES, Case
from tests.helpers import build_model

pytestmark = pytest.mark.gpu
GRAD_RTOL = 2e-4     # relative to the largest |gradient| of the tensor (float atomics reorder the sums)


# every case through the binned scatter (the default); three also through the direct scatter (per-tap atomics with
# line replicas), the path jobs beyond the sort's key limit fall back to
# ... and every case again with the sorts issued on the second stream during the forward (`early_sort = True`)
@pytest.mark.parametrize("name,binned", [(n, True) for n in GRAD_CASES] +
                         [("vm_cubic_train", False), ("vm_noncubic_relu", False), ("cp_train_mask", False)] +
                         [(n, "early") for n in GRAD_CASES])
def test_gradients_match_reference(recon, name, binned):
    c = Case(name)
    dev = "cuda:0"
    model = build_model(recon, c, dev)
    model.binned_scatter = bool(binned)
    model.early_sort = binned == "early"
    call = c.call
    torch.manual_seed(call["seed"])
    if call["ndc_ray"] and call["is_train"]:
        n = call["N_samples"] if call["N_samples"] > 0 else model.nSamples
        model._jitter_override = torch.rand(1, n)
    rgb, depth, nvalid = model(c.rays.to(dev), c.mask_to(dev), white_bg=call["white_bg"], is_train=call["is_train"],
                               ndc_ray=call["ndc_ray"], N_samples=call["N_samples"])
    assert rgb.requires_grad and not depth.requires_grad
    target = torch.from_numpy(c.expect("grad/target")).to(dev)
    loss = torch.mean((rgb - target) ** 2)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(c.expect("grad/loss"))) < 1e-6
    assert int(nvalid) == int(c.expect("out/num_valid_samples"))
    worst = {}
    for k, p in model.named_parameters():
        ref = c.expect("grad/" + k)
        got = p.grad.detach().cpu().numpy()
        assert got.shape == ref.shape, k
        scale = max(np.abs(ref).max(), 1e-12)
        err = np.abs(got - ref).max() / scale
        worst[k] = err
        assert err <= GRAD_RTOL, f"{k}: rel err {err:.3e} (scale {scale:.3e})"
    print(name, "worst rel grad err", max(worst.values()), max(worst, key=worst.get))


def test_adam_step_runs_on_channel_last_parameters(recon):
    """train.py:272-273, 374-376: Adam over get_optparam_groups, zero_grad / backward / step."""
    c = Case("vm_cubic_train")
    dev = "cuda:0"
    model = build_model(recon, c, dev)
    opt = torch.optim.Adam(model.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
    rays = c.rays.to(dev)
    target = torch.from_numpy(c.expect("grad/target")).to(dev)
    losses = []
    for it in range(5):
        torch.manual_seed(it)
        rgb, _, _ = model(rays, None, white_bg=True, is_train=True)
        loss = torch.mean((rgb - target) ** 2)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert recon.is_channel_last(model.density_plane[0]) and recon.is_channel_last(model.app_line[2])
    assert losses[-1] < losses[0], losses


def test_graphed_train_step_matches_eager(recon):
    """hipGraph-captured step (graph.py): replayed gradients == eager gradients on the same data and jitter.
    lr = 0 keeps the parameters fixed, so the comparison is not blurred by Adam amplifying rounding noise."""
    c = Case("vm_cubic_train")
    dev = "cuda:0"
    rays = c.rays.to(dev)
    target = torch.from_numpy(c.expect("grad/target")).to(dev)
    me, mg = build_model(recon, c, dev), build_model(recon, c, dev)
    oe, og = torch.optim.SGD(me.parameters(), lr=0.0), torch.optim.SGD(mg.parameters(), lr=0.0)
    gs = recon.GraphedTrainStep(mg, og, rays.shape[0], -1, warmup=1)
    for it in range(4):
        torch.manual_seed(100 + it)
        rgb, _, _ = me(rays, None, white_bg=True, is_train=True)
        loss = torch.mean((rgb - target) ** 2)
        oe.zero_grad()
        loss.backward()
        torch.manual_seed(100 + it)
        lg = gs.step(rays, target)
        torch.cuda.synchronize()
        assert abs(lg.item() - loss.item()) < 1e-6 * max(1.0, abs(loss.item()))
        for (k, a), (_, b) in zip(me.named_parameters(), mg.named_parameters()):
            scale = max(a.grad.abs().max().item(), 1e-12)
            assert (a.grad - b.grad).abs().max().item() <= 1e-4 * scale, (it, k)
    assert gs.graph is not None


def test_graphed_step_follows_per_iteration_free_masks(recon):
    """`free_reg: true` (configs/config.yaml:62; train.py:303-318 builds a new FreeNeRF mask dict every iteration): the
    captured step reads its masks from a static device buffer that set_mask() refreshes, so ONE capture serves the whole
    schedule.  Replayed loss and gradients == the eager step's with the same mask, at the start, the middle and the end
    of a 3000-iteration mask schedule (lr = 0: the parameters stay put)."""
    c = Case("vm_cubic_train")
    dev = "cuda:0"
    rays = c.rays.to(dev)
    target = torch.from_numpy(c.expect("grad/target")).to(dev)
    me, mg = build_model(recon, c, dev), build_model(recon, c, dev)
    oe, og = torch.optim.SGD(me.parameters(), lr=0.0), torch.optim.SGD(mg.parameters(), lr=0.0)

    def free_mask(step, device):
        return recon.get_free_mask(pos_bl=me.pos_bit_length, view_bl=me.view_bit_length, fea_bl=me.fea_bit_length,
                                   den_bl=me.density_n_comp, app_bl=me.app_n_comp, step=step, total_step=3000, device=device)

    gs = recon.GraphedTrainStep(mg, og, rays.shape[0], -1, warmup=1, mask=free_mask(0, "cpu"))
    captured = None
    losses = []
    for it, step in enumerate([0, 0, 0, 500, 1500, 2999, 3000]):
        torch.manual_seed(200 + it)
        rgb, _, _ = me(rays, free_mask(step, dev), white_bg=True, is_train=True)
        loss = torch.mean((rgb - target) ** 2)
        oe.zero_grad()
        loss.backward()
        torch.manual_seed(200 + it)
        gs.set_mask(free_mask(step, "cpu"))
        lg = gs.step(rays, target)
        torch.cuda.synchronize()
        losses.append(loss.item())
        assert abs(lg.item() - loss.item()) < 1e-6 * max(1.0, abs(loss.item())), (step, lg.item(), loss.item())
        for (k, a), (_, b) in zip(me.named_parameters(), mg.named_parameters()):
            scale = max(a.grad.abs().max().item(), 1e-12)
            assert (a.grad - b.grad).abs().max().item() <= 1e-4 * scale, (step, k)
        if gs.graph is not None and captured is None:
            captured = gs.graph
        assert captured is None or gs.graph is captured, "a new mask VALUE must not trigger a new capture"
    assert captured is not None
    assert abs(losses[3] - losses[2]) > 1e-7 and abs(losses[5] - losses[3]) > 1e-7      # the masks did change the result


def test_graphed_adam_training_reduces_loss(recon):
    c = Case("vm_cubic_train")
    dev = "cuda:0"
    model = build_model(recon, c, dev)
    opt = torch.optim.Adam(model.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), fused=True, capturable=True)
    gs = recon.GraphedTrainStep(model, opt, c.rays.shape[0], -1, warmup=2)
    rays, target = c.rays.to(dev), torch.from_numpy(c.expect("grad/target")).to(dev)
    torch.manual_seed(0)
    losses = [gs.step(rays, target).item() for _ in range(8)]
    assert gs.graph is not None and losses[-1] < losses[0], losses


def _same_trajectory(finals, init, losses):
    """Two runs of the same optimisation.  The packed sample order (atomic reservations) and with it every fp32
    summation order varies from run to run, and Adam turns the sign of a rounding-noise gradient into a full +-lr
    step, so single weights may differ: compare the loss curves, the strongly driven tensors element-wise
    against the distance moved, and the MLP weights in the L2 sense."""
    for a, b in zip(*losses):
        assert abs(a - b) <= 2e-3 * abs(a), losses
    for k in finals[0]:
        d0 = finals[0][k] - init[k]
        gap = finals[0][k] - finals[1][k]
        if k.startswith("renderModule"):
            assert gap.norm().item() <= 0.35 * d0.norm().item() + 1e-7, (k, gap.norm().item(), d0.norm().item())
        else:
            assert gap.abs().max().item() <= 0.05 * d0.abs().max().item() + 1e-7, k


@pytest.mark.gpu
def test_split_graph_step_equals_single_graph_step(recon):
    """The data-parallel form (backward graph | gradient all-reduce | optimizer graph) takes the same steps."""
    c = Case("vm_cubic_train")
    dev = "cuda:0"
    rays, target = c.rays.to(dev), torch.from_numpy(c.expect("grad/target")).to(dev)
    finals, losses = [], []
    for split in (False, True):
        model = build_model(recon, c, dev)
        init = {k: v.detach().clone() for k, v in model.state_dict().items()}
        opt = torch.optim.Adam(model.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), fused=True, capturable=True)
        gs = recon.GraphedTrainStep(model, opt, rays.shape[0], -1, warmup=1, split=split)
        torch.manual_seed(0)
        losses.append([gs.step(rays, target).item() for _ in range(6)])
        assert gs.graph is not None and (gs.graph_opt is not None) == split
        finals.append({k: v.detach().clone() for k, v in model.state_dict().items()})
    _same_trajectory(finals, init, losses)


@pytest.mark.gpu
def test_fused_adam_matches_torch_adam(recon):
    """tf_adam_step against torch.optim.Adam on the same gradients: dense, channel-last and odd-sized tensors, two
    parameter groups, a learning rate that decays every step (train.py:391-392)."""
    dev = "cuda:0"
    g = torch.Generator().manual_seed(3)
    shapes = [(1, 16, 37, 41), (1, 48, 33, 1), (27, 144), (128, 150), (128,), (3,), (5, 7)]
    def make():
        ps = []
        for i, s in enumerate(shapes):
            t = torch.randn(*s, generator=torch.Generator().manual_seed(10 + i)).to(dev)
            if len(s) == 4:
                t = recon.channel_last_param(t).data
            ps.append(torch.nn.Parameter(t))
        return ps
    pa, pb = make(), make()
    groups = lambda ps: [{'params': ps[:2], 'lr': 0.02}, {'params': ps[2:], 'lr': 1e-3}]
    oa = torch.optim.Adam(groups(pa), betas=(0.9, 0.99))
    ob = recon.FusedAdam(groups(pb), betas=(0.9, 0.99))
    for it in range(7):
        for a, b in zip(pa, pb):
            gr = torch.randn(a.shape, generator=g).to(dev) * (10.0 ** (it % 3 - 2))
            if it == 3:
                gr = gr * (torch.rand(a.shape, generator=g).to(dev) > 0.5)          # exact zeros
            a.grad = torch.empty_like(a).copy_(gr)          # empty_like keeps the (channel-last) strides
            b.grad = torch.empty_like(b).copy_(gr)
        oa.step()
        ob.step()
        for grp in oa.param_groups + ob.param_groups:
            grp['lr'] = grp['lr'] * 0.9
    for a, b, s in zip(pa, pb, shapes):
        assert a.stride() == b.stride()
        err = (a - b).abs().max().item()
        assert err <= 2e-6 * max(1.0, a.abs().max().item()), (s, err)
        sa, sb = oa.state[a], ob.state[b]
        for key in ('exp_avg', 'exp_avg_sq'):       # sums of terms of either sign: bound by the tensor's scale
            scale = sa[key].abs().max().item()
            assert (sa[key] - sb[key]).abs().max().item() <= 2e-6 * scale, (s, key)
    assert float(ob.state[pb[0]]['step']) == 7


@pytest.mark.gpu
def test_graphed_fused_adam_follows_lr_schedule(recon):
    """GraphedTrainStep + FusedAdam under a decaying learning rate takes the steps of the eager loop."""
    c = Case("vm_cubic_train")
    dev = "cuda:0"
    rays, target = c.rays.to(dev), torch.from_numpy(c.expect("grad/target")).to(dev)
    finals, losses = [], []
    for graphed in (False, True):
        model = build_model(recon, c, dev)
        init = {k: v.detach().clone() for k, v in model.state_dict().items()}
        opt = recon.FusedAdam(model.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
        gs = recon.GraphedTrainStep(model, opt, rays.shape[0], -1, warmup=1) if graphed else None
        torch.manual_seed(0)
        losses.append([])
        for it in range(6):
            if graphed:
                loss = gs.step(rays, target)
            else:
                rgb, _, _ = model(rays, None, white_bg=True, is_train=True)
                loss = torch.mean((rgb - target) ** 2)
                opt.zero_grad()
                loss.backward()
                opt.step()
            losses[-1].append(loss.item())
            for grp in opt.param_groups:
                grp['lr'] = grp['lr'] * 0.7
        assert not graphed or gs.graph is not None
        if graphed:     # the captured step accumulates into ONE gradient buffer that FusedAdam returns to zero
            flat = gs._gstore['flat']
            assert flat is not None and gs._gstore['clean'] and int(torch.count_nonzero(flat)) == 0
            assert all(p.grad is not None and flat.data_ptr() <= p.grad.data_ptr() < flat.data_ptr() + 4 * flat.numel()
                       for p in model.parameters())
            assert opt.consume_grads is False          # the optimizer's own setting is restored after every step
        finals.append({k: v.detach().clone() for k, v in model.state_dict().items()})
    _same_trajectory(finals, init, losses)


@pytest.mark.gpu
def test_fused_adam_untouched_regions_and_state_restore(recon):
    """Regions whose gradient stays zero for a while (texels no sample reaches yet): FusedAdam skips their moments
    (TfAdamJob.touched) and must still take torch.optim.Adam's steps once gradient arrives; then a state_dict round
    trip into a fresh optimizer continues the same trajectory."""
    dev = "cuda:0"
    torch.manual_seed(0)
    shape = (1, 16, 97, 53)
    pa = torch.nn.Parameter(recon.channel_last_param(torch.randn(*shape)).data.to(dev))
    pb = torch.nn.Parameter(pa.detach().clone())
    assert pa.stride() == pb.stride()
    oa = torch.optim.Adam([pa], lr=0.02, betas=(0.9, 0.99))
    ob = recon.FusedAdam([pb], lr=0.02, betas=(0.9, 0.99))
    g = torch.Generator().manual_seed(1)

    def grads(it):
        gr = torch.randn(*shape, generator=g).to(dev)
        live = torch.zeros(97, dtype=torch.bool, device=dev)
        live[: 10 + 20 * it] = True                     # rows (contiguous in the channel-last storage) wake up over time
        return gr * live[None, None, :, None]

    def run(n, oa, ob, first):
        for it in range(first, first + n):
            gr = grads(it)
            pa.grad = torch.empty_like(pa).copy_(gr)
            pb.grad = torch.empty_like(pb).copy_(gr)
            oa.step()
            ob.step()
    run(4, oa, ob, 0)
    assert (pa - pb).abs().max().item() <= 2e-6
    assert 0 < int((ob._touched != 0).sum()) and int((ob._touched == -1).sum()) < ob._touched.numel()
    ob2 = recon.FusedAdam([pb], lr=0.02, betas=(0.9, 0.99))
    ob2.load_state_dict(ob.state_dict())
    run(3, oa, ob2, 4)
    assert (pa - pb).abs().max().item() <= 3e-6
    assert float(ob2.state[pb]['step']) == 7
    assert (oa.state[pa]['exp_avg'] - ob2.state[pb]['exp_avg']).abs().max().item() <= 2e-6 * oa.state[pa]['exp_avg'].abs().max().item()


@pytest.mark.gpu
def test_fused_adam_returns_consumed_gradients_to_zero(recon):
    """FusedAdam.consume_grads (TfAdamJob.clear_grads): the update is the same, and the gradient buffer is all zeros when the
    launch ends — what lets GraphedTrainStep accumulate every step's gradients into one buffer without a fill."""
    dev = "cuda:0"
    torch.manual_seed(0)
    shape = (1, 16, 97, 53)          # 82,256 floats: ten whole 8192-chunks and a ragged tail
    pa = torch.nn.Parameter(recon.channel_last_param(torch.randn(*shape)).data.to(dev))
    pb = torch.nn.Parameter(pa.detach().clone())
    qa, qb = torch.nn.Parameter(torch.randn(131, device=dev)), None      # a second, odd-sized segment
    qb = torch.nn.Parameter(qa.detach().clone())
    oa = recon.FusedAdam([pa, qa], lr=0.02, betas=(0.9, 0.99))
    ob = recon.FusedAdam([pb, qb], lr=0.02, betas=(0.9, 0.99))
    ob.consume_grads = True
    g = torch.Generator().manual_seed(1)
    gb, hb = torch.zeros_like(pb), torch.zeros_like(qb)     # the consumer's buffers live across the steps
    pb.grad, qb.grad = gb, hb
    for it in range(4):
        gr = torch.randn(*shape, generator=g).to(dev)
        live = torch.zeros(97, dtype=torch.bool, device=dev)
        live[5 * it: 30 + 15 * it] = True
        gr = gr * live[None, None, :, None]
        hr = torch.randn(131, generator=g).to(dev)
        pa.grad, qa.grad = torch.empty_like(pa).copy_(gr), hr.clone()
        gb += gr                                            # accumulate into the (zero) buffers, as the backward does
        hb += hr
        oa.step()
        ob.step()
        assert int(torch.count_nonzero(gb)) == 0 and int(torch.count_nonzero(hb)) == 0
        assert int(torch.count_nonzero(pa.grad)) > 0        # the default leaves the gradients alone
        assert torch.equal(pa, pb) and torch.equal(qa, qb)


@pytest.mark.gpu
def test_graphed_step_with_ndc_rays(recon):
    """Forward-facing configuration (BASELINE config 4) through the captured step: the NDC jitter is a device-side
    draw (tensorBase.py:183-184), which must keep advancing under graph replay."""
    c = Case("vm_ndc_train")
    dev = "cuda:0"
    model = build_model(recon, c, dev)
    call = c.call
    n = call["N_samples"] if call["N_samples"] > 0 else model.nSamples
    opt = recon.FusedAdam(model.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
    rays, target = c.rays.to(dev), torch.from_numpy(c.expect("grad/target")).to(dev)
    gs = recon.GraphedTrainStep(model, opt, rays.shape[0], n, ndc_ray=True, warmup=2)
    torch.manual_seed(0)
    losses = [gs.step(rays, target).item() for _ in range(10)]
    assert gs.graph is not None and all(np.isfinite(losses))
    assert len(set(losses[3:])) == len(losses[3:]), losses      # every replay sees new jitter and new parameters
    assert min(losses[5:]) < losses[0], losses


@pytest.mark.gpu
def test_graphed_step_with_random_background(recon):
    """white_bg=False: the reference adds the white background to a training batch with probability 1/2
    (tensorBase.py:380).  The captured step keeps one graph per outcome and follows the same host draws as the eager loop."""
    c = Case("vm_cubic_train")
    dev = "cuda:0"
    rays, target = c.rays.to(dev), torch.from_numpy(c.expect("grad/target")).to(dev)
    finals, losses = [], []
    for graphed in (False, True):
        model = build_model(recon, c, dev)
        init = {k: v.detach().clone() for k, v in model.state_dict().items()}
        opt = recon.FusedAdam(model.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
        gs = recon.GraphedTrainStep(model, opt, rays.shape[0], -1, warmup=1, white_bg=False) if graphed else None
        torch.manual_seed(5)
        losses.append([])
        for it in range(14):
            if graphed:
                loss = gs.step(rays, target)
            else:
                rgb, _, _ = model(rays, None, white_bg=False, is_train=True)
                loss = torch.mean((rgb - target) ** 2)
                opt.zero_grad()
                loss.backward()
                opt.step()
            losses[-1].append(loss.item())
        if graphed:
            assert sorted(gs._graphs) == [False, True], list(gs._graphs)
        finals.append({k: v.detach().clone() for k, v in model.state_dict().items()})
    assert len(set(round(v, 4) for v in losses[0])) > 2          # the two backgrounds give visibly different losses
    _same_trajectory(finals, init, losses)


@pytest.mark.gpu
def test_binned_scatter_refuses_positions_outside_its_buffers(recon):
    """Round-1 faults came from a key histogram that was not zeroed before a graph replay: the fill kernel then wrote
    through offsets beyond `binned[]`.  The kernels now check every position against the buffer sizes in TfBinJob and
    report through the sticky status word (tf_bin_status) instead: poison the histogram, run a backward, expect the
    error — and no memory fault."""
    import ctypes as C
    from recon_amd import synthetic as S
    dev = "cuda:0"
    torch.manual_seed(0)
    aabb = torch.tensor(S.LEGO_AABB, device=dev)
    model = recon.TensorVMSplit(S.lego_args(density_n_comp=(8, 8, 8), app_n_comp=(16, 16, 16)), aabb, [32] * 3,
                                S.LEGO_NEAR_FAR, dev)
    S.make_trained_like(model, recon.AlphaGridMask, mask_res=32, radius=0.7)
    model.early_sort = False                           # the sorts run inside the backward, behind the poisoning below
    rays = S.blender_rays(1)[:2048].to(dev).contiguous()
    rgb, _, _ = model(rays, None, white_bg=True, is_train=True, N_samples=100)
    rgb.sum().backward()
    model.check_scatter_status()                       # a clean step raises nothing
    ws = model.last["ws"]
    model.zero_grad()
    rgb, _, _ = model(rays, None, white_bg=True, is_train=True, N_samples=100)
    ws = model.last["ws"]
    ws.hist_app.fill_(1 << 20)                         # what a skipped zero-fill leaves behind, exaggerated
    ws.hist_density.fill_(1 << 20)
    rgb.sum().backward()
    torch.cuda.synchronize()
    with pytest.raises(recon._hip.HipError, match="refused out-of-range"):
        model.check_scatter_status()


@pytest.mark.gpu
def test_compositing_launch_forms_loss_and_gradient(recon):
    """tf_composite_forward_loss (the captured step's form of train.py:334): the compositing launch also writes
    mean((rgb_map - target)^2) and its gradient.  Gradient: the same expression as tf_mse_grad, element for element;
    loss: the same sum in another order.  Run twice: the kernel re-arms its two state words itself."""
    import ctypes as C
    c = Case("vm_cubic_train")
    dev = "cuda:0"
    model = build_model(recon, c, dev)
    rays = c.rays.to(dev)
    target = torch.from_numpy(c.expect("grad/target")).to(dev)
    H = recon._hip
    grad = torch.zeros_like(target)
    loss, state = torch.full((), -1.0, device=dev), torch.zeros(2, device=dev)
    f = H.TfLossFuse()
    f.target, f.grad_scale, f.grad, f.loss, f.state = target.data_ptr(), 0.5, grad.data_ptr(), loss.data_ptr(), state.data_ptr()
    for rep in range(2):
        model._loss_fuse = f
        try:
            torch.manual_seed(c.call["seed"])
            rgb, _, _ = model(rays, None, white_bg=True, is_train=True, N_samples=c.call["N_samples"])
        finally:
            model._loss_fuse = None
        torch.cuda.synchronize()
        ref_loss = torch.mean((rgb.detach() - target) ** 2)
        ref_grad = torch.empty_like(target)
        ref_l = torch.zeros((), device=dev)
        H.check(H.lib().tf_mse_grad(rgb.detach().data_ptr(), target.data_ptr(), target.numel(), 0.5, ref_l.data_ptr(),
                                    ref_grad.data_ptr(), None), "tf_mse_grad")
        torch.cuda.synchronize()
        assert torch.equal(grad, ref_grad)
        assert abs(loss.item() - ref_loss.item()) <= 1e-6 * abs(ref_loss.item()) + 1e-12
        assert state.abs().max().item() == 0.0
        grad.zero_()
        loss.fill_(-1.0)
