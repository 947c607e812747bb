"""GPU parity of the HIP backward: d loss / d parameter for loss = mean((rgb_map - target)^2) against the
gradients autograd produced on the reference itself (golden fixtures)."""
import numpy as np
import pytest
import torch

from tests._golden import GRAD_CASES, Case
from tests.helpers import build_model

pytestmark = pytest.mark.gpu
GRAD_RTOL = 2e-4     # relative to the largest |gradient| of the tensor (float atomics reorder the sums)


@pytest.mark.parametrize("name", GRAD_CASES)
def test_gradients_match_reference(recon, name):
    c = Case(name)
    dev = "cuda:0"
    model = build_model(recon, c, dev)
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
