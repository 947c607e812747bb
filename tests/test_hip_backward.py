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
