"""The optimizer-visible behaviour of the first iterations of a fresh field (train.py:272-273, 374-376), against the
reference's own 16-step run (tests/golden/adam_trajectory.npz): while nothing is shaded the appearance factors, basis
matrix and MLP have NO gradient in the reference (tensorBase.py:370) and torch.optim.Adam leaves them and their step
counts alone.  Three ways to drive the HIP path must all land on the reference's parameters:
  * the drop-in loop (HIP model + torch.optim.Adam): gradients are None where the reference's are;
  * HIP model + FusedAdam, eager;
  * GraphedTrainStep (hipGraph replay): the gates are device words, no host decision.
(The round-2 PSNR gap came from here: one step count for all parameters made the first appearance / MLP updates up to
26 % smaller than the reference's.)"""
import numpy as np
import pytest
import torch

from tests._golden import _npz

pytestmark = pytest.mark.gpu

ARGS = dict(step_ratio=0.5, fea2denseAct="softplus", density_n_comp=[8, 8, 8], app_n_comp=[8, 8, 8], app_dim=27,
            density_shift=-10.0, distance_scale=25.0, alphaMask_thres=0.001, shadingMode="MLP_Fea", pos_pe=2, view_pe=2,
            fea_pe=2, featureC=64)
CUBE = [[-1.5, -1.5, -1.5], [1.5, 1.5, 1.5]]


def _model(recon, z, dev="cuda:0"):
    m = recon.TensorVMSplit(ARGS, torch.tensor(CUBE, device=dev), [24, 24, 24], [2.0, 6.0], dev)
    m.load_state_dict({k[len("state0/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("state0/")})
    return m


def _check_final(model, z, names, steps_of):
    """Per-parameter step counts: exact.  Final parameters: against the distance each tensor TRAVELLED from its initial
    value (an error in the bias correction — the round-2 bug was 26 % on the first appearance / MLP update — shows there),
    plus a max-norm bound.  Not tighter: Adam turns gradients near zero (|g| ~ eps = 1e-8, e.g. MLP columns fed by
    encodings of still-tiny features) into steps of either sign, so a handful of elements differ by a fraction of
    lr x steps between any two implementations that round differently."""
    worst = (0.0, 0.0)
    for k, p in model.named_parameters():
        ref, init = z["final/" + k], z["state0/" + k]
        got = p.detach().cpu().numpy()
        assert float(steps_of(p)) == float(z["adam_step/" + k]), (k, float(steps_of(p)), float(z["adam_step/" + k]))
        travelled = float(np.linalg.norm(ref - init))
        l2 = float(np.linalg.norm(got - ref)) / max(travelled, 1e-12)
        mx = float(np.abs(got - ref).max() / np.abs(ref).max())
        print(f"  {k:28s} |d|_2 / travelled = {l2:.2e}   max |d| / max |ref| = {mx:.2e}")
        worst = (max(worst[0], l2), max(worst[1], mx))
        assert l2 <= 2e-2 and mx <= 1e-2, (k, l2, mx)
    return worst


@pytest.mark.parametrize("optimizer", ["torch", "fused"])
def test_eager_steps_follow_the_reference_trajectory(recon, optimizer):
    z = _npz("adam_trajectory")
    names = [str(n) for n in z["names"]]
    model = _model(recon, z)
    assert [k for k, _ in model.named_parameters()] == names
    groups = model.get_optparam_groups(0.02, 1e-3)
    opt = (torch.optim.Adam if optimizer == "torch" else recon.FusedAdam)(groups, betas=(0.9, 0.99))
    rays, target = torch.from_numpy(z["rays"]).cuda(), torch.from_numpy(z["target"]).cuda()
    for it in range(len(z["loss"])):
        torch.manual_seed(1000 + it)
        rgb, _, nv = model(rays, None, white_bg=True, is_train=True)
        loss = torch.mean((rgb - target) ** 2)
        opt.zero_grad()
        loss.backward()
        assert [int(p.grad is None) for _, p in model.named_parameters()] == z["grad_is_none"][it].tolist(), it
        # (a few samples sit on the `weight > 1e-4` threshold once the parameters carry five steps of rounding)
        assert abs(int(nv) - int(z["num_valid"][it])) <= max(2, 0.002 * int(z["num_valid"][it])), (it, int(nv))
        assert abs(loss.item() - float(z["loss"][it])) <= 2e-5 * max(1e-3, abs(float(z["loss"][it]))), it
        opt.step()
    worst = _check_final(model, z, names, lambda p: opt.state[p]["step"])
    print(optimizer, "worst final parameter error (of the tensor's max):", worst)


def test_fused_adam_gates_without_host_knowledge(recon):
    """reference_none_grads off: every gradient is a tensor (zeros where the reference has None), as inside a graph
    capture; FusedAdam must take the decision from the device-side sample counts alone."""
    z = _npz("adam_trajectory")
    names = [str(n) for n in z["names"]]
    model = _model(recon, z)
    model.reference_none_grads = False
    opt = recon.FusedAdam(model.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
    rays, target = torch.from_numpy(z["rays"]).cuda(), torch.from_numpy(z["target"]).cuda()
    for it in range(len(z["loss"])):
        torch.manual_seed(1000 + it)
        rgb, _, _ = model(rays, None, white_bg=True, is_train=True)
        loss = torch.mean((rgb - target) ** 2)
        opt.zero_grad()
        loss.backward()
        assert all(p.grad is not None for p in model.parameters())
        opt.step()
    _check_final(model, z, names, lambda p: opt.state[p]["step"])


def test_graphed_steps_follow_the_reference_trajectory(recon):
    z = _npz("adam_trajectory")
    names = [str(n) for n in z["names"]]
    model = _model(recon, z)
    opt = recon.FusedAdam(model.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
    rays, target = torch.from_numpy(z["rays"]).cuda(), torch.from_numpy(z["target"]).cuda()
    gs = recon.GraphedTrainStep(model, opt, rays.shape[0], int(model.nSamples), warmup=1)
    for it in range(len(z["loss"])):
        torch.manual_seed(1000 + it)
        loss = gs.step(rays, target)
        assert abs(float(loss) - float(z["loss"][it])) <= 2e-5 * max(1e-3, abs(float(z["loss"][it]))), it
    assert gs.graph is not None
    _check_final(model, z, names, lambda p: opt.state[p]["step"])
    model.check_scatter_status()


def test_a_regulariser_term_opens_the_gates_of_its_tensors(recon):
    """With e.g. the TV-appearance term on (train.py:360-371) the appearance PLANES have a gradient from the first step,
    shaded samples or not; the lines, basis and MLP still do not.  FusedAdam's gates follow `set_regularizer_activity`."""
    z = _npz("adam_trajectory")
    model = _model(recon, z)
    model.reference_none_grads = False
    opt = recon.FusedAdam(model.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
    opt.set_regularizer_activity(tv_app=True)
    rays, target = torch.from_numpy(z["rays"]).cuda(), torch.from_numpy(z["target"]).cuda()
    # (the steps before the first shaded sample: decided by the density factors alone, which the appearance term leaves alone)
    closed = int(z["grad_is_none"][:, [str(n) for n in z["names"]].index("basis_mat.weight")].sum())
    for it in range(len(z["loss"])):
        torch.manual_seed(1000 + it)
        rgb, _, _ = model(rays, None, white_bg=True, is_train=True)
        opt.zero_grad()
        torch.mean((rgb - target) ** 2).backward()
        recon.add_regularizer_grads_(model, 0.0, 0.0, 0.0, 1.0)
        opt.step()
    n = len(z["loss"])
    steps = {k: float(opt.state[p]["step"]) for k, p in model.named_parameters()}
    assert steps["app_plane.0"] == n and steps["density_plane.1"] == n
    assert steps["app_line.0"] == n - closed and steps["basis_mat.weight"] == n - closed
    assert steps["renderModule.mlp.0.weight"] == n - closed
