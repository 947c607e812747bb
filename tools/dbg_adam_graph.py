import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, recon_amd as recon
from _golden import Case
from helpers import build_model
c = Case("vm_cubic_train"); dev = "cuda:0"
rays, target = c.rays.to(dev), torch.from_numpy(c.expect("grad/target")).to(dev)
for graphed in (False, True):
    model = build_model(recon, c, dev)
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}
    opt = recon.FusedAdam(model.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
    gs = recon.GraphedTrainStep(model, opt, rays.shape[0], -1, warmup=1) if graphed else None
    torch.manual_seed(0)
    for it in range(6):
        if graphed:
            gs.step(rays, target)
        else:
            rgb, _, _ = model(rays, None, white_bg=True, is_train=True)
            loss = torch.mean((rgb - target) ** 2)
            opt.zero_grad(); loss.backward(); opt.step()
        for grp in opt.param_groups:
            grp['lr'] = grp['lr'] * 0.7
        torch.cuda.synchronize()
        print(graphed, it, "mlp0 moved", (model.state_dict()['renderModule.mlp.0.weight'] - init['renderModule.mlp.0.weight']).abs().max().item(),
              "basis moved", (model.state_dict()['basis_mat.weight'] - init['basis_mat.weight']).abs().max().item(),
              "lr_dev", opt._lr_dev.tolist(), "step", float(opt._step_dev))
