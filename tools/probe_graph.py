import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, copy
import numpy as np
import recon_amd
from tests._golden import Case
from tests.helpers import build_model
c = Case("vm_cubic_train"); dev = "cuda:0"
rays = c.rays.to(dev); target = torch.from_numpy(c.expect("grad/target")).to(dev)
def mk(capt):
    m = build_model(recon_amd, c, dev)
    o = torch.optim.Adam(m.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), fused=True, capturable=capt)
    return m, o
mg, og = mk(True); me, oe = mk(True)
gs = recon_amd.GraphedTrainStep(mg, og, rays.shape[0], -1, warmup=1)
for stepi in range(3):
    jit = torch.rand(rays.shape[0], 1)
    me._jitter_override = jit
    rgb, _, _ = me(rays, None, white_bg=True, is_train=True)
    loss = torch.mean((rgb - target) ** 2); oe.zero_grad(); loss.backward(); oe.step()
    gs._stage = lambda r, t: (gs.rays.copy_(r), gs.target.copy_(t), gs.jitter.copy_(jit.view(-1).to(dev)))
    lg = gs.step(rays, target)
    torch.cuda.synchronize()
    print("step", stepi, "graph" if gs.graph is not None else "eager-warm", "loss eager", loss.item(), "graph", lg.item())
    for (k, a), (_, b) in zip(me.named_parameters(), mg.named_parameters()):
        d = (a - b).abs().max().item()
        gd = (a.grad - b.grad).abs().max().item() if (a.grad is not None and b.grad is not None) else float("nan")
        if d > 1e-7 or gd > 1e-7:
            print(f"   {k:32s} max|dparam| {d:.3e}  grad diff {gd:.3e} gradmax {a.grad.abs().max().item():.3e}")
