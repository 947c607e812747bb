import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recon_amd
from recon_amd import synthetic as S
dev = "cuda:0"
def mk():
    torch.manual_seed(0)
    a = S.lego_args(); a["featureC"] = 64; a["fea_pe"] = 1
    m = recon_amd.TensorVMSplit(a, torch.tensor(S.LEGO_AABB, device=dev), [64] * 3, S.LEGO_NEAR_FAR, dev)
    S.make_trained_like(m, recon_amd.AlphaGridMask, mask_res=32)
    o = torch.optim.SGD(m.parameters(), lr=0.0)     # lr 0: parameters stay identical, only gradients are compared
    return m, o
rays = S.blender_rays(1, H=64, W=64).to(dev)[:2048].contiguous()
target = torch.rand(rays.shape[0], 3, device=dev)
me, oe = mk(); mg, og = mk()
gs = recon_amd.GraphedTrainStep(mg, og, rays.shape[0], -1, warmup=1)
for stepi in range(4):
    jit = torch.rand(rays.shape[0], 1)
    me._jitter_override = jit
    rgb, _, _ = me(rays, None, white_bg=True, is_train=True)
    loss = torch.mean((rgb - target) ** 2); oe.zero_grad(); loss.backward(); oe.step()
    gs._stage = lambda r, t: (gs.rays.copy_(r), gs.target.copy_(t), gs.jitter.copy_(jit.view(-1).to(dev)))
    lg = gs.step(rays, target)
    torch.cuda.synchronize()
    worst = 0.0; wk = ""
    for (k, a), (_, b) in zip(me.named_parameters(), mg.named_parameters()):
        gd = (a.grad - b.grad).abs().max().item() / max(a.grad.abs().max().item(), 1e-20)
        if gd > worst: worst, wk = gd, k
    print("step", stepi, "graph" if gs.graph is not None else "eager-warm", "loss eager %.9f graph %.9f" % (loss.item(), lg.item()), "worst rel grad diff %.3e (%s)" % (worst, wk))
