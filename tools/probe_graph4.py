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
    o = torch.optim.Adam(m.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), fused=True, capturable=True)
    return m, o
rays = S.blender_rays(1, H=64, W=64).to(dev)[:2048].contiguous()
target = torch.rand(rays.shape[0], 3, device=dev)
res = {}
for mode in ("eager", "graph"):
    m, o = mk()
    torch.manual_seed(3)
    out = []
    if mode == "graph":
        gs = recon_amd.GraphedTrainStep(m, o, rays.shape[0], -1, warmup=2)
        for it in range(6):
            out.append(gs.step(rays, target).item())
    else:
        for it in range(6):
            rgb, _, _ = m(rays, None, white_bg=True, is_train=True)
            loss = torch.mean((rgb - target) ** 2); o.zero_grad(); loss.backward(); o.step(); out.append(loss.item())
    res[mode] = out
    print(mode, out, "shaded", int(m.last["ws"].counters2d[:, 0].sum()) if mode == "eager" else "")
print("max rel diff", max(abs(a - b) / abs(b) for a, b in zip(res["graph"], res["eager"])))
