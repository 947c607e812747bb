import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, recon_amd as recon
from oracle import ref_torch as R
from tests.helpers import oracle_of
from tests.test_full_size import _scene
DEV = "cuda:0"
model, rays, N, ndc, white = _scene(recon, "C2_vm300")
if len(sys.argv) > 1 and sys.argv[1] == "direct":
    model.binned_scatter = False
target = torch.rand(rays.shape[0], 3, generator=torch.Generator().manual_seed(5)).to(DEV)
torch.manual_seed(3)
rgb, _, _ = model(rays, None, white_bg=True, is_train=True, ndc_ray=ndc, N_samples=N)
loss = torch.mean((rgb - target) ** 2); loss.backward()
cfg, params = oracle_of(model, DEV)
for p in params.values(): p.requires_grad_(True)
torch.manual_seed(3)
o_rgb, _, _ = R.render_rays(cfg, params, rays, None, white_bg=True, is_train=True, ndc_ray=ndc, n_samples=N)
o_loss = torch.mean((o_rgb - target) ** 2); o_loss.backward()
for k, p in model.named_parameters():
    g, og = p.grad, params[k].grad
    err = (g - og).abs()
    print(f"{k:28s} max|og|={og.abs().max().item():.3e} maxerr={err.max().item():.3e} rel={err.max().item()/og.abs().max().item():.2e} l2rel={(g-og).norm().item()/og.norm().item():.2e}")
    if "app_plane.0" in k or "density_plane.0" in k:
        idx = torch.nonzero(err > 0.2 * err.max())
        print("   worst entries (b,c,y,x):", idx[:12].tolist())
        for i in idx[:6]:
            b, c, y, x = i.tolist()
            print("     ", (c, y, x), "hip", g[b, c, y, x].item(), "oracle", og[b, c, y, x].item())
