import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recon_amd
from tests._golden import Case
from tests.helpers import build_model
c = Case("vm_cubic_train"); dev = "cuda:0"
rays = c.rays.to(dev); target = torch.from_numpy(c.expect("grad/target")).to(dev)
m = build_model(recon_amd, c, dev)
jit = torch.rand(rays.shape[0]).to(dev)
m.static_jitter = jit
# eager reference
rgb_e, _, _ = m(rays, None, white_bg=True, is_train=True)
loss_e = torch.mean((rgb_e - target) ** 2); m.zero_grad(); loss_e.backward()
ge = {k: p.grad.clone() for k, p in m.named_parameters()}
rgb_e = rgb_e.detach().clone()
torch.cuda.synchronize()
# (A) forward only in a graph
g = torch.cuda.CUDAGraph()
with torch.no_grad():
    with torch.cuda.graph(g):
        rgb_g, _, _ = m(rays, None, white_bg=True, is_train=True)
g.replay(); torch.cuda.synchronize()
print("A fwd-only graph: max|drgb|", (rgb_g - rgb_e).abs().max().item())
# (B) forward + backward in a graph
for p in m.parameters(): p.grad = None
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    rgb2, _, _ = m(rays, None, white_bg=True, is_train=True)
    loss2 = torch.mean((rgb2 - target) ** 2)
    loss2.backward()
g2.replay(); torch.cuda.synchronize()
print("B fwd+bwd graph: loss", loss2.item(), "eager", loss_e.item(), "max|drgb|", (rgb2 - rgb_e).abs().max().item())
for k, p in m.named_parameters():
    d = (p.grad - ge[k]).abs().max().item()
    print(f"   {k:32s} grad diff {d:.3e} gradmax {ge[k].abs().max().item():.3e}")
g2.replay(); torch.cuda.synchronize()
print("B replay 2:")
for k, p in m.named_parameters():
    d = (p.grad - ge[k]).abs().max().item()
    if d > 1e-6: print(f"   {k:32s} grad diff {d:.3e} gradmax {ge[k].abs().max().item():.3e}")
