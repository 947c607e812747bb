import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import recon_amd as recon
from recon_amd import synthetic as S
dev = "cuda:0"
model, rays_all, N, ndc, white = S.baseline_scene("C2_vm300", dev, views=1)
rays_all = rays_all.to(dev)
g = torch.Generator().manual_seed(1)
perm = torch.randperm(rays_all.shape[0], generator=g)[:4096 * 8].to(dev)
batches = [rays_all[perm[i * 4096:(i + 1) * 4096]].contiguous() for i in range(8)]
with torch.no_grad():
    targets = [recon.OctreeRender_trilinear_fast(b, model, chunk=4096, N_samples=N, white_bg=True, device=dev)[0] + 0.05 * torch.rand(4096, 3, device=dev) for b in batches]
opt = recon.FusedAdam(model.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
def step(i):
    def fwd_bwd():
        rgb = recon.OctreeRender_trilinear_fast(batches[i % 8], model, chunk=4096, N_samples=N, white_bg=True, is_train=True, device=dev)[0]
        loss = torch.mean((rgb - targets[i % 8]) ** 2)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        return loss
    model.retry_on_overflow(fwd_bwd)
    opt.step()
for early in (False, 'auto', False, 'auto', False, 'auto', False, 'auto'):
    model.early_sort = early
    for i in range(5): step(i)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(40): step(i)
    torch.cuda.synchronize()
    print("early_sort", early, "%.3f ms/step" % ((time.perf_counter() - t) / 40 * 1e3))
