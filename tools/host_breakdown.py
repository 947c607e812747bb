"""Where the host time of the eager train step goes (perf_counter around the pieces, no device syncs inside)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import recon_amd
sys.argv = [sys.argv[0]]
import bench
dev = torch.device("cuda", 0)
model, rays, targets, n_samples, reso, _ndc, _white = bench.build_scene(recon_amd, dev, bench.parse(), 0)
model.lazy_sample_count = True
model.reference_none_grads = bool(int(os.environ.get("NONE_GRADS", "1")))
opt = recon_amd.FusedAdam(model.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
perm = torch.randperm(rays.shape[0], device=dev)
import gc; gc.collect(); gc.freeze()
T = {}
def tick(name, t0):
    t1 = time.perf_counter(); T[name] = T.get(name, 0.0) + (t1 - t0); return t1
def step(i):
    t = time.perf_counter()
    ids = perm[i * 4096:(i + 1) * 4096]
    r, tg = rays[ids], targets[ids]
    t = tick("index", t)
    rgb = recon_amd.OctreeRender_trilinear_fast(r, model, None, chunk=4096, N_samples=n_samples, white_bg=True,
                                               device=dev, is_train=True)[0]
    t = tick("forward", t)
    loss = torch.mean((rgb - tg) ** 2)
    t = tick("loss", t)
    opt.zero_grad()
    t = tick("zero_grad", t)
    loss.backward()
    t = tick("backward", t)
    opt.step()
    t = tick("opt.step", t)
for i in range(20): step(i)
torch.cuda.synchronize(); T.clear()
t0 = time.perf_counter()
for i in range(20, 220): step(i)
issue = time.perf_counter() - t0
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f"issue {issue/200*1e3:.3f} ms/step, total {tot/200*1e3:.3f} ms/step")
for k, v in T.items(): print(f"  {k:10s} {v/200*1e3:.3f} ms")
