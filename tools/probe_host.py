"""Host-side overhead of one training step (no GPU sync inside the loop): cProfile of 30 steps."""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recon_amd
from recon_amd import synthetic as S

dev = "cuda:0"
torch.manual_seed(0)
grid, R = 300, 4096
aabb = torch.tensor(S.LEGO_AABB, device=dev)
model = recon_amd.TensorVMSplit(S.lego_args(), aabb, [grid] * 3, S.LEGO_NEAR_FAR, dev)
S.make_trained_like(model, recon_amd.AlphaGridMask)
model.lazy_sample_count = True
N = recon_amd.cal_n_samples([grid] * 3, 0.5)
rays = S.blender_rays(1).to(dev)
perm = torch.randperm(rays.shape[0], generator=torch.Generator().manual_seed(1)).to(dev)
tgt = torch.rand(rays.shape[0], 3, device=dev)
opt = torch.optim.Adam(model.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), fused=True)
renderer = recon_amd.OctreeRender_trilinear_fast

def step(i):
    ids = perm[i * R:(i + 1) * R]
    rgb = renderer(rays[ids], model, None, chunk=R, N_samples=N, white_bg=True, device=dev, is_train=True)[0]
    loss = torch.mean((rgb - tgt[ids]) ** 2)
    opt.zero_grad()
    loss.backward()
    opt.step()

for i in range(5):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for i in range(5, 35):
    step(i)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue time per step {1e3*(t1-t0)/30:.3f} ms (with cProfile overhead); drain {1e3*(t2-t1):.2f} ms")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
