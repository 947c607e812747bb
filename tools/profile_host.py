"""cProfile of the eager train step's host side (bench scene): where the Python time per step goes."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import recon_amd
sys.argv = [sys.argv[0]]
import bench
dev = torch.device("cuda", 0)
model, rays, targets, n_samples, reso, _ndc, _white = bench.build_scene(recon_amd, dev, bench.parse(), 0)
model.reference_none_grads = bool(int(os.environ.get("NONE_GRADS", "1")))
model.lazy_sample_count = True
opt = recon_amd.FusedAdam(model.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
perm = torch.randperm(rays.shape[0], device=dev)
def step(i):
    ids = perm[i * 4096:(i + 1) * 4096]
    rgb = recon_amd.OctreeRender_trilinear_fast(rays[ids], model, None, chunk=4096, N_samples=n_samples, white_bg=True,
                                               device=dev, is_train=True)[0]
    loss = torch.mean((rgb - targets[ids]) ** 2)
    opt.zero_grad(); loss.backward(); opt.step()
import gc; gc.collect(); gc.freeze()
for i in range(20): step(i)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(20, 220): step(i)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
st.print_callers("parameters")
