"""cProfile of harness.train(graphed=True) on a trained-like 300^3 field (no schedule events): where the host time of
one iteration of the captured training loop goes."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import recon_amd as recon
from recon_amd import synthetic as S, harness

dev = "cuda:0"
aabb = torch.tensor(S.LEGO_AABB, device=dev)
torch.manual_seed(0)
model = recon.TensorVMSplit(S.lego_args(), aabb, [300] * 3, S.LEGO_NEAR_FAR, dev)
S.make_trained_like(model, recon.AlphaGridMask)
rays = S.blender_rays(3).to(dev)
with torch.no_grad():
    gt = recon.OctreeRender_trilinear_fast(rays, model, chunk=8192, white_bg=True, device=dev)[0]
gt = (gt + 0.05 * torch.randn_like(gt)).clamp(0, 1)
cfg = dict(n_iters=int(os.environ.get("ITERS", "600")), batch_size=4096, N_voxel_init=300 ** 3, N_voxel_final=300 ** 3, upsamp_list=[],
           update_AlphaMask_list=[], TV_weight_density=0.1, TV_weight_app=0.01, L1_weight_inital=8e-5, L1_weight_rest=4e-5,
           free_reg=bool(int(os.environ.get("FREE_REG", "0"))), lr_init=0.002, lr_basis=1e-4)
graphed = bool(int(os.environ.get("GRAPHED", "1")))
pr = cProfile.Profile()
torch.cuda.synchronize(); t0 = time.perf_counter()
pr.enable()
harness.train(model, rays, gt, cfg, device=dev, log_every=0, seed=1, graphed=graphed)
pr.disable()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"graphed={graphed}: {dt / cfg['n_iters'] * 1e3:.3f} ms per iteration (under the profiler)")
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
