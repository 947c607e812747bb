"""harness.train (intended schedule: MSE + regularisers, alpha-mask update + shrink, coarse-to-fine up-sampling) driven
eagerly and through GraphedTrainStep: wall time and end PSNR on a synthetic teacher scene."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import recon_amd as recon
from recon_amd import synthetic as S, harness

dev = "cuda:0"
aabb = torch.tensor(S.LEGO_AABB, device=dev)
args = S.lego_args()
torch.manual_seed(0)
teacher = recon.TensorVMSplit(args, aabb, [128] * 3, S.LEGO_NEAR_FAR, dev)
S.make_trained_like(teacher, recon.AlphaGridMask, mask_res=128)
rays = S.blender_rays(20, H=200, W=200, seed=3).to(dev)
with torch.no_grad():
    gt = recon.OctreeRender_trilinear_fast(rays, teacher, chunk=8192, white_bg=True, device=dev)[0]
cfg = dict(n_iters=1500, batch_size=4096, N_voxel_init=64 ** 3, N_voxel_final=160 ** 3, upsamp_list=[500, 800, 1100],
           update_AlphaMask_list=[400, 1000], TV_weight_density=0.01, TV_weight_app=0.01, L1_weight_inital=8e-5,
           L1_weight_rest=4e-5, Ortho_weight=0.01)
for graphed in (False, True):
    torch.manual_seed(11)
    student = recon.TensorVMSplit(args, aabb, recon.N_to_reso(64 ** 3, aabb), S.LEGO_NEAR_FAR, dev)
    torch.manual_seed(12)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hist = harness.train(student, rays, gt, cfg, device=dev, log_every=0, seed=1, graphed=graphed)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    p = harness.evaluate_psnr(student, rays[:40000], gt[:40000], device=dev)
    print(f"graphed={graphed}: {cfg['n_iters']} iterations in {dt:.2f} s = {dt / cfg['n_iters'] * 1e3:.3f} ms/iteration, "
          f"PSNR {p:.2f} dB, final grid {student.gridSize.tolist()}, events {[e[:2] for e in hist['events']]}")
