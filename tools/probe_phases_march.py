"""Diagnostic: per-phase shader cycles of the march forward kernel, summed over lane 0 of every ray's wave (TF_DIAG=1 build)."""
import os, sys, ctypes
os.environ["TF_DIAG"] = "1"
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
N = recon_amd.cal_n_samples([grid] * 3, 0.5)
rays = S.blender_rays(1).to(dev)
perm = torch.randperm(rays.shape[0], generator=torch.Generator().manual_seed(1))[: R * 8].to(dev)
batches = [rays[perm[i * R:(i + 1) * R]].contiguous() for i in range(8)]
lib = recon_amd._hip.lib()
out = (ctypes.c_ulonglong * 16)()
with torch.no_grad():
    for rep in range(2):
        for b in batches:
            model(b, None, N_samples=N)
        torch.cuda.synchronize()
        lib.tf_debug_phase_cycles_march(out, 1)
names = ["A validity walk", "B density gather", "C alpha / scan / queue", "epilogue (lists)"]
tot = sum(out[i] for i in range(4))
for i, n in enumerate(names):
    print(f"{n:24s} {out[i] / 8 / R:10.0f} cycles per ray  {100 * out[i] / tot:5.1f}%")
print(f"{'total':24s} {tot / 8 / R:10.0f} cycles per ray (mean over rays; the launch lasts as long as its slowest ray)")
