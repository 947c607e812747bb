"""Diagnostic: per-phase shader cycles of the shade backward kernel (TF_DIAG=1 build)."""
import os, sys, ctypes
os.environ.setdefault("TF_DIAG", "1")
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
perm = torch.randperm(rays.shape[0], generator=torch.Generator().manual_seed(1))[: R * 4].to(dev)
batches = [rays[perm[i * R:(i + 1) * R]].contiguous() for i in range(4)]
lib = recon_amd._hip.lib()
out = (ctypes.c_ulonglong * 16)()
out4 = (ctypes.c_ulonglong * 16)()
if len(sys.argv) > 1:
    lib.tf_debug_set_bwd_wgs(int(sys.argv[1]))
tgt = torch.rand(R, 3, device=dev)
for rep in range(2):
    for b in batches:
        rgb, _, _ = model(b, None, N_samples=N, is_train=True)
        loss = ((rgb - tgt) ** 2).mean()
        model.zero_grad()
        loss.backward()
    torch.cuda.synchronize()
    lib.tf_debug_phase_cycles_bwd(out, 1)
    lib.tf_debug_phase_cycles_bwd_w4(out4, 1)
names = ["P1 wait rows, dO", "-", "P3 dZ2 pass", "P4a dH1->dZ1", "P4b dW2", "P5a dW1", "P5b dX", "P6 dfeat,V->LDS",
         "P7a dV", "P7b dB", "P7 next-chunk requests"]
tot = sum(out[i] for i in range(len(names)))
ntile = sum((int(c) + 63) // 64 for c in model.last["ws"].counters2d[:, 0].tolist())
print("tiles in last batch", ntile)
for i, n in enumerate(names):
    print(f"{n:24s} wave 0: {out[i]/4/ntile:8.0f}   wave 4: {out4[i]/4/ntile:8.0f} cycles/chunk  {100*out[i]/tot:5.1f}%")
print(f"{'total':24s} wave 0: {tot/4/ntile:8.0f}")
