"""Diagnostic: per-phase shader cycles of the binned scatter kernel (TF_DIAG=1 build)."""
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
perm = torch.randperm(rays.shape[0], generator=torch.Generator().manual_seed(1))[: R * 4].to(dev)
batches = [rays[perm[i * R:(i + 1) * R]].contiguous() for i in range(4)]
lib = recon_amd._hip.lib()
out = (ctypes.c_ulonglong * 16)()
tgt = torch.rand(R, 3, device=dev)
for rep in range(2):
    for b in batches:
        rgb, _, _ = model(b, None, N_samples=N, is_train=True)
        loss = ((rgb - tgt) ** 2).mean()
        model.zero_grad()
        loss.backward()
    torch.cuda.synchronize()
    lib.tf_debug_phase_cycles_bin(out, 1)
ws = model.last["ws"]
ints = ws.bin_ints.cpu()
nkeys = ws.binned_cfg[1]; nmax = max(ws.binned_cfg[0], ws.binned_cfg[1])
hist = ws.hist_app.cpu()[:nkeys]            # (left as counted: the scan kernel reads it, the next forward zeroes it)
choff = ints[2 * (nmax + 8): 2 * (nmax + 8) + nkeys + 1]
print("app job: nkeys", nkeys, "entries*6", int(hist.sum()), "nonempty keys", int((hist > 0).sum()), "max bin", int(hist.max()),
      "work items", int(choff[nkeys]))
names = ["search+setup", "zero/sync", "stage", "accumulate", "flush", "", "", "loop"]
tot = sum(out[i] for i in range(8))
for i, n in enumerate(names):
    if n:
        print(f"{n:14s} {out[i]/1e6:10.1f} Mcycles (thread0 sums over 8 launches) {100*out[i]/tot:5.1f}%")
