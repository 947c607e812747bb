"""Diagnostic (TF_DIAG=1 build): backward kernel time with plane / line atomics ablated."""
import os, sys
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
tgt = torch.rand(R, 3, device=dev)
for flags, label in ((0, "all atomics"), (1, "no plane atomics"), (2, "no line atomics"), (3, "no atomics")):
    lib.tf_debug_set_flags_march(flags)
    lib.tf_debug_set_flags_shade(flags)
    for rep in range(3):
        model.kernel_events = {} if rep == 2 else None
        for b in batches:
            rgb, _, _ = model(b, None, N_samples=N, is_train=True)
            loss = ((rgb - tgt) ** 2).mean()
            model.zero_grad()
            loss.backward()
        torch.cuda.synchronize()
    ev = model.kernel_events
    msg = "  ".join(f"{k[3:]}={sum(a.elapsed_time(b) for a, b in v) / len(v) * 1e3:.0f}us" for k, v in ev.items())
    print(f"{label:18s} {msg}")
