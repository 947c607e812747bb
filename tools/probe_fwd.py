"""Quick forward timing probe at BASELINE config 2 (300^3, N=1039, 4096-ray batches)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recon_amd
from recon_amd import synthetic as S

dev = "cuda:0"
torch.manual_seed(0)
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 300
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
aabb = torch.tensor(S.LEGO_AABB, device=dev)
model = recon_amd.TensorVMSplit(S.lego_args(), aabb, [grid] * 3, S.LEGO_NEAR_FAR, dev)
S.make_trained_like(model, recon_amd.AlphaGridMask)
N = min(int(1e6), recon_amd.cal_n_samples([grid] * 3, 0.5))
rays = S.blender_rays(1).to(dev)
g = torch.Generator().manual_seed(1)
perm = torch.randperm(rays.shape[0], generator=g)[: R * 8].to(dev)
batches = [rays[perm[i * R:(i + 1) * R]].contiguous() for i in range(8)]
print("N", N, "rays", rays.shape)
with torch.no_grad():
    for mode in ("random", "coherent"):
        bs = batches if mode == "random" else [rays[i * R:(i + 1) * R + 0].contiguous() for i in range(300, 308)]
        for _ in range(3):
            for b in bs:
                model(b, None, N_samples=N)
        torch.cuda.synchronize()
        t = time.perf_counter()
        iters = 10
        for _ in range(iters):
            for b in bs:
                out = model(b, None, N_samples=N)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / (iters * len(bs))
        ws = model.last["ws"]
        c = ws.counters2d[:, :3].sum(0).tolist()
        print(f"{mode}: {dt*1e6:.1f} us / {R}-ray batch = {R/dt/1e6:.2f} M rays/s | per ray: shaded {c[0]/R:.1f} density {c[1]/R:.1f} bbox {c[2]/R:.1f}")
