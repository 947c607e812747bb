"""Diagnostic: per-phase shader cycles of the shade kernel (TF_DIAG=1 build)."""
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
if os.environ.get("TF_FORWARD_CLASSIC"):
    lib.tf_shade_forward_variant(1)
if os.environ.get("TF_ABLATE"):      # timing only (results are wrong): 4 = no gather, 8 = no hidden-layer MFMAs, 16 = no basis / encodings
    lib.tf_debug_set_flags_fwd(int(os.environ["TF_ABLATE"]))
    print("ablation flags", os.environ["TF_ABLATE"])
out = (ctypes.c_ulonglong * 16)()
train = len(sys.argv) > 1 and sys.argv[1] == "train"      # training mode: the kernel also writes the saved rows
with torch.set_grad_enabled(train):
    for rep in range(2):
        for b in batches:
            r = model(b, None, N_samples=N, is_train=train)
            del r
        torch.cuda.synchronize()
        lib.tf_debug_phase_cycles(out, 1)
        if rep == 0:
            lib.tf_debug_phase_cycles_w4((ctypes.c_ulonglong * 16)(), 1)
print("mode:", "train (V, X, H1, H2 rows saved)" if train else "eval")
ntile = sum((int(c) + 63) // 64 for c in model.last["ws"].counters2d[:, 0].tolist())
print("tiles in last batch", ntile)
if os.environ.get("TF_FORWARD_CLASSIC"):
    names = ["info", "gather", "basis+view", "pe", "layer1", "layer2", "out", "locate"]
    tot = sum(out[i] for i in range(8))
    for i, n in enumerate(names):
        print(f"{n:12s} {out[i]/8/ntile:10.0f} cycles/tile  {100*out[i]/tot:5.1f}%")
else:       # the pipelined kernel: work and barrier wait of each phase, per crew (thread 0 / thread 512), per 64-sample chunk
    out4 = (ctypes.c_ulonglong * 16)()
    lib.tf_debug_phase_cycles_w4(out4, 1)
    names = [f"phase {k + 1} {w}" for k in range(5) for w in ("work", "wait")]
    print(f"{'':14s} {'MLP crew':>12s} {'front crew':>12s}   cycles / chunk")
    for i, n in enumerate(names):
        print(f"{n:14s} {out[i]/8/ntile:12.0f} {out4[i]/8/ntile:12.0f}")
    print(f"{'total':14s} {sum(out[:16])/8/ntile:12.0f} {sum(out4[:16])/8/ntile:12.0f}")
    nwg = 256
    print("prologue, cycles per workgroup (8 batches x %d workgroups): MLP crew %.0f, gather crew %.0f" % (nwg, out[13]/8/nwg, out4[13]/8/nwg))
    print("gather crew, phase 5: basis product + side tile %.0f (operand reads + MFMAs %.0f, side-tile writes %.0f), then locate %.0f"
          % ((out4[11] + out4[14] + out4[15])/8/ntile, out4[14]/8/ntile, out4[15]/8/ntile, out4[8]/8/ntile))
    print("loop top: MLP crew %.0f, gather crew %.0f" % (out[10]/8/ntile, out4[10]/8/ntile))
