import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, recon_amd as recon
from recon_amd import synthetic as S
dev = "cuda:0"
aabb = torch.tensor(S.LEGO_AABB, device=dev)
model = recon.TensorVMSplit(S.lego_args(), aabb, [300] * 3, S.LEGO_NEAR_FAR, dev)
tv = recon.TVLoss()
def step():
    total = 0.01 * model.vector_comp_diffs() + 8e-5 * model.density_L1() + 0.01 * model.TV_loss_density(tv) + 0.01 * model.TV_loss_app(tv)
    model.zero_grad()
    total.backward()
    return total
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print("eager regularisers (ortho + L1 + TV density + TV app, fwd+bwd) at 300^3: %.3f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))
w = dict(ortho_weight=0.01, l1_weight=8e-5, tv_weight_density=0.01, tv_weight_app=0.01)
def fstep():
    model.zero_grad()
    t = recon.fused_regularizers(model, **w)
    t.backward()
for _ in range(3): fstep()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): fstep()
torch.cuda.synchronize()
print("fused_regularizers (autograd entry, fwd+bwd): %.3f ms/step" % ((time.perf_counter() - t0) / 50 * 1e3))
for p in model.parameters(): p.grad = torch.zeros_like(p)
for _ in range(3): recon.add_regularizer_grads_(model, **w)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): recon.add_regularizer_grads_(model, **w)
torch.cuda.synchronize()
print("add_regularizer_grads_ (in place): %.3f ms/step" % ((time.perf_counter() - t0) / 50 * 1e3))
