"""Study for a later round (CPU, no GPU needed): would a 3-term bf16 split of the shading GEMMs (A_hi B_hi + A_hi B_lo + A_lo B_hi,
fp32 accumulation: what three bf16 MFMAs per fp32 MFMA compute) stay inside the path's parity bars?  Runs the MLP_Fea head's
forward and backward on random but realistically scaled inputs in fp64 (truth), fp32 (today's kernels) and the split form."""
import torch

torch.manual_seed(0)
S, IN, FC = 4096, 150, 128


def split(x):
    hi = x.to(torch.bfloat16).to(torch.float32)
    lo = (x - hi).to(torch.bfloat16).to(torch.float32)
    return hi, lo


def mm_split(a, b):      # a @ b with both operands split; products exact in fp32 (bf16 x bf16), fp32 accumulation
    ah, al = split(a)
    bh, bl = split(b)
    return ah @ bh + (ah @ bl + al @ bh)


def mm_split4(a, b):     # with the lo x lo term as well
    ah, al = split(a)
    bh, bl = split(b)
    return ah @ bh + (ah @ bl + al @ bh) + al @ bl


def head(x, w1, b1, w2, b2, w3, b3, mm):
    h1 = torch.relu(mm(x, w1.t()) + b1)
    h2 = torch.relu(mm(h1, w2.t()) + b2)
    return torch.sigmoid(mm(h2, w3.t()) + b3), h1, h2


def backward(x, h1, h2, rgb, w1, w2, w3, g, mm):
    do = g * rgb * (1 - rgb)
    dw3 = mm(do.t(), h2)
    dz2 = mm(do, w3) * (h2 > 0)
    dw2 = mm(dz2.t(), h1)
    dz1 = mm(dz2, w2) * (h1 > 0)
    dw1 = mm(dz1.t(), x)
    dx = mm(dz1, w1)
    return dw1, dw2, dw3, dx


x = torch.cat([torch.randn(S, 27) * 0.5, torch.nn.functional.normalize(torch.randn(S, 3), dim=-1),
               torch.sin(torch.randn(S, 108) * 2), torch.cos(torch.randn(S, 12) * 2)], 1)
lin = [torch.nn.Linear(IN, FC), torch.nn.Linear(FC, FC), torch.nn.Linear(FC, 3)]
P = [t.detach() for l in lin for t in (l.weight, l.bias)]
g = torch.randn(S, 3) * 1e-4
P64 = [t.double() for t in P]
rgb64, h1_64, h2_64 = head(x.double(), *P64, torch.matmul)
ref = backward(x.double(), h1_64, h2_64, rgb64, P64[0], P64[2], P64[4], g.double(), torch.matmul)
for name, mm in (("fp32", torch.matmul), ("bf16 x3", mm_split), ("bf16 x4", mm_split4)):
    rgb, h1, h2 = head(x, *P, mm)
    flips = int(((h1 > 0) != (h1_64 > 0)).sum() + ((h2 > 0) != (h2_64 > 0)).sum())
    out = backward(x, h1, h2, rgb, P[0], P[2], P[4], g, mm)
    errs = [((a.double() - b).abs().max() / b.abs().max()).item() for a, b in zip(out, ref)]
    print(f"{name:8s} rgb max rel err {((rgb.double() - rgb64).abs() / rgb64.abs()).max().item():.2e}   "
          f"gradients (max err / tensor max) dW1 {errs[0]:.1e} dW2 {errs[1]:.1e} dW3 {errs[2]:.1e} dX {errs[3]:.1e}   "
          f"ReLU on/off differences vs fp64: {flips} of {2 * S * FC}")
