"""Regularisers on the factor tensors that the reference's training loop adds to the MSE (train.py:340-371).

`TVLoss` keeps the reference's call interface (`tvreg = TVLoss(); tensorf.TV_loss_density(tvreg)`, loss.py:120-141):
for a plane x of shape (B, C, H, W) it returns  weight * 2 * (sum (d_h x)^2 / (C (H-1) W) + sum (d_w x)^2 / (C H (W-1))) / B.
SURVEY row f-3: these run as plain torch ops on the channel-last parameters for now (not part of the
ray-marching kernels)."""
import torch


def total_variation(x: torch.Tensor) -> torch.Tensor:
    b, c, h, w = x.shape
    dh = torch.diff(x, dim=2)
    dw = torch.diff(x, dim=3)
    return 2.0 * (dh.square().sum() / (c * (h - 1) * w) + dw.square().sum() / (c * h * (w - 1))) / b


class TVLoss(torch.nn.Module):
    def __init__(self, TVLoss_weight=1):
        super().__init__()
        self.TVLoss_weight = TVLoss_weight

    def forward(self, x):
        return self.TVLoss_weight * total_variation(x)
