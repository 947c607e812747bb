"""MI355X-native TensoRF ray-marching hot path (import as `recon_amd`, see /recon_amd.py)."""
from . import _hip
from .field import (AlphaGridMask, MLPRender, MLPRender_Fea, MLPRender_PE, TensorBase, TensorCP, TensorVMSplit,
                    channel_last_param, is_channel_last)
from . import harness
from .graph import GraphedTrainStep
from .optim import FusedAdam
from .rays import generate_rays
from .regularizers import TVLoss, add_regularizer_grads_, fused_regularizers
from .renderer import OctreeRender_trilinear_fast
from .utils import N_to_reso, cal_n_samples, get_free_mask

__all__ = ["FusedAdam", "generate_rays", "fused_regularizers", "add_regularizer_grads_", "AlphaGridMask", "MLPRender", "MLPRender_Fea", "MLPRender_PE", "TensorBase", "TensorCP",
           "TensorVMSplit", "OctreeRender_trilinear_fast", "N_to_reso", "cal_n_samples", "get_free_mask",
           "channel_last_param", "is_channel_last", "GraphedTrainStep", "TVLoss", "harness", "_hip"]
