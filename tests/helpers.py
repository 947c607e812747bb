"""Shared helpers for the GPU parity tests: build a product model from a golden case."""
import torch


def build_model(recon, case, device):
    """Constructs recon.TensorVMSplit / TensorCP like train.py:227-247 does and loads the fixture state."""
    c = case.cfg_d
    aabb = torch.tensor(c["aabb"], device=device)
    cls = recon.TensorCP if c["model"] == "TensorCP" else recon.TensorVMSplit
    model = cls(case.ctor_args(), aabb, c["gridSize"], c["near_far"], device)
    missing = model.load_state_dict(case.state, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    if case.alpha_volume is not None:
        model.alphaMask = recon.AlphaGridMask(device, case.alpha_aabb.to(device), case.alpha_volume.to(device))
    return model
