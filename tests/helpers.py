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


def oracle_of(model, device):
    """(FieldCfg, params) of the plain-PyTorch oracle describing the same field as a product model."""
    from oracle import ref_torch as R
    dev = torch.device(device)
    name = "TensorCP" if type(model).__name__ == "TensorCP" else "TensorVMSplit"
    cfg = R.FieldCfg(model=name, aabb=model.aabb.detach().to(dev), gridSize=model.gridSize.tolist(),
                     near_far=model.near_far, step_ratio=model.step_ratio, fea2denseAct=model.fea2denseAct,
                     density_n_comp=model.density_n_comp, app_n_comp=model.app_n_comp, app_dim=model.app_dim,
                     density_shift=model.density_shift, distance_scale=model.distance_scale,
                     shadingMode=model.shadingMode, pos_pe=model.pos_pe, view_pe=model.view_pe, fea_pe=model.fea_pe,
                     featureC=model.featureC).finalize()
    if model.alphaMask is not None:
        cfg.alpha_volume = model.alphaMask.alpha_volume[0, 0].to(dev)
        cfg.alpha_aabb = model.alphaMask.aabb.to(dev)
    params = {k: v.detach().to(dev).contiguous().clone() for k, v in model.state_dict().items()}
    return cfg, params
