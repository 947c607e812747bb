"""Host-side helpers the hot path takes its inputs from (restated from the reference's utils.py).

`get_free_mask` reproduces the *observable* behaviour of utils.py:11-70 including its quirks (SURVEY
warning 7): `get_freq_reg_mask` returns from inside its loop, so every encoding mask is the FIRST element
of a clamped 1-D mask (a 0-dim tensor) and every decomposition "mask list" is one (n_comp[0],) vector that
the models index by plane id."""
import numpy as np
import torch


def N_to_reso(n_voxels, bbox):
    """utils.py:117-121."""
    xyz_min, xyz_max = bbox
    dim = len(xyz_min)
    voxel_size = ((xyz_max - xyz_min).prod() / n_voxels).pow(1 / dim)
    return ((xyz_max - xyz_min) / voxel_size).long().tolist()


def cal_n_samples(reso, step_ratio=0.5):
    """utils.py:124-125."""
    return int(np.linalg.norm(reso) / step_ratio)


def _first_freq_mask(lengths, current_iter, total_reg_iter, ratio, max_visible, device):
    """What utils.get_freq_reg_mask returns for the default (max_visible=None) schedule: the mask of the
    FIRST entry of `lengths` only (utils.py:14-28)."""
    if max_visible is not None:
        out = []
        for n in lengths:
            m = torch.zeros(n).to(device)
            m[: int(n * max_visible)] = 1.0
            out.append(m)
        return out
    n = lengths[0]
    if current_iter < total_reg_iter:
        m = torch.zeros(n).to(device)
        scaled = n * ratio
        ptr = scaled / 4 * current_iter / total_reg_iter + 1
        ptr = ptr if ptr < scaled / 4 else scaled / 4
        ip = int(ptr)
        m[: ip * 4] = 1.0
        m[ip * 4: ip * 4 + 4] = (ptr - ip)
        return torch.clamp(m, 1e-8, 1 - 1e-8)
    return torch.ones(n).to(device)


def get_free_mask(pos_bl=[0], view_bl=[0], fea_bl=[0], den_bl=[], app_bl=[], step=-1, total_step=1, ratio=1,
                  using_decomp_mask=True, max_visible=None, device='cpu'):
    """utils.py:38-70: {'encoding': {'pos','view','fea'}, 'decomp': {'den','app'}}."""
    pos_mask = view_mask = fea_mask = den_mask = app_mask = None
    if pos_bl[0] > 0:
        pos_mask = _first_freq_mask(pos_bl, step, total_step, ratio, max_visible, device)[0]
    if view_bl[0] > 0:
        view_mask = _first_freq_mask(view_bl, step, total_step, ratio, max_visible, device)[0]
    if fea_bl[0] > 0:
        fea_mask = _first_freq_mask(fea_bl, step, total_step, ratio, max_visible, device)[0]
    if using_decomp_mask:
        if len(den_bl) > 0:
            den_mask = _first_freq_mask(den_bl, step, total_step, ratio, max_visible, device)
        if len(app_bl) > 0:
            app_mask = _first_freq_mask(app_bl, step, total_step, ratio, max_visible, device)
    return {'encoding': {'pos': pos_mask, 'view': view_mask, 'fea': fea_mask},
            'decomp': {'den': den_mask, 'app': app_mask}}
