"""Fixture loading shared by the CPU (oracle) and GPU (HIP) parity tests."""
import json
import os

import numpy as np
import torch

from oracle import ref_torch as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = ["vm_cubic_eval", "vm_cubic_train", "vm_cubic_eval_nobg_n96", "vm_cubic_train_randbg",
         "vm_cubic_train_randbg2", "vm_cubic_mask_scalar", "vm_cubic_mask_vector", "vm_noncubic_relu",
         "vm_ndc_eval", "vm_ndc_train", "vm_tnt_inside", "cp_eval", "cp_train_mask", "vm_head_MLP",
         "vm_head_MLP_PE"]
GRAD_CASES = ["vm_cubic_train", "vm_cubic_mask_scalar", "vm_noncubic_relu", "vm_ndc_train", "cp_train_mask",
              "vm_head_MLP", "vm_head_MLP_PE"]


def _npz(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def unpack(bits, shape):
    n = int(np.prod(shape))
    return np.unpackbits(bits)[:n].reshape(shape).astype(bool)


class Case:
    def __init__(self, name):
        z = _npz(name)
        self.name = name
        self.raw = z
        self.cfg_d = json.loads(str(z["cfg_json"]))
        self.call = self.cfg_d["call"]
        src = _npz(self.cfg_d["state_from"]) if "state_from" in self.cfg_d else z
        self.state = {k[len("state/"):]: torch.from_numpy(src[k]) for k in src.files if k.startswith("state/")}
        self.rays = torch.from_numpy(z["rays"])
        self.shape = tuple(int(v) for v in z["mid/shape"])
        self.alpha_volume = torch.from_numpy(z["alpha_volume"]).float() if "alpha_volume" in z.files else None
        self.alpha_aabb = torch.from_numpy(z["alpha_aabb"]) if "alpha_aabb" in z.files else None
        self.mask = self._mask(z)

    @staticmethod
    def _mask(z):
        keys = [k for k in z.files if k.startswith("mask/")]
        if not keys:
            return None
        m = {"encoding": {"pos": None, "view": None, "fea": None}, "decomp": {"den": None, "app": None}}
        lists = {}
        for k in keys:
            parts = k.split("/")
            if len(parts) == 3:
                m[parts[1]][parts[2]] = torch.from_numpy(z[k])
            else:
                lists.setdefault((parts[1], parts[2]), {})[int(parts[3])] = torch.from_numpy(z[k])
        for (g, k), d in lists.items():
            m[g][k] = [d[i] for i in sorted(d)]
        return m

    def field_cfg(self, device="cpu"):
        c = self.cfg_d
        cfg = R.FieldCfg(model=c["model"], aabb=torch.tensor(c["aabb"], device=device), gridSize=c["gridSize"],
                         near_far=c["near_far"], step_ratio=c["step_ratio"], fea2denseAct=c["fea2denseAct"],
                         density_n_comp=c["density_n_comp"], app_n_comp=c["app_n_comp"], app_dim=c["app_dim"],
                         density_shift=c["density_shift"], distance_scale=c["distance_scale"],
                         shadingMode=c["shadingMode"], pos_pe=c["pos_pe"], view_pe=c["view_pe"],
                         fea_pe=c["fea_pe"], featureC=c["featureC"],
                         rayMarch_weight_thres=c["rayMarch_weight_thres"]).finalize()
        if self.alpha_volume is not None:
            cfg.alpha_volume = self.alpha_volume.to(device)
            cfg.alpha_aabb = self.alpha_aabb.to(device)
        return cfg

    def ctor_args(self):
        c = self.cfg_d
        return dict(step_ratio=c["step_ratio"], fea2denseAct=c["fea2denseAct"], density_n_comp=c["density_n_comp"],
                    app_n_comp=c["app_n_comp"], app_dim=c["app_dim"], density_shift=c["density_shift"],
                    distance_scale=c["distance_scale"], alphaMask_thres=c["alphaMask_thres"],
                    shadingMode=c["shadingMode"], pos_pe=c["pos_pe"], view_pe=c["view_pe"], fea_pe=c["fea_pe"],
                    featureC=c["featureC"])

    def mask_to(self, device):
        if self.mask is None:
            return None

        def mv(v):
            if v is None:
                return None
            if isinstance(v, list):
                return [x.to(device) for x in v]
            return v.to(device)
        return {g: {k: mv(v) for k, v in d.items()} for g, d in self.mask.items()}

    def expect(self, key):
        return self.raw[key]

    def expect_mask(self, key):
        return unpack(self.raw[key], self.shape)
