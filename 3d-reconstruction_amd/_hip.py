"""ctypes binding of libtensorf_hip.so (C ABI declared in include/tensorf_hip.h).

The library is the product: there is no CPU or eager fallback.  `lib()` raises when the shared object is
missing, and every entry point raises `HipError` on a non-zero hipError_t.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TF_DIAG=1 loads the instrumented build (phase timers; tools/probe_phases*.py), TF_DIAG=<name> the experiment build
# lib/libtensorf_hip_<name>.so (tools only; never used for reported numbers)
_diag = os.environ.get("TF_DIAG", "")
LIB_PATH = os.path.join(_HERE, "lib", "libtensorf_hip.so" if not _diag else
                        ("libtensorf_hip_diag.so" if _diag == "1" else f"libtensorf_hip_{_diag}.so"))

N_SHARDS = 64
SHARD_STRIDE = 32
MAX_SAMPLES = 8192
TILE = 64

MODEL_VM, MODEL_CP = 0, 1
ACT_SOFTPLUS, ACT_RELU = 0, 1
HEAD_MLP, HEAD_SH, HEAD_RGB = 0, 1, 2
SRC_FEAT, SRC_VIEW, SRC_PTS = 0, 1, 2

_fp = C.c_void_p


class TfFactors(C.Structure):
    _fields_ = [("plane", _fp * 3), ("line", _fp * 3), ("mask", _fp * 3), ("n_comp", C.c_int * 3)]


class TfFactorGrads(C.Structure):
    _fields_ = [("plane", _fp * 3), ("line", _fp * 3), ("n_rep", C.c_int), ("rep_stride", C.c_int)]


class TfField(C.Structure):
    _fields_ = [("model", C.c_int), ("act", C.c_int), ("grid", C.c_int * 3),
                ("aabb_lo", C.c_float * 3), ("aabb_hi", C.c_float * 3), ("inv_aabb", C.c_float * 3),
                ("near_", C.c_float), ("far_", C.c_float), ("step", C.c_float),
                ("distance_scale", C.c_float), ("density_shift", C.c_float), ("weight_thres", C.c_float),
                ("density", TfFactors),
                ("alpha_cells", _fp), ("alpha_grid", C.c_int * 3), ("alpha_lo", C.c_float * 3),
                ("alpha_inv", C.c_float * 3)]


class TfMarchIO(C.Structure):
    _fields_ = [("rays", _fp), ("n_rays", C.c_int), ("n_samples", C.c_int), ("ndc", C.c_int),
                ("jitter", _fp), ("z_table", _fp), ("save_valid", C.c_int), ("t_stop", C.c_float),
                ("acc", _fp), ("depth", _fp), ("app_offset", _fp), ("app_count", _fp), ("val_count", _fp),
                ("counters", _fp), ("app_ray", _fp), ("app_xyz", _fp), ("app_w", _fp),
                ("val_idx", _fp), ("val_feat", _fp), ("dbg_bbox_bits", _fp), ("dbg_valid_bits", _fp),
                ("dbg_app_bits", _fp), ("ent_xyz", _fp), ("ent_offset", _fp), ("dbg_z", _fp), ("seg_cap", C.c_int),
                ("ent_seg_cap", C.c_int)]


class TfPeBlock(C.Structure):
    _fields_ = [("src", C.c_int), ("freqs", C.c_int), ("mask", _fp)]


class TfShade(C.Structure):
    _fields_ = [("model", C.c_int), ("grid", C.c_int * 3), ("app", TfFactors), ("app_dim", C.c_int),
                ("n_app_total", C.c_int), ("head", C.c_int), ("basis", _fp), ("n_pe", C.c_int),
                ("pe", TfPeBlock * 3), ("in_c", C.c_int), ("feature_c", C.c_int),
                ("w1", _fp), ("b1", _fp), ("w2", _fp), ("b2", _fp), ("w3", _fp), ("b3", _fp),
                ("w1t", _fp), ("w2t", _fp)]


class TfShadeGrads(C.Structure):
    _fields_ = [("w1", _fp), ("b1", _fp), ("w2", _fp), ("b2", _fp), ("w3", _fp), ("b3", _fp),
                ("basis", _fp), ("app", TfFactorGrads), ("dv_out", _fp), ("wslab", _fp), ("direct_scatter", C.c_int),
                ("x_saved", _fp), ("h1_saved", _fp), ("h2_saved", _fp), ("rgb_fwd", _fp)]


class TfShadeSave(C.Structure):
    _fields_ = [("x", _fp), ("v", _fp), ("h1", _fp), ("h2", _fp)]


class TfBinJob(C.Structure):
    _fields_ = [("model", C.c_int), ("factors", TfFactors), ("grads", TfFactorGrads), ("grid", C.c_int * 3), ("counters", _fp),
                ("slot", C.c_int), ("seg_cap", C.c_int), ("xyz", _fp), ("grad", _fp), ("grad_ld", C.c_int),
                ("tile", C.c_int), ("bucket", C.c_int), ("chunk", C.c_int),
                ("hist", _fp), ("offsets", _fp), ("cursor", _fp), ("chunk_off", _fp), ("binned", _fp),
                ("nkeys", C.c_int), ("hist_zeroed", C.c_int), ("stage", C.c_int), ("binned_cap", C.c_int),
                ("items_cap", C.c_int), ("share_groups", C.c_int), ("status", _fp)]


class TfLossFuse(C.Structure):
    _fields_ = [("target", _fp), ("grad_scale", C.c_float), ("grad", _fp), ("loss", _fp), ("state", _fp)]


class TfCamera(C.Structure):
    _fields_ = [("height", C.c_int), ("width", C.c_int), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("c2w", C.c_float * 12), ("opengl", C.c_int), ("normalize", C.c_int), ("ndc", C.c_int),
                ("ndc_near", C.c_float)]


PACK_MAX = 8


class TfPackItem(C.Structure):
    _fields_ = [("src", _fp), ("dst", _fp), ("rows", C.c_int), ("cols", C.c_int), ("rows_pad", C.c_int),
                ("transpose", C.c_int)]


class TfPackJob(C.Structure):
    _fields_ = [("n", C.c_int), ("n_zero", C.c_int), ("item", TfPackItem * PACK_MAX), ("zero", _fp)]


class TfRegJob(C.Structure):
    _fields_ = [("density", TfFactors), ("app", TfFactors), ("density_grad", TfFactorGrads), ("app_grad", TfFactorGrads),
                ("grid", C.c_int * 3), ("w_ortho", C.c_float), ("w_l1", C.c_float), ("w_tv_density", C.c_float),
                ("w_tv_app", C.c_float), ("loss", _fp), ("scale", _fp), ("want_grad", C.c_int), ("weights_dev", _fp)]


ADAM_MAX_SEG, ADAM_CHUNK = 32, 8192
BIN_MAX_KEYS = 262144     # TF_BIN_MAX_KEYS: tf_binned_scatter rejects jobs with more keys


class TfAdamSeg(C.Structure):
    _fields_ = [("p", _fp), ("g", _fp), ("m", _fp), ("v", _fp), ("n", C.c_longlong), ("group", C.c_int), ("gate", C.c_int)]


class TfAdamJob(C.Structure):
    _fields_ = [("n_seg", C.c_int), ("clear_grads", C.c_int), ("seg", TfAdamSeg * ADAM_MAX_SEG),
                ("chunk_end", C.c_int * ADAM_MAX_SEG), ("lrs", _fp), ("step", _fp),
                ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("arrivals", _fp), ("touched", _fp),
                ("live", _fp), ("reg_active", _fp), ("skip_mask", C.c_uint), ("pad_", C.c_int)]


# TfAdamSeg.gate: which sample count opens a segment (TfLive) and which regulariser terms force it open
GATE_DENSITY, GATE_SHADED = 1, 2
REG_ORTHO, REG_L1, REG_TV_DENSITY, REG_TV_APP = 1 << 4, 2 << 4, 4 << 4, 8 << 4


class TfLive(C.Structure):
    _fields_ = [("dev", _fp), ("host", _fp), ("slot", _fp), ("n_slots", C.c_int), ("pad_", C.c_int)]


OVERFLOW_SLOT = 5


class HipError(RuntimeError):
    pass


class WorkspaceOverflow(HipError):
    """A training step produced more samples than its right-sized workspace holds: its gradients are incomplete (and
    FusedAdam's device-side gate would refuse them).  Raised from `loss.backward()`, AFTER the model has enlarged its
    workspace for the next call; `jitter` / `z_table` / `use_bg` are the step's random draws:
    `model.retry_on_overflow(step_fn)` runs the step again with them (harness.train and bench.py do)."""

    def __init__(self, msg, jitter=None, z_table=None, use_bg=None):
        super().__init__(msg)
        self.jitter, self.z_table, self.use_bg = jitter, z_table, use_bg


_lib = None

_SIGS = {
    "tf_pack_alpha_cells": [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp],
    "tf_pack_matrix": [_fp, C.c_int, C.c_int, _fp, C.c_int, _fp],
    "tf_pack_matrix_t": [_fp, C.c_int, C.c_int, _fp, C.c_int, _fp],
    "tf_pack_matrices": [C.POINTER(TfPackJob), _fp],
    "tf_gather_batch": [_fp, _fp, C.c_longlong, _fp, C.c_int, _fp, _fp, _fp],
    "tf_gather_batch_staged": [_fp, _fp, C.c_longlong, _fp, C.c_int, _fp, _fp, _fp, _fp, C.c_int, C.POINTER(TfPackJob), _fp],
    "tf_gather_rows": [_fp, _fp, C.c_int, C.c_int, _fp, _fp],
    "tf_scatter_rows": [_fp, _fp, C.c_int, C.c_int, _fp, _fp],
    "tf_mse_grad": [_fp, _fp, C.c_int, C.c_float, _fp, _fp, _fp],
    "tf_generate_rays": [C.POINTER(TfCamera), _fp, C.c_longlong, C.c_int, _fp, _fp],
    "tf_march_forward": [C.POINTER(TfField), C.POINTER(TfMarchIO), _fp],
    "tf_shade_forward": [C.POINTER(TfShade), _fp, C.c_int, _fp, C.c_int, _fp, _fp, _fp, C.c_int, C.POINTER(TfShadeSave), _fp],
    "tf_shade_forward_variant": [C.c_int],
    "tf_composite_forward": [C.c_int, _fp, _fp, _fp, _fp, _fp, C.c_int, _fp, _fp, _fp, _fp, C.POINTER(TfLive), _fp],
    "tf_composite_forward_loss": [C.c_int, _fp, _fp, _fp, _fp, _fp, C.c_int, _fp, _fp, _fp, _fp, C.POINTER(TfLossFuse),
                                  C.POINTER(TfLive), _fp],
    "tf_density_points": [C.POINTER(TfField), _fp, C.c_int, _fp, _fp],
    "tf_appfeature_points": [C.POINTER(TfShade), _fp, C.c_int, _fp, _fp],
    "tf_shade_points": [C.POINTER(TfShade), _fp, _fp, _fp, C.c_int, _fp, _fp],
    "tf_alpha_points": [C.POINTER(TfField), _fp, C.c_int, C.c_float, _fp, _fp],
    "tf_alpha_lattice": [C.POINTER(TfField), _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp, _fp],
    "tf_alpha_pool_threshold": [_fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp, _fp, _fp],
    "tf_sample_alpha_points": [_fp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float * 3), C.POINTER(C.c_float * 3), _fp,
                               C.c_int, _fp, _fp],
    "tf_filter_rays": [C.POINTER(TfField), _fp, C.c_int, C.c_int, C.c_int, _fp, _fp],
    "tf_reduce_replicas": [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp],
    "tf_march_backward": [C.POINTER(TfField), C.POINTER(TfMarchIO), _fp, _fp, C.c_int, _fp, _fp,
                          C.POINTER(TfFactorGrads), _fp, _fp, _fp],
    "tf_shade_backward_wslab_floats": [C.POINTER(TfShade)],
    "tf_shade_backward_supported": [C.POINTER(TfShade)],
    "tf_bin_nkeys": [C.c_int, C.POINTER(C.c_int * 3), C.POINTER(C.c_int * 3), C.c_int, C.c_int, C.c_int],
    "tf_bin_keys_per_entry": [C.c_int, C.POINTER(C.c_int * 3)],
    "tf_binned_scatter": [C.POINTER(TfBinJob), _fp],
    "tf_binned_sort_pair": [C.POINTER(TfBinJob), C.POINTER(TfBinJob), _fp],
    "tf_bin_status": [_fp, C.POINTER(C.c_int), _fp],
    "tf_shade_backward": [C.POINTER(TfShade), _fp, C.c_int, _fp, C.c_int, _fp, _fp, _fp,
                          C.POINTER(TfShadeGrads), _fp],
    "tf_adam_step": [C.POINTER(TfAdamJob), _fp],
    "tf_regularizers": [C.POINTER(TfRegJob), _fp],
}
EXPORTS = tuple(_SIGS) + ("tf_build_info",)


def lib():
    """Loads the HIP library once.  No fallback: a missing build is a hard error."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                           f"g.build()'` (or `make -C 3d-reconstruction_amd/csrc`). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        missing = [n for n in EXPORTS if not hasattr(L, n)]
        if missing:
            raise HipError(f"{LIB_PATH} does not export {missing}: stale build, rebuild it")
        for name, args in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_int
        L.tf_build_info.restype = C.c_char_p
        L.tf_shade_backward_wslab_floats.restype = C.c_size_t
        _lib = L
    return _lib


class PinnedRing:
    """Small host -> device uploads that recur every training step (learning rates, regulariser weights, FreeNeRF mask
    values): a ring of pinned staging buffers allocated ONCE.  `torch.tensor(v).pin_memory()` per upload goes through the
    pinned allocator every time — measured as the larger part of a millisecond of host time per iteration of the captured
    training loop once the rates decay every step (train.py:391-392).  A slot is rewritten only after the copy that read
    it has run (its event): the host may be many replays ahead of the GPU."""

    def __init__(self, n, depth=64):
        import torch
        self._torch = torch
        self._slots = [[torch.empty(n, dtype=torch.float32).pin_memory(), None] for _ in range(depth)]
        self._i = 0

    def upload(self, dst, values):
        torch = self._torch
        buf, ev = self._slots[self._i % len(self._slots)]
        if ev is not None:
            ev.synchronize()
        src = values if torch.is_tensor(values) else torch.tensor(values, dtype=torch.float32)
        buf[:src.numel()].copy_(src.reshape(-1))
        dst.reshape(-1)[:src.numel()].copy_(buf[:src.numel()], non_blocking=True)
        if ev is None:
            ev = self._slots[self._i % len(self._slots)][1] = torch.cuda.Event()
        ev.record()
        self._i += 1


def check(err, what):
    if err != 0:
        raise HipError(f"{what} failed with hipError_t {err}")


def ptr(t):
    """Device (or host) address of a tensor, None -> NULL."""
    return None if t is None else t.data_ptr()
