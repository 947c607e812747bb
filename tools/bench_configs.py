"""Timing of one 4096-ray step on the other BASELINE configurations (C1, C3, C4, C5; C2 is bench.py's workload)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import recon_amd as recon
from tests.test_full_size import _scene

DEV = "cuda:0"
import gc
for name in sys.argv[1:] or ["C1_vm128", "C2_vm300", "C3_cp300_sh", "C3_cp300_mlp", "C4_ndc", "C5_tt640"]:
    model, rays, N, ndc, white = _scene(recon, name)
    model.lazy_sample_count = True
    model.early_sort = bool(int(os.environ.get("EARLY", "0")))      # EARLY=1: sorts on a second stream during the forward
    gc.collect(); gc.freeze()
    target = torch.rand(rays.shape[0], 3, device=DEV)
    def ev():
        with torch.no_grad():
            model(rays, None, white_bg=white, is_train=False, ndc_ray=ndc, N_samples=N)
    for _ in range(25): ev()      # (also grows the allocator's pools: the first calls after a model switch stall in hipMalloc)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): ev()
    torch.cuda.synchronize(); t_eval = (time.perf_counter() - t0) / 30
    cs = model.last["ws"].counters2d[:, :3].sum(0).tolist()
    line = (f"{name:14s} grid {model.gridSize.tolist()} N={N} per ray: {cs[2]/4096:.0f} in-bbox {cs[1]/4096:.1f} density "
            f"{cs[0]/4096:.1f} shaded: eval {t_eval*1e3:.3f} ms = {4096/t_eval/1e6:.2f} M rays/s")
    if model.shadingMode not in ("SH", "RGB"):
        opt = recon.FusedAdam(model.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
        def tr():
            rgb, _, _ = model(rays, None, white_bg=True, is_train=True, ndc_ray=ndc, N_samples=N)
            loss = torch.mean((rgb - target) ** 2)
            opt.zero_grad(); loss.backward(); opt.step()
        for _ in range(5): tr()
        model.kernel_events = {}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): tr()
        torch.cuda.synchronize(); t_tr = (time.perf_counter() - t0) / 20
        ev_ = model.kernel_events; model.kernel_events = None
        ks = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in ev_.items()}
        line += f"; train (eager) {t_tr*1e3:.3f} ms = {4096/t_tr/1e6:.2f} M rays/s  " + " ".join(f"{k[3:]}={v*1e3:.0f}us" for k, v in ks.items())
    print(line, flush=True)
    del model
    torch.cuda.empty_cache()
