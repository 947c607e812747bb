"""Spread of the whole-schedule PSNR comparison (oracle/psnr_parity.run_full): how far identical runs of the HIP harness end
from each other (atomic summation order is the only difference between them), for a few schedule shapes.  ORACLE=1 adds two
oracle runs per shape.  Test infrastructure (uses the oracle)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recon_amd
from oracle import psnr_parity as P
from oracle import ref_torch as R
from recon_amd import harness

SHAPES = {
    "full400_reset": dict(n_iters=400, N_voxel_init=32 ** 3, upsamp_list=[220, 300], update_AlphaMask_list=[150]),
    "full400_noreset": dict(lr_upsample_reset=0, n_iters=400, N_voxel_init=32 ** 3, upsamp_list=[220, 300], update_AlphaMask_list=[150]),
    "full800_noreset": dict(lr_upsample_reset=0, N_voxel_init=32 ** 3),
    "full800_noreset_48": dict(lr_upsample_reset=0),
    "full800_reset_48": dict(),      # = oracle/psnr_parity.FULL_CFG, the shape the test uses
}
want = os.environ.get("SHAPES", ",".join(SHAPES)).split(",")
for name in want:
    c = dict(P.FULL_CFG)
    c.update(SHAPES[name])
    scene = P.Scene(recon_amd, "cuda:0", 64, c["n_iters"], 20, 100, c["batch_size"])
    g0 = round(c["N_voxel_init"] ** (1 / 3))
    init = scene.initial_state(g0, 5)
    hip = {}
    for graphed in (False, True):
        hip[graphed] = []
        for rep in range(3):
            m = scene.make_model(g0, 0)
            m.load_state_dict(init)
            torch.manual_seed(99)
            hist = harness.train(m, scene.rays_train, scene.gt_train, c, device="cuda:0", log_every=0, seed=1, graphed=graphed)
            with torch.no_grad():
                o = recon_amd.OctreeRender_trilinear_fast(scene.rays_test, m, chunk=4096, N_samples=hist["n_samples"][-1],
                                                          white_bg=True, device="cuda:0")[0]
            hip[graphed].append(round(P.psnr_db(torch.mean((o.clamp(0, 1) - scene.gt_test) ** 2)), 3))
    ora = []
    if int(os.environ.get("ORACLE", "0")):
        for rep in range(2):
            torch.manual_seed(99)
            fc, p, n_or, _ = P.oracle_train(scene, init, g0, c, seed=1)
            with torch.no_grad():
                out = R.render_chunked(fc, p, scene.rays_test, None, chunk=4096, n_samples=n_or, white_bg=True, device="cuda:0")[0]
            ora.append(round(P.psnr_db(torch.mean((out.clamp(0, 1) - scene.gt_test) ** 2)), 3))
    print(f"{name}: HIP eager {hip[False]}  captured {hip[True]}  oracle {ora}", flush=True)
