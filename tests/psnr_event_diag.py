#!/usr/bin/env python3
"""Where do the HIP student and the eager oracle student of oracle/psnr_parity.py part ways?  (VERDICT r2 item 1:
a reproducible +0.116 dB at 400 iterations that appears only with the schedule events in the run.)

Both students are trained in LOCK-STEP on the same scene, each on its own copy of the CPU-generator jitter stream, and
compared as they go:

  * every `--every` iterations: train loss of both, max |d param| / max |param| per tensor group;
  * at the alpha-mask event: differing voxels of the two occupancy volumes, both tight boxes;
  * at the up-sampling: max |d param| right before and right after the resize, N, learning rates, Adam moments;
  * test PSNR of both at every `--cuts` iteration.

Experiments (what is exchanged between the students at an event):
  --swap mask      the eager student continues with the HIP student's occupancy volume
  --swap params    at both events the eager student's parameters are overwritten with the HIP student's (trajectories
                   re-joined at the event: what remains afterwards is divergence grown since the event)
  --pair eager     no HIP student at all: TWO eager students, the second one's initial factor tensors nudged by
                   `--nudge` relative (default 1e-6 ~ the HIP path's per-step gradient noise): the spread the reference
                   arithmetic has against itself
  --pair hip       two HIP students, same nudge

Test infrastructure (imports the oracle); run on the GPU box:  python -m tests.psnr_event_diag --iters 400 ..."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch


def group_of(k):
    for g in ("density_plane", "density_line", "app_plane", "app_line", "basis_mat"):
        if k.startswith(g):
            return g
    return "mlp"


def param_distance(pa, pb):
    """per group: max |a - b| / max |a|"""
    out = {}
    for k in pa:
        if pa[k].shape != pb[k].shape:
            out[group_of(k)] = float("nan")
            continue
        d = float((pa[k] - pb[k]).abs().max())
        s = float(pa[k].abs().max())
        g = group_of(k)
        out[g] = max(out.get(g, 0.0), d / max(s, 1e-30))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=64)
    ap.add_argument("--init-grid", type=int, default=48)
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--mask-at", type=int, default=None)
    ap.add_argument("--upsample-at", type=int, default=None)
    ap.add_argument("--no-mask", action="store_true")
    ap.add_argument("--no-upsample", action="store_true")
    ap.add_argument("--every", type=int, default=25)
    ap.add_argument("--cuts", type=int, nargs="*", default=[])
    ap.add_argument("--swap", choices=["none", "mask", "params"], default="none")
    ap.add_argument("--pair", choices=["hip-eager", "eager", "hip"], default="hip-eager")
    ap.add_argument("--nudge", type=float, default=1e-6)
    ap.add_argument("--grad-check", type=int, default=0, help="for the first K iterations B restarts every iteration "
                    "from A's parameters and Adam moments; gradients and updated parameters are compared")
    ap.add_argument("--out", default=None)
    a = ap.parse_args(argv)

    import recon_amd
    from oracle import psnr_parity as P
    scene = P.Scene(recon_amd, "cuda:0", a.grid, a.iters)
    g0 = a.init_grid
    mask_at = (a.iters * 3 // 8) if a.mask_at is None else a.mask_at
    up_at = (a.iters * 5 // 8) if a.upsample_at is None else a.upsample_at
    init = scene.initial_state(g0, a.seed)
    nudged = {k: (v * (1.0 + a.nudge * torch.sign(torch.randn_like(v))) if ("_plane." in k or "_line." in k) else v.clone())
              for k, v in init.items()}
    if a.pair == "hip-eager":
        A, B = P.HipStudent(scene, g0, init), P.EagerStudent(scene, g0, init)
    elif a.pair == "eager":
        A, B = P.EagerStudent(scene, g0, init), P.EagerStudent(scene, g0, nudged)
    else:
        A, B = P.HipStudent(scene, g0, init), P.HipStudent(scene, g0, nudged)
    torch.manual_seed(99)
    rng = [torch.get_rng_state(), torch.get_rng_state()]
    log = {"args": vars(a), "mask_at": mask_at, "upsample_at": up_at, "trace": [], "events": [], "cuts": {}}
    cuts = set(a.cuts) | {a.iters}

    def both(fn):
        for i, s in enumerate((A, B)):
            torch.set_rng_state(rng[i])
            fn(s)
            rng[i] = torch.get_rng_state()

    for it in range(a.iters):
        if it < a.grad_check:
            B.load_params(A.params())
            if it > 0:
                B.load_moments(A.moments())
            both(lambda s: s.backward(it))
            ga, gb = A.grads(), B.grads()
            gd = {}
            for k in ga:
                g = group_of(k)
                ref = float(gb[k].abs().max())
                gd[g] = max(gd.get(g, 0.0), float((ga[k] - gb[k]).abs().max()) / max(ref, 1e-30))
            both(lambda s: s.update())
            rec = {"grad_check": it, "loss_a": float(A.loss), "loss_b": float(B.loss), "grad_rel": gd,
                   "after_update": param_distance(A.params(), B.params())}
            log["trace"].append(rec)
            print(json.dumps(rec), flush=True)
            continue
        both(lambda s: s.step(it))
        if it % a.every == 0 or it in (mask_at, mask_at + 1, up_at, up_at + 1, up_at + 2):
            rec = {"it": it, "loss_a": float(A.loss), "loss_b": float(B.loss), "dist": param_distance(A.params(), B.params())}
            log["trace"].append(rec)
            print(json.dumps(rec), flush=True)
        if it == mask_at and not a.no_mask:
            boxes = [s.mask_event((64, 64, 64)) for s in (A, B)]
            va, vb = A.alpha(), B.alpha()
            diff = (va != vb)
            ev = {"event": "alpha_mask", "it": it, "kept_a": int(va.sum()), "kept_b": int(vb.sum()),
                  "voxels_differ": int(diff.sum()), "box_a": boxes[0].tolist(), "box_b": boxes[1].tolist(),
                  "only_a": int((va & ~vb).sum()), "only_b": int((vb & ~va).sum())}
            if a.swap == "mask":
                B.set_alpha(va)
                ev["swapped"] = "B continues with A's volume"
            if a.swap == "params":
                B.load_params(A.params())
                B.set_alpha(va)
                ev["swapped"] = "B continues with A's parameters and volume (optimizer moments kept)"
            log["events"].append(ev)
            print(json.dumps(ev), flush=True)
        if it == up_at and not a.no_upsample:
            before = param_distance(A.params(), B.params())
            mom = {}
            ma, mb = A.moments(), B.moments()
            for k in ma:
                if k in mb:
                    g = group_of(k)
                    for j, nm in enumerate(("m", "v")):
                        d = float((ma[k][j] - mb[k][j]).abs().max()) / max(float(ma[k][j].abs().max()), 1e-30)
                        mom[g + "." + nm] = max(mom.get(g + "." + nm, 0.0), d)
            for s in (A, B):
                s.upsample_event(a.grid, it)
            if a.swap == "params":
                B.load_params(A.params())
            ev = {"event": "upsample", "it": it, "dist_before": before, "moments_before": mom,
                  "dist_after": param_distance(A.params(), B.params()), "N": [A.N, B.N],
                  "step": [A.step_size(), B.step_size()],
                  "lr_a": [g["lr"] for g in A.opt.param_groups], "lr_b": [g["lr"] for g in B.opt.param_groups]}
            log["events"].append(ev)
            print(json.dumps(ev), flush=True)
        if (it + 1) in cuts:
            pa, pb = A.test_psnr(), B.test_psnr()
            log["cuts"][it + 1] = {"a": pa, "b": pb, "delta_db": pa - pb}
            print(json.dumps({"cut": it + 1, A.name + "_a": pa, B.name + "_b": pb, "delta_db": pa - pb}), flush=True)
    A.finish()
    B.finish()
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(log, f, indent=1)
    return log


if __name__ == "__main__":
    main()
