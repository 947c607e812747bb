#!/usr/bin/env python3
"""Child process of tests/test_parallel_gpu.py::test_split_step_over_rccl_in_a_one_rank_group.

The data-parallel step as `bench.py --gpus N` runs it — three hipGraph replays per step with the two gradient-bucket
all-reduces between them — over the REAL collective library: backend "nccl" (= RCCL), a one-rank group
(TF_DP_FORCE_EXCHANGE=1 makes the exchange run although nothing needs to travel).  This is what a 1-GPU box can show of
the RCCL-specific behaviour: asynchronous `work.wait()` stream semantics, collectives on buffers of the graphs' private
memory pool, `capture_error_mode="thread_local"` while RCCL's proxy thread is alive.  Runs in a FRESH process (never a
re-exec of one that has touched the GPU); prints one JSON line."""
import json
import os
import sys

os.environ["TF_DP_FORCE_EXCHANGE"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def main():
    port = sys.argv[1] if len(sys.argv) > 1 else "29533"
    os.environ["MASTER_PORT"] = port
    torch.cuda.set_device(0)
    saved = os.dup(1)
    os.dup2(2, 1)                    # RCCL's banner goes to stderr
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    warm = torch.zeros(1, device="cuda:0")
    dist.all_reduce(warm)
    torch.cuda.synchronize()
    import recon_amd as recon
    from recon_amd import parallel
    from tests._golden import Case
    from tests.helpers import build_model
    assert parallel.FORCE_EXCHANGE
    c = Case("vm_cubic_train")
    dev = "cuda:0"
    rays, target = c.rays.to(dev), torch.from_numpy(c.expect("grad/target")).to(dev)
    out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    finals, losses = [], []
    for split in (False, True):
        model = build_model(recon, c, dev)
        init = {k: v.detach().clone() for k, v in model.state_dict().items()}
        opt = recon.FusedAdam(model.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
        gs = recon.GraphedTrainStep(model, opt, rays.shape[0], -1, warmup=1, split=split, regularizers=True)
        gs.set_regularizer_weights(0.01, 8e-5, 0.01, 0.01)
        torch.manual_seed(0)
        losses.append([float(gs.step(rays, target)) for _ in range(8)])
        torch.cuda.synchronize()
        model.check_scatter_status()
        out["split_graphs" if split else "single_graph"] = bool(gs.graph is not None and (gs.graph_opt is not None) == split)
        if split:
            out["buckets"] = [len(gs._items_d), len(gs._items_r)]
            out["bucket_floats"] = [int(sum(t.numel() for t, _, _ in gs._items_d)), int(sum(t.numel() for t, _, _ in gs._items_r))]
        finals.append({k: v.detach().clone() for k, v in model.state_dict().items()})
        del gs, opt
    out["losses"] = losses
    worst_loss = max(abs(a - b) / abs(a) for a, b in zip(*losses))
    gaps = {}
    for k in finals[0]:
        d0 = finals[0][k] - init[k]
        gap = finals[0][k] - finals[1][k]
        if k.startswith("renderModule"):
            gaps[k] = float(gap.norm() / (d0.norm() + 1e-12))
        else:
            gaps[k] = float(gap.abs().max() / (d0.abs().max() + 1e-12))
    out["worst_loss_gap"] = worst_loss
    out["worst_param_gap_mlp"] = max(v for k, v in gaps.items() if k.startswith("renderModule"))
    out["worst_param_gap_other"] = max(v for k, v in gaps.items() if not k.startswith("renderModule"))
    out["finite"] = all(bool(torch.isfinite(v).all()) for v in finals[1].values())
    dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(saved, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
