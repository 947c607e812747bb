"""CPU, world_size 2 over gloo: ray sharding and the gradient all-reduce of the data-parallel path."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import recon_amd
    from recon_amd import parallel
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # 1. shards are disjoint and cover the batch
    ids = torch.arange(4096)
    mine = parallel.shard_ids(ids, rank, world)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    union = torch.cat(gathered).sort().values
    ok_shard = torch.equal(union, ids)
    # 2. gradient all-reduce (copy path: gradients are not the HIP backward's flat buffer on CPU)
    torch.manual_seed(0)
    model = recon_amd.TensorVMSplit(dict(step_ratio=0.5, fea2denseAct="softplus", density_n_comp=[4, 4, 4],
                                         app_n_comp=[8, 8, 8], app_dim=27, density_shift=-10.0, distance_scale=25.0,
                                         alphaMask_thres=0.001, shadingMode="MLP_Fea", pos_pe=2, view_pe=2, fea_pe=2,
                                         featureC=64),
                                    torch.tensor([[-1.5] * 3, [1.5] * 3]), [12, 12, 12], [2.0, 6.0], "cpu")
    g = torch.Generator().manual_seed(100 + rank)
    local = {}
    for k, p in model.named_parameters():
        grad = torch.randn(p.shape, generator=g)
        if p.dim() == 4:   # channel-last like the parameter
            buf = torch.empty(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2)
            buf.copy_(grad)
            grad = buf
        p.grad = grad
        local[k] = grad.clone()
    parallel.allreduce_gradients(model)
    expect = {}
    for k in local:
        parts = [torch.empty_like(local[k].contiguous()) for _ in range(world)]
        dist.all_gather(parts, local[k].contiguous())
        expect[k] = sum(parts) / world
    ok_grad = all(torch.allclose(p.grad, expect[k], atol=1e-6) for k, p in model.named_parameters())
    # 3. the flat-buffer path: grads are views of model.grad_flat -> one collective, no copies
    flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
    model2 = torch.nn.Linear(2, 3)
    model2.weight.grad = flat[:6].view(3, 2)
    model2.bias.grad = flat[6:9]
    model2.grad_flat = flat
    parallel.allreduce_gradients(model2)
    ok_flat = torch.allclose(flat, torch.arange(10, dtype=torch.float32) * (1 + 2) / 2)
    # 4. support-limited exchange: with an alpha mask only the plane rows that can receive gradient travel; the
    #    result must equal the full all-reduce (gradients are exactly zero outside the support on every rank)
    torch.manual_seed(0)
    model = recon_amd.TensorVMSplit(dict(step_ratio=0.5, fea2denseAct="softplus", density_n_comp=[4, 4, 4],
                                         app_n_comp=[8, 8, 8], app_dim=27, density_shift=-10.0, distance_scale=25.0,
                                         alphaMask_thres=0.001, shadingMode="MLP_Fea", pos_pe=2, view_pe=2, fea_pe=2,
                                         featureC=64),
                                    torch.tensor([[-1.5] * 3, [1.5] * 3]), [40, 44, 36], [2.0, 6.0], "cpu")
    vol = torch.zeros(17, 17, 17)
    vol[6:9, 5:8, 7:11] = 1.0
    model.alphaMask = recon_amd.AlphaGridMask("cpu", model.aabb, vol)
    named = sorted(model.named_parameters(), key=lambda kv: 0 if "_line." in kv[0] else 1)
    offs, total = {}, 0
    for k, p in named:
        offs[k] = total
        total += (p.numel() + 63) // 64 * 64
    flat3 = torch.zeros(total)
    model.grad_flat, model.grad_layout = flat3, (offs, total)
    for k, p in named:
        chunk = flat3[offs[k]:offs[k] + p.numel()]
        if p.dim() == 4:
            b, c, h, w = p.shape
            p.grad = chunk.view(b, h, w, c).permute(0, 3, 1, 2)
        else:
            p.grad = chunk.view(p.shape)
    segs = parallel.gradient_support(model)
    inside = torch.zeros(total, dtype=torch.bool)
    for a, b in (segs or [(0, total)]):
        inside[a:b] = True
    g3 = torch.Generator().manual_seed(200 + rank)
    flat3.copy_(torch.randn(total, generator=g3) * inside)
    mine3 = flat3.clone()
    parallel.allreduce_gradients(model, use_support="blocks")
    parts = [torch.empty_like(mine3) for _ in range(world)]
    dist.all_gather(parts, mine3)
    ok_support = segs is not None and float(inside.float().mean()) < 0.9 and torch.allclose(flat3, sum(parts) / world, atol=1e-6)
    # 5. the cell-level support (the default): a subset of the row blocks, same result where the gradients live
    rows = parallel.gradient_support_rows(model)
    ok_cells = rows is not None
    if ok_cells:
        w, idx = rows
        in_rows = torch.zeros(total // w, dtype=torch.bool)
        in_rows[idx] = True
        in_cells = in_rows[:, None].expand(-1, w).reshape(-1)
        p0, p1 = offs["density_plane.0"], offs["app_plane.2"] + model.app_plane[2].numel()
        ok_cells = bool((in_cells <= inside)[p0:p1].all()) and float(in_cells.float().mean()) < 0.7 * float(inside.float().mean())
        flat3.copy_(torch.randn(total, generator=g3) * in_cells)
        mine5 = flat3.clone()
        parallel.allreduce_gradients(model, average=False)
        parts = [torch.empty_like(mine5) for _ in range(world)]
        dist.all_gather(parts, mine5)
        ok_cells = ok_cells and torch.allclose(flat3, sum(parts), atol=1e-6)
    ok_support = ok_support and ok_cells
    # 6. evaluation: contiguous ray blocks per rank, whole image on every rank after one all-gather
    g6 = torch.Generator().manual_seed(5)
    rays6 = torch.randn(1001, 6, generator=g6)
    fake = lambda r: (r[:, :3] * 2.0 + 1.0, r[:, 3] - r[:, 4])
    rgb6, dep6 = parallel.render_sharded(fake, rays6)
    mine6 = parallel.eval_blocks(1001, world)[rank]
    part_rgb, _ = parallel.render_sharded(fake, rays6, gather=False)
    ok_eval = torch.equal(rgb6, fake(rays6)[0]) and torch.equal(dep6, fake(rays6)[1]) and \
        part_rgb.shape[0] == mine6[1] - mine6[0] and parallel.eval_blocks(1001, world) == [(0, 501), (501, 1001)]
    ok_support = ok_support and ok_eval
    s = parallel.allreduce_scalar(torch.tensor(float(rank)))
    q.put((rank, ok_shard, ok_grad, ok_flat and ok_support, float(s)))
    dist.destroy_process_group()


def test_data_parallel_gradients_over_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_shard, ok_grad, ok_flat, s in res:
        assert ok_shard and ok_grad and ok_flat, (rank, ok_shard, ok_grad, ok_flat)
        assert abs(s - 0.5) < 1e-6
