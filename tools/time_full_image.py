"""Throughput of rendering one whole 800 x 800 view (640 000 rays) through OctreeRender_trilinear_fast, chunk = 4096 as in
the reference's evaluation loop (renderer.py:67), on the bench scene (config 2)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import recon_amd
sys.argv = [sys.argv[0]]
import bench
model, rays, targets, n_samples, reso = bench.build_scene(recon_amd, torch.device("cuda", 0), 300, 1)
model.lazy_sample_count = True
img = rays[:640000]
import gc; gc.collect(); gc.freeze()
def run():
    with torch.no_grad():
        return recon_amd.OctreeRender_trilinear_fast(img, model, chunk=4096, N_samples=n_samples, white_bg=True, device="cuda:0")
for _ in range(2): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): out = run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"{img.shape[0]} rays in {dt*1e3:.2f} ms = {img.shape[0]/dt/1e6:.2f} M rays/s (super-chunks of {getattr(model, 'super_chunk', 32768)} rays per launch)")
for sc in (4096, 65536, 131072):
    model.super_chunk = sc
    for _ in range(2): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"  super_chunk {sc}: {dt*1e3:.2f} ms = {img.shape[0]/dt/1e6:.2f} M rays/s")
