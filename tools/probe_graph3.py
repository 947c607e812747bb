import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, threading
import recon_amd
from recon_amd import field as F, autograd as A
from tests._golden import Case
from tests.helpers import build_model
c = Case("vm_cubic_train"); dev = "cuda:0"
rays = c.rays.to(dev); target = torch.from_numpy(c.expect("grad/target")).to(dev)
m = build_model(recon_amd, c, dev)
opt = torch.optim.Adam(m.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), fused=True, capturable=True)
gs = recon_amd.GraphedTrainStep(m, opt, rays.shape[0], -1, warmup=2)
orig_timed = m._timed
def spy(name, fn, *args):
    print(f"  launch {name:28s} thread={threading.current_thread().name:12s} stream={torch.cuda.current_stream().cuda_stream:#x} capturing={torch.cuda.is_current_stream_capturing()} arg_stream={args[-1]:#x}")
    return orig_timed(name, fn, *args)
m._timed = spy
for i in range(3):
    print("step", i, "side stream", hex(gs._side.cuda_stream), "default", hex(torch.cuda.default_stream().cuda_stream))
    gs._stage(rays, target)
    m.static_jitter = gs.jitter
    if i < 2:
        cur = torch.cuda.current_stream(); gs._side.wait_stream(cur)
        with torch.cuda.stream(gs._side):
            gs._body()
        cur.wait_stream(gs._side)
    else:
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=gs._side):
            gs._body()
        print("captured ok (no replay)")
torch.cuda.synchronize()
