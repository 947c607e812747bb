"""Cost of the data-parallel exchange's local part on one GPU (config 2): gather of the exchanged cells, the write-back,
against the row-block variant (cat / foreach_copy) and a whole-buffer pass; the collective itself needs N > 1 GPUs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import recon_amd as recon
from recon_amd import parallel
from tests.test_full_size import _scene

for name in sys.argv[1:] or ["C2_vm300"]:
    model, rays, N, ndc, white = _scene(recon, name)
    rgb, _, _ = model(rays, None, white_bg=True, is_train=True, ndc_ray=ndc, N_samples=N)
    (rgb ** 2).mean().backward()
    flat = model.grad_flat
    segs = parallel.gradient_support(model)
    w, idx = parallel.gradient_support_rows(model)
    table = flat.view(-1, w)
    def t(fn, n=50):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    buf = table.index_select(0, idx)
    pieces = [flat[a:b] for a, b in segs] if segs else [flat]
    cat = torch.cat(pieces)
    sizes = [p.numel() for p in pieces]
    print(f"{name}: buffer {flat.numel()*4/1e6:.1f} MB; row blocks {cat.numel()*4/1e6:.1f} MB; cells {buf.numel()*4/1e6:.1f} MB (rows of {w} floats)")
    from recon_amd import _hip as H
    from recon_amd.field import _stream
    idx32 = idx.to(torch.int32)
    g = lambda: H.lib().tf_gather_rows(flat.data_ptr(), idx32.data_ptr(), idx.numel(), w, buf.data_ptr(), _stream())
    sc = lambda: H.lib().tf_scatter_rows(flat.data_ptr(), idx32.data_ptr(), idx.numel(), w, buf.data_ptr(), _stream())
    ref = table.index_select(0, idx); g(); torch.cuda.synchronize()
    assert torch.equal(ref, buf)
    print(f"  cells (tf_gather_rows / tf_scatter_rows): gather {t(g):.1f} us, write-back {t(sc):.1f} us")
    print(f"  cells (torch index_select / index_copy_): gather {t(lambda: table.index_select(0, idx)):.1f} us, write-back {t(lambda: table.index_copy_(0, idx, buf)):.1f} us")
    print(f"  blocks: cat {t(lambda: torch.cat(pieces)):.1f} us, write-back {t(lambda: torch._foreach_copy_(pieces, list(cat.split(sizes)))):.1f} us, "
          f"scale pass {t(lambda: cat.mul_(0.5)):.1f} us")
