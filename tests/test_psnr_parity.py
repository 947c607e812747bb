"""BASELINE metric, quality half: PSNR of the HIP path within 0.1 dB of the reference PyTorch path after equal
iterations (north_star).  The eager side is the oracle on the same GPU (the reference's arithmetic, pinned by the golden
fixtures, its schedule steps by tests/test_oracle_schedule.py); both students start from the same parameters and see
the same batches and jitter.

Round 2 missed this bar at 400 iterations with the schedule events in the run (+0.116 dB, reproducible) and moved the
comparison to 600.  The cause (tests/psnr_event_diag.py): not the events — the gap was already there at
iteration 25 of any run that starts from a 48^3 grid.  While a fresh field has no shaded sample the reference's
appearance tensors and MLP have no gradient and torch.optim.Adam does not count those steps for them
(tensorBase.py:370, train.py:374-376); the HIP path handed Adam zero gradients instead, so its bias corrections ran a
few steps ahead and the first appearance / MLP updates were up to 26 % smaller.  Fixed in autograd.py (None gradients)
and FusedAdam (per-parameter step counts behind device-side gates; tests/test_adam_trajectory.py).  The bar is back at
400 iterations, checked at several cut points and for a second seed."""
import numpy as np
import pytest

from oracle import psnr_parity

pytestmark = pytest.mark.gpu


def _check(r):
    print(r)
    assert r["psnr_hip_db"] > 25.0 and r["psnr_eager_db"] > 25.0, r          # both actually learned the scene
    assert abs(r["delta_db"]) <= 0.1, r
    for c, v in r["cuts"].items():
        assert abs(v["delta_db"]) <= 0.1, (c, v)


def test_psnr_within_a_tenth_of_a_db_after_equal_iterations(recon):
    _check(psnr_parity.run(recon, grid=64, iters=400, cuts=(100, 200, 300)))


@pytest.mark.parametrize("iters,seed", [(400, 5), (400, 6), (600, 5)])
def test_psnr_parity_through_mask_update_and_upsampling(recon, iters, seed):
    """The same with the schedule's two kinds of events inside the run (train.py:450-481): an alpha-mask rebuild at 3/8
    and a 48^3 -> 64^3 up-sampling (optimizer rebuilt) at 5/8 of the iterations, each student doing its own; compared at
    the end and at cut points before, between and after the events."""
    cuts = tuple(iters * k // 8 for k in (2, 4, 6, 7))
    _check(psnr_parity.run(recon, grid=64, iters=iters, schedule=True, init_grid=48, seed=seed, cuts=cuts))


def test_psnr_parity_through_the_whole_schedule(recon):
    """the reference's INTENDED schedule end to end — bbox ray filtering, MSE + ortho / L1 / TV terms with
    decaying TV weights, alpha-mask update + SHRINK (L1 weight switched, optimizer rebuilt at the decayed rates), two
    up-samplings (N rule of train.py:472, learning rates reset): the product's `harness.train`, eager AND captured,
    against the same loop restated on the oracle, whose schedule steps and regulariser terms are pinned to the reference's
    own outputs (tests/test_oracle_schedule.py).  Same events, same grids / boxes / sample counts, PSNR within 0.1 dB
    (measured: HIP eager 34.53-34.57, captured 34.54-34.57, oracle 34.54-34.55 dB over repeated runs)."""
    r = psnr_parity.run_full(recon)
    print(r)
    assert len(r["hip"]) == 2
    for h in r["hip"]:
        assert [e[:2] for e in h["events"]] == [e[:2] for e in r["events_eager"]], (h, r["events_eager"])
        assert h["grid"] == r["grid_eager"] and h["n_samples"] == r["n_samples_eager"], (h, r)
        assert np.allclose(np.array(h["aabb"]), np.array(r["aabb_eager"]), rtol=0, atol=1e-6), (h["aabb"], r["aabb_eager"])
        assert h["psnr_hip_db"] > 30.0 and r["psnr_eager_db"] > 30.0, (h, r)
        assert abs(h["delta_db"]) <= 0.1, (h, r["psnr_eager_db"])
