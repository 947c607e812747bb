"""BASELINE metric, quality half: PSNR of the HIP path within 0.1 dB of the reference PyTorch path after equal
iterations (north_star).  The eager side is the oracle on the same GPU (the reference's arithmetic, pinned by the golden
fixtures); both students start from the same parameters and see the same batches and jitter."""
import pytest

from tests import psnr_parity

pytestmark = pytest.mark.gpu


def test_psnr_within_a_tenth_of_a_db_after_equal_iterations(recon):
    r = psnr_parity.run(recon, grid=64, iters=400)
    print(r)
    assert r["psnr_hip_db"] > 25.0 and r["psnr_eager_db"] > 25.0, r          # both actually learned the scene
    assert abs(r["delta_db"]) <= 0.1, r


def test_psnr_parity_through_mask_update_and_upsampling(recon):
    """The same with the schedule's two kinds of events inside the run (train.py:450-481): an alpha-mask rebuild at
    iteration 225 and a 48^3 -> 64^3 up-sampling (optimizer rebuilt) at 375 of 600, each student doing its own.
    (Measured: 33.718 vs 33.717 dB.  Right after an event the two trajectories are transiently further apart — the
    same run cut at 400 iterations, 150 after its up-sampling, reads 32.97 vs 32.86 dB — and meet again as the learning
    rate decays: "equal iterations" is compared where both have settled.)"""
    r = psnr_parity.run(recon, grid=64, iters=600, schedule=True, init_grid=48)
    print(r)
    assert r["psnr_hip_db"] > 25.0 and r["psnr_eager_db"] > 25.0, r
    assert abs(r["delta_db"]) <= 0.1, r
