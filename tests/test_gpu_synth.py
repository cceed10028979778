"""-m gpu: the on-device synthetic generator.  Exact against its NumPy restatement
(oracle/synth_ref.py: same Philox stream, same event table) and distribution-level against waterfalls
captured from the reference generator (tests/golden/preprocess.npz)."""
import os

import numpy as np
import pytest

from oracle import synth_ref
from rfi_toolbox_amd.data_generation import SyntheticWaterfalls, make_training_patches_device
from rfi_toolbox_amd.data_generation.synthetic import YAML_4K_COUNTS

pytestmark = pytest.mark.gpu


def test_philox_known_answers():
    """Random123 known-answer vectors for philox4x32_10 pin the oracle's (and hence the kernel's) stream."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = synth_ref.philox4x32_10(*[[c] for c in ctr], *key)
        assert tuple(int(g[0]) for g in got) == want


@pytest.mark.parametrize("npol,shape,dtype", [(1, (128, 128), np.complex128), (4, (96, 160), np.complex128),
                                              (2, (128, 128), np.complex64)])
def test_device_generator_matches_oracle(npol, shape, dtype):
    counts = {"narrowband_persistent": 2, "broadband_persistent": 1, "frequency_sweep": 2,
              "narrowband_bursty": 2, "broadband_bursty": 1, "narrowband_intermittent": 1}
    gen = SyntheticWaterfalls(shape[0], shape[1], npol, counts=counts, seed=5)
    planes, flags, events = gen.sample_device(3, dtype=dtype, seed=0x1234567890abcdef)
    want_p, want_f = synth_ref.generate(0x1234567890abcdef, events, npol, shape[0], shape[1])
    got_p, got_f = planes.numpy(), flags.numpy()
    np.testing.assert_array_equal(got_f, want_f)                       # exact mask
    assert got_f[:, :2].any() and (npol < 3 or not got_f[:, 2:].any())
    scale = np.abs(want_p).max()
    tol = 1e-12 if dtype == np.complex128 else 2e-7                    # libm differences only / fp32 rounding
    assert np.abs(got_p - want_p).max() <= tol * scale
    # the flagged pixels carry the summed event amplitudes (>= 1e6 mJy), the rest stays at the noise level
    amp = np.abs(got_p[:, 0])
    assert amp[got_f[:, 0] == 1].min() > 9e5 and amp[got_f[:, 0] == 0].max() < 2.0


def test_distribution_against_reference_fixture(golden_dir):
    """Moments of the unflagged pixels and of the phase, next to a waterfall captured from the
    reference's _generate_single_sample (same 64 x 64 geometry, bandpass on)."""
    g = np.load(os.path.join(golden_dir, "preprocess.npz"))
    ref_w, ref_m = g["a_w"][0, 0], g["a_m"][0, 0].astype(bool)
    gen = SyntheticWaterfalls(64, 64, 1, counts={"narrowband_persistent": 1, "broadband_persistent": 1}, seed=8)
    planes, flags, _ = gen.sample_device(16, seed=77)
    w, m = planes.numpy()[:, 0], flags.numpy()[:, 0].astype(bool)
    core = slice(8, 56)                                                # inside the bandpass plateau
    a_ref, a_dev = np.abs(ref_w[core][~ref_m[core]]), np.abs(w[:, core][~m[:, core]])
    assert abs(a_dev.mean() - a_ref.mean()) < 0.02 and abs(a_dev.std() - a_ref.std()) < 0.02    # N(1, 0.1)
    ph_ref, ph_dev = np.angle(ref_w).ravel(), np.angle(w).ravel()
    for ph in (ph_ref, ph_dev):                                        # uniform on (-pi, pi]
        assert abs(ph.mean()) < 0.12 and abs(ph.var() - np.pi ** 2 / 3) < 0.25
    edge_ref, edge_dev = np.abs(ref_w[2][~ref_m[2]]).mean(), np.abs(w[:, 2][~m[:, 2]]).mean()
    assert edge_dev == pytest.approx((2 / 6) ** 8, rel=0.2) and edge_ref == pytest.approx((2 / 6) ** 8, rel=0.5)


def test_pipeline_stays_in_hbm():
    imgs, labs = make_training_patches_device(8, 128, seed=3)
    x, y = imgs.numpy(), labs.numpy()
    assert x.shape == (8, 128, 128, 3) and y.shape == (8, 128, 128)
    assert np.isfinite(x).all() and set(np.unique(y)) <= {0, 1} and y.any(axis=(1, 2)).all()
    # channel 0 is min-max scaled per patch before the ImageNet normalisation
    c0 = x[..., 0] * 0.229 + 0.485
    assert np.allclose(c0.min(axis=(1, 2)), 0, atol=1e-6) and np.allclose(c0.max(axis=(1, 2)), 1, atol=1e-6)
