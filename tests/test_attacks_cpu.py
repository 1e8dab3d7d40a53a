"""oracle/attacks.py pinned against the third-party routines the reference calls (scipy)."""
import numpy as np
import pytest
from scipy import signal

from oracle import attacks as A


def test_butter_matches_scipy():
    b, a = A.butter_lowpass(8, 0.5)
    bs, as_ = signal.butter(8, 0.5, "lowpass")          # audio_attack.py:27-28 with Fs=16000, 8000 -> wn = 0.5
    np.testing.assert_allclose(b, bs, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(a, as_, rtol=1e-10, atol=1e-13)


def test_filtfilt_matches_scipy():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3, 1500))
    bs, as_ = signal.butter(8, 0.5, "lowpass")
    np.testing.assert_allclose(A.low_pass(x), signal.filtfilt(bs, as_, x), rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("L", [8002, 801, 64])
def test_resampling_matches_resample_poly(L):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, L))
    np.testing.assert_allclose(A.kaiser_lowpass_41(), signal.firwin(41, 0.5, window=("kaiser", 5.0)), atol=1e-15)
    d = signal.resample_poly(x, 1, 2, axis=-1)
    np.testing.assert_allclose(A.resample_down2(x), d, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(A.resampling(x), signal.resample_poly(d, 2, 1, axis=-1)[..., :L], rtol=1e-10, atol=1e-12)


def test_awgn_snr_and_amp():
    rng = np.random.default_rng(2)
    x = rng.standard_normal((4, 8002)) * 0.1
    z = rng.standard_normal(x.shape)
    y = A.awgn(x, 20.0, z)
    snr = 10 * np.log10((x ** 2).mean(-1) / ((y - x) ** 2).mean(-1))
    assert np.all(np.abs(snr - 20.0) < 0.3)
    np.testing.assert_allclose(A.amplitude_scaling(x, 0.8), 0.8 * x)
