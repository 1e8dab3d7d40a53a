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


def test_torch_restatement_matches_numpy_oracle_and_is_differentiable():
    """oracle/attacks_torch.py (autograd-capable, used by the config-3 CPU baseline and gradient checks) against the numpy
    float64 chain that the scipy tests above pin."""
    import torch
    from oracle import attacks_torch as AT
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 8002)) * 0.2
    z = rng.standard_normal(x.shape)
    xt = torch.from_numpy(x).requires_grad_()
    np.testing.assert_allclose(AT.low_pass(xt).detach().numpy(), A.low_pass(x), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(AT.resampling(xt).detach().numpy(), A.resampling(x), rtol=1e-10, atol=1e-12)
    full = AT.attack_chain(xt, resample=True, snr_db=20.0, z=torch.from_numpy(z), lowpass=True, amp=0.8)
    ref = A.attack_chain(x, resample=True, snr_db=20.0, z=z, lowpass=True, amp=0.8)
    np.testing.assert_allclose(full.detach().numpy(), ref, rtol=1e-9, atol=1e-11)
    g = torch.from_numpy(rng.standard_normal(x.shape))
    (full * g).sum().backward()
    v = rng.standard_normal(x.shape)
    eps = 1e-6
    f = lambda xx: (A.attack_chain(xx, resample=True, snr_db=20.0, z=z, lowpass=True, amp=0.8) * g.numpy()).sum()
    fd = (f(x + eps * v) - f(x - eps * v)) / (2 * eps)
    assert abs(fd - (xt.grad.numpy() * v).sum()) <= 1e-6 * abs(fd) + 1e-9
    # odd length / short rows
    xs = rng.standard_normal((1, 801))
    np.testing.assert_allclose(AT.attack_chain(torch.from_numpy(xs), resample=True, lowpass=True).numpy(),
                               A.attack_chain(xs, resample=True, lowpass=True), rtol=1e-9, atol=1e-11)
