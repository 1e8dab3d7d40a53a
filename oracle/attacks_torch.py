"""Differentiable torch (CPU) restatement of the in-loop attack chain: the same operators as oracle/attacks.py, written
with torch ops so that autograd supplies the gradients the config-3 train step needs.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Reference: uformerWM/audio_attack.py:21-30 (low_pass_filter),
:55-58 (amplitude_scaling), :71-83 (resampling), :99-125 (awgn), applied at the hook point model.py:2461-2462 in the
order SURVEY.md 8(d) fixes: resample -> AWGN -> low-pass -> amplitude scale.  Pinned in tests/test_attacks_cpu.py
against oracle/attacks.py (numpy float64), which is itself pinned against scipy; `resampling` (librosa in the
reference) and the AWGN random stream stay PARITY UNPINNED exactly as described there.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import attacks as A

LP_TAPS = 160          # impulse response of butter(8, 0.5): below 1e-14 beyond 160 taps (float64 here)


def _impulse_response(taps=LP_TAPS):
    b, a = A.butter_lowpass(8, 0.5)
    h = np.zeros(taps)
    for n in range(taps):
        acc = b[n] if n < len(b) else 0.0
        for i in range(1, min(n, len(a) - 1) + 1):
            acc -= a[i] * h[n - i]
        h[n] = acc
    return h


def _causal_fir(e, h, tail):
    """y[n] = sum_{j <= n} h[j] e[n - j] + e[0] * tail[min(n, J - 1)]: lfilter(b, a, e, zi = zi0 * e[0]) with the IIR
    response expanded to J taps (the steady-state initial condition makes the filter see e[0] to the left for ever)."""
    J, N = h.numel(), e.shape[-1]
    y = F.conv1d(F.pad(e[:, None], (J - 1, 0)), h.flip(0)[None, None])[:, 0]
    t = torch.cat([tail, tail[-1:].expand(max(N - J, 0))])[:N]
    return y + e[:, :1] * t


def low_pass(x):
    """scipy.signal.filtfilt(butter(8, 0.5), x): odd extension by padlen 27, forward pass, backward pass, trim."""
    dt = x.dtype
    h = torch.from_numpy(_impulse_response()).to(dt)
    tail = 1.0 - torch.cumsum(h, 0)
    p = A.PADLEN
    left = 2 * x[:, :1] - x[:, 1:p + 1].flip(-1)
    right = 2 * x[:, -1:] - x[:, -p - 1:-1].flip(-1)
    e = torch.cat([left, x, right], -1)
    y = _causal_fir(e, h, tail)
    y = _causal_fir(y.flip(-1), h, tail).flip(-1)
    return y[:, p:-p]


def resampling(x):
    """resample_poly(x, 1, 2) then (…, 2, 1) with the 41-tap Kaiser(5) half-band filter (oracle/attacks.py)."""
    L = x.shape[-1]
    h = torch.from_numpy(A.kaiser_lowpass_41()).to(x.dtype)
    w = h.flip(0)[None, None]
    Ld = (L + 1) // 2
    xp = F.pad(x[:, None], (20, 21))
    down = F.conv1d(xp, w, stride=2)[:, 0, :Ld]                         # out[m] = sum_d h[20 + d] x[2 m - d]
    up = torch.zeros(x.shape[0], 2 * Ld, dtype=x.dtype)
    up[:, ::2] = down
    upp = F.pad(up[:, None], (20, 22))
    return 2.0 * F.conv1d(upp, w)[:, 0, :L]


def awgn(x, snr_db, z):
    p = (x * x).mean(-1, keepdim=True)
    return x + torch.sqrt(p * 10.0 ** (-snr_db / 10.0)) * z


def attack_chain(x, resample=False, snr_db=None, z=None, lowpass=False, amp=None):
    if resample:
        x = resampling(x)
    if snr_db is not None:
        x = awgn(x, snr_db, z)
    if lowpass:
        x = low_pass(x)
    if amp is not None:
        x = x * float(amp)
    return x
