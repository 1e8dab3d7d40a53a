"""CPU restatement (numpy) of the phase-vocoder attacks of the evaluation loop.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Reference call sites: /root/reference/uformerWM/audio_attack.py:195-231
``time_scaling`` = ``librosa.effects.time_stretch(sw, scaling_factor)`` and :233-239 ``pitch_scaling`` =
``librosa.effects.pitch_shift(S, sr=16000, n_steps=-6)``.  librosa (and resampy behind ``librosa.resample``) is a
third-party dependency that is NOT in the reference tree, pinned nowhere (no requirements file; the positional call at :199
only works before librosa 0.10) and not installed here, and the reference holds no fixture for these attacks:
**PARITY UNPINNED**.  What follows restates the published algorithms of librosa 0.9 (core.stft / phase_vocoder / istft,
effects.time_stretch / pitch_shift) and of resampy's ``kaiser_best`` interpolator with numpy; the HIP kernels
(csrc/vocoder.hip) are held to this restatement.
"""
import numpy as np

N_FFT, HOP = 2048, 512


def hann():
    return (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N_FFT) / N_FFT)).astype(np.float32)      # get_window('hann', fftbins=True)


def stft(y):
    yp = np.pad(np.asarray(y, dtype=np.float32), N_FFT // 2, mode="reflect")
    T = 1 + (len(yp) - N_FFT) // HOP
    frames = np.stack([yp[HOP * t:HOP * t + N_FFT] for t in range(T)], axis=1) * hann()[:, None]
    return np.fft.rfft(frames, axis=0).astype(np.complex64)                                      # [1025, T]


def phase_vocoder(D, rate):
    nb = D.shape[0]
    steps = np.arange(0, D.shape[1], rate, dtype=np.float64)
    out = np.zeros((nb, len(steps)), dtype=np.complex64)
    phi = np.linspace(0, np.pi * HOP, nb)
    acc = np.angle(D[:, 0]).astype(np.float32)
    Dp = np.pad(D, [(0, 0), (0, 2)], mode="constant")
    for t, step in enumerate(steps):
        cols = Dp[:, int(step):int(step + 2)]
        alpha = np.mod(step, 1.0)
        mag = (1.0 - alpha) * np.abs(cols[:, 0]) + alpha * np.abs(cols[:, 1])
        out[:, t] = mag * np.exp(1j * acc)
        dphase = np.angle(cols[:, 1]) - np.angle(cols[:, 0]) - phi
        dphase = dphase - 2.0 * np.pi * np.round(dphase / (2.0 * np.pi))
        acc = (acc + phi + dphase).astype(np.float32)
    return out


def istft(S, length):
    T = S.shape[1]
    w = hann()
    expected = N_FFT + HOP * (T - 1)
    y = np.zeros(expected, dtype=np.float32)
    wss = np.zeros(expected, dtype=np.float32)
    fr = np.fft.irfft(S, n=N_FFT, axis=0).astype(np.float32) * w[:, None]
    for t in range(T):
        y[HOP * t:HOP * t + N_FFT] += fr[:, t]
        wss[HOP * t:HOP * t + N_FFT] += w * w
    nz = wss > np.finfo(np.float32).tiny
    y[nz] /= wss[nz]
    y = y[N_FFT // 2:]
    return np.pad(y, (0, max(0, length - len(y))))[:length]


def time_stretch(y, rate):
    return istft(phase_vocoder(stft(y), rate), int(round(len(y) / rate)))


def kaiser_best_table(num_zeros=64, precision=9, rolloff=0.9475937167399596, beta=14.769656459379492):
    """resampy.filters.sinc_window: the right half of rolloff * sinc(rolloff * t) * kaiser, 2**precision samples per crossing."""
    num_bits = 2 ** precision
    n = num_bits * num_zeros
    t = np.linspace(0, num_zeros, num=n + 1, endpoint=True)
    taper = np.kaiser(2 * n + 1, beta)[n:]
    win = (rolloff * np.sinc(rolloff * t) * taper).astype(np.float32)
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    return win, delta, num_bits


def sinc_resample(x, ratio, n_out):
    """resampy.interpn.resample_f for one channel, written with numpy per output sample."""
    win, delta, precision = kaiser_best_table()
    x = np.asarray(x, dtype=np.float32)
    scale = min(1.0, ratio)
    gain = np.float32(ratio) if ratio < 1 else np.float32(1.0)
    index_step = int(scale * precision)
    inc = 1.0 / ratio
    nwin = len(win)
    out = np.zeros(n_out, dtype=np.float32)
    for t in range(n_out):
        treg = t * inc
        n = int(treg)
        if n >= len(x):
            continue
        frac = scale * (treg - n)
        idx = frac * precision
        off = int(idx)
        eta = np.float32(idx - off)
        imax = min(n + 1, (nwin - off) // index_step)
        i = np.arange(imax)
        wts = (win[off + i * index_step] + eta * delta[off + i * index_step]) * gain
        acc = np.sum(wts * x[n - i], dtype=np.float64)
        frac = scale - frac
        idx = frac * precision
        off = int(idx)
        eta = np.float32(idx - off)
        kmax = min(len(x) - n - 1, (nwin - off) // index_step)
        k = np.arange(kmax)
        wts = (win[off + k * index_step] + eta * delta[off + k * index_step]) * gain
        acc += np.sum(wts * x[n + k + 1], dtype=np.float64)
        out[t] = acc
    return out


def pitch_shift(y, n_steps=-6, bins_per_octave=12):
    """librosa.effects.pitch_shift: stretch by rate = 2^(-n_steps / 12), resample by the same ratio, fix the length."""
    rate = 2.0 ** (-float(n_steps) / bins_per_octave)
    ys = time_stretch(y, rate)
    n = int(np.ceil(len(ys) * rate))
    out = sinc_resample(ys, rate, n)
    return np.pad(out, (0, max(0, len(y) - n)))[:len(y)]
