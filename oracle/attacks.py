"""CPU restatement (numpy float64) of the four waveform attacks BASELINE.json names.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows /root/reference/uformerWM/audio_attack.py:
  low_pass_filter   :21-30   scipy.signal.butter(8, 0.5) + scipy.signal.filtfilt  — third-party arithmetic
                             (scipy, unpinned version in the reference; 1.15.3 here).  Restated below from the
                             published algorithms and PINNED against scipy itself in tests/test_attacks_cpu.py.
  amplitude_scaling :55-58
  resampling        :71-83   librosa.resample 16k->8k->16k.  librosa/resampy are not installed and the reference
                             pins neither a version nor a test vector => PARITY UNPINNED.  The build defines the
                             operation as scipy.signal.resample_poly(x,1,2) then (…,2,1) (Kaiser beta=5, 41 taps),
                             restated here and pinned against scipy.signal.resample_poly.
  awgn              :99-125  noise power from the per-waveform signal power; the random stream itself is
                             PARITY UNPINNED by design (np.random.normal), the caller supplies z ~ N(0,1).
  echo_addition :33-52, jittering_2 :176-193, requantization :85-96 — evaluation-loop attacks (row N3), see below.
All functions act on the last axis of an array [..., L].
"""
import numpy as np

PADLEN = 27


# ---- Butterworth design (analog prototype -> bilinear), equals scipy.signal.butter(N, wn) --------
def butter_lowpass(order=8, wn=0.5):
    k = np.arange(-order + 1, order, 2)
    p = -np.exp(1j * np.pi * k / (2 * order))            # unit-circle left-half-plane poles
    fs = 2.0
    warped = 2 * fs * np.tan(np.pi * wn / fs)
    p = p * warped                                          # lp2lp
    gain = warped ** order
    fs2 = 2 * fs
    pz = (fs2 + p) / (fs2 - p)                              # bilinear
    zz = -np.ones(order)
    kz = gain * np.real(1.0 / np.prod(fs2 - p))
    b = kz * np.real(np.poly(zz))
    a = np.real(np.poly(pz))
    return b, a


def lfilter_zi(b, a):
    n = len(a)
    comp = np.zeros((n - 1, n - 1))
    comp[0, :] = -a[1:] / a[0]
    comp[1:, :-1] = np.eye(n - 2)
    rhs = b[1:] - a[1:] * b[0]
    return np.linalg.solve(np.eye(n - 1) - comp.T, rhs)


def lfilter(b, a, x, zi):
    """Direct-form II transposed along the last axis; zi has shape [..., order]."""
    x = np.asarray(x, dtype=np.float64)
    z = np.array(zi, dtype=np.float64, copy=True)
    y = np.empty_like(x)
    n = len(a) - 1
    for t in range(x.shape[-1]):
        xt = x[..., t]
        yt = b[0] * xt + z[..., 0]
        for i in range(n - 1):
            z[..., i] = b[i + 1] * xt + z[..., i + 1] - a[i + 1] * yt
        z[..., n - 1] = b[n] * xt - a[n] * yt
        y[..., t] = yt
    return y


def filtfilt(b, a, x, padlen=PADLEN):
    x = np.asarray(x, dtype=np.float64)
    left = 2 * x[..., :1] - x[..., padlen:0:-1]
    right = 2 * x[..., -1:] - x[..., -2:-padlen - 2:-1]
    ext = np.concatenate([left, x, right], axis=-1)
    zi = lfilter_zi(b, a)
    y = lfilter(b, a, ext, zi * ext[..., :1])
    y = lfilter(b, a, y[..., ::-1], zi * y[..., -1:])[..., ::-1]
    return np.ascontiguousarray(y[..., padlen:-padlen])


def low_pass(x):
    b, a = butter_lowpass(8, 0.5)
    return filtfilt(b, a, x)


# ---- resampling ---------------------------------------------------------------------------------
def kaiser_lowpass_41():
    """scipy.signal.firwin(41, 0.5, window=('kaiser', 5.0)) restated."""
    n = np.arange(41) - 20
    h = 0.5 * np.sinc(0.5 * n) * np.kaiser(41, 5.0)
    return h / h.sum()


def resample_down2(x):
    h = kaiser_lowpass_41()
    x = np.asarray(x, dtype=np.float64)
    L = x.shape[-1]
    xp = np.concatenate([np.zeros(x.shape[:-1] + (20,)), x, np.zeros(x.shape[:-1] + (21,))], axis=-1)
    out = np.zeros(x.shape[:-1] + ((L + 1) // 2,))
    for d in range(-20, 21):                      # out[m] = sum_d h[20+d] x[2m - d]
        out += h[20 + d] * xp[..., 20 - d:20 - d + 2 * out.shape[-1]:2]
    return out


def resample_up2(y, L):
    h = 2.0 * kaiser_lowpass_41()
    y = np.asarray(y, dtype=np.float64)
    up = np.zeros(y.shape[:-1] + (2 * y.shape[-1],))
    up[..., ::2] = y
    up = np.concatenate([np.zeros(y.shape[:-1] + (20,)), up, np.zeros(y.shape[:-1] + (22,))], axis=-1)
    out = np.zeros(y.shape[:-1] + (L,))
    for d in range(-20, 21):                      # out[n] = sum_d h[20+d] up[n - d]
        out += h[20 + d] * up[..., 20 - d:20 - d + L]
    return out


def resampling(x):
    L = np.asarray(x).shape[-1]
    return resample_up2(resample_down2(x), L)


# ---- the rest ------------------------------------------------------------------------------------
def amplitude_scaling(x, factor=0.8):
    return np.asarray(x, dtype=np.float64) * float(factor)


def awgn_sigma(x, snr_db):
    p = np.mean(np.asarray(x, dtype=np.float64) ** 2, axis=-1, keepdims=True)
    return np.sqrt(p * 10.0 ** (-snr_db / 10.0))


def awgn(x, snr_db, z):
    return np.asarray(x, dtype=np.float64) + awgn_sigma(x, snr_db) * z


# ---- state-free attacks of the evaluation loop -----------------------------------
def echo_addition(x, fs=16000, td=0.5, aa=0.2):
    """audio_attack.py:33-52: x + aa * (x delayed by td * fs samples)."""
    x = np.asarray(x, dtype=np.float64)
    d = int(td * fs)
    echo = np.concatenate([np.zeros(d), aa * x[:len(x) - d]]) if len(x) > d else np.zeros(len(x))
    return x + echo


def jittering_2(x, indices):
    """audio_attack.py:176-193 with the drawn positions supplied by the caller (random.randint there: PARITY UNPINNED
    stream, like awgn)."""
    y = np.array(x, dtype=np.float64)
    y[np.asarray(indices, dtype=np.int64)] = 0
    return y


def requantization(x):
    """audio_attack.py:85-96: soundfile write as PCM_U8 and read back.  libsndfile (absent here, unpinned in the
    reference => PARITY UNPINNED) converts float -> u8 as lrintf(x * 127) + 128 and u8 -> float as (v - 128) / 128
    (its pcm.c f2uc_array / uc2f_array, normalised mode); inputs are clipped to [-1, 1] first (libsndfile without
    clipping wraps there).  fp32 arithmetic like the library."""
    v = np.clip(np.asarray(x, dtype=np.float32), -1.0, 1.0) * np.float32(127.0)
    return (np.rint(v) / np.float32(128.0)).astype(np.float32)


def attack_chain(x, resample=False, snr_db=None, z=None, lowpass=False, amp=None, echo=None, jitter_idx=None,
                 requantize=False):
    """Order fixed by SURVEY.md 8(d): resample -> AWGN -> low-pass -> amplitude scale; then the row-N3 stages
    echo (delay, gain) -> jittering_2 (per-row index lists) -> 8-bit requantisation.  x: [rows, L]."""
    x = np.asarray(x, dtype=np.float64)
    if resample:
        x = resampling(x)
    if snr_db is not None:
        x = awgn(x, snr_db, z)
    if lowpass:
        x = low_pass(x)
    if amp is not None:
        x = amplitude_scaling(x, amp)
    if echo is not None:
        d, g = echo
        x = np.stack([echo_addition(r, fs=1, td=d, aa=g) for r in np.atleast_2d(x)]).reshape(x.shape)
    if jitter_idx is not None:
        x = np.stack([jittering_2(r, i) for r, i in zip(np.atleast_2d(x), jitter_idx)]).reshape(x.shape)
    if requantize:
        x = requantization(x).astype(np.float64)
    return x
