"""Host side of the fused waveform attacks (kernels: csrc/attack.hip).

Mirrors the callables of the reference's uformerWM/audio_attack.py for the four distortions
BASELINE.json names and the "name-param" string dispatch of uformerWM/audio_test.py:631-660, but as
differentiable device ops usable inside the train step (hook point model.py:2461-2462).
Filter coefficients are derived here in float64 and handed to the kernel as small device tables.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import ops
from ._lib import WmAttack, ptr

LP_TAPS = 128          # impulse response of butter(8, 0.5) is < 5e-12 beyond 128 taps


def _butter8_half():
    """Butterworth(8) low-pass at wn = 0.5 via bilinear transform (what audio_attack.py:27-28 asks scipy for)."""
    order, wn, fs = 8, 0.5, 2.0
    k = np.arange(-order + 1, order, 2)
    warped = 2 * fs * np.tan(np.pi * wn / fs)
    p = -np.exp(1j * np.pi * k / (2 * order)) * warped
    fs2 = 2 * fs
    a = np.real(np.poly((fs2 + p) / (fs2 - p)))
    b = (warped ** order) * np.real(1.0 / np.prod(fs2 - p)) * np.real(np.poly(-np.ones(order)))
    return b, a


def lowpass_fir_tables(taps=LP_TAPS):
    """h[j] = impulse response of b/a, tail[j] = 1 - sum_{i<=j} h[i] (steady-state term of filtfilt's zi)."""
    b, a = _butter8_half()
    h = np.zeros(taps)
    for n in range(taps):
        acc = b[n] if n < len(b) else 0.0
        for i in range(1, min(n, len(a) - 1) + 1):
            acc -= a[i] * h[n - i]
        h[n] = acc
    return np.concatenate([h, 1.0 - np.cumsum(h)]).astype(np.float32)


def resample_fir_table():
    n = np.arange(41) - 20
    h = 0.5 * np.sinc(0.5 * n) * np.kaiser(41, 5.0)
    return (h / h.sum()).astype(np.float32)


@dataclass
class AttackConfig:
    resample: bool = False
    snr_db: Optional[float] = None
    lowpass: bool = False
    amp: Optional[float] = None
    seed: int = 7
    echo: Optional[tuple] = None          # (delay in samples, gain): audio_attack.py:33-52 uses (8000, 0.2)
    jitter: Optional[int] = None          # number of zeroed samples per row (jittering_2, :176-193)
    requantize: bool = False              # 8-bit PCM round trip (:85-96), straight-through gradient
    mfma_fir: bool = True                 # low-pass passes on the matrix cores (False: the scalar fp32 loops)

    @property
    def flags(self):
        return (1 if self.resample else 0) | (2 if self.snr_db is not None else 0) | (4 if self.lowpass else 0) | \
               (8 if self.amp is not None else 0) | (16 if self.echo is not None else 0) | \
               (32 if self.jitter is not None else 0) | (64 if self.requantize else 0)

    @staticmethod
    def identity():
        return AttackConfig()

    @staticmethod
    def full_stack():
        """BASELINE config 3 / SURVEY 8(d): resample 16k->8k->16k, AWGN 20 dB, Butterworth-8 low-pass, x0.8."""
        return AttackConfig(True, 20.0, True, 0.8, 7)

    @staticmethod
    def from_string(spec):
        """'name-param' dispatch of audio_test.py:631-660 for the in-scope attacks (several joined by '+')."""
        cfg = AttackConfig()
        for part in spec.split("+"):
            name, _, param = part.partition("-")
            if name in ("", "no", "closed_loop", "identity"):
                continue
            elif name == "resampling":
                cfg.resample = True
            elif name == "awgn":
                cfg.snr_db = float(param) if param else 15.0
            elif name in ("low_pass", "lowpass", "low_pass_filter"):   # 'low_pass' is the reference's dispatch string (:637)
                cfg.lowpass = True
            elif name in ("amplitude", "amplitude_scaling"):
                cfg.amp = float(param) if param else 0.8
            elif name == "echo_addition":
                # the reference ignores any parameter (audio_test.py:632-633 calls echo_addition(audio) with the defaults
                # td = 0.5 s, AA = 0.2 of audio_attack.py:33); a delay in SAMPLES is accepted as 'echo_addition_samples-<n>'
                cfg.echo = (8000, 0.2)
            elif name == "echo_addition_samples":
                cfg.echo = (int(param), 0.2)
            elif name == "jittering_2":
                cfg.jitter = int(param) if param else 1000
            elif name == "requantization":
                cfg.requantize = True
            else:
                raise ValueError(f"attack '{name}' is outside the MI355X hot path (see DESIGN.md, out of scope)")
        return cfg


_tables = {}


def rank_seed(seed):
    """Per-rank Philox key: every data-parallel rank draws its own AWGN / jitter / DropPath streams."""
    import torch.distributed as dist
    r = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    return (int(seed) + 0x9E3779B1 * r) & 0xFFFFFFFFFFFFFFFF


def _device_tables(device):
    """(low-pass taps | tails, resampling taps, MFMA operand image of the low-pass taps) on `device`, built once (outside
    any graph capture: the first, eager, call makes them)."""
    key = str(device)
    if key not in _tables:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("attack tables: first use inside a graph capture; run one eager step first")
        lp, rs = torch.from_numpy(lowpass_fir_tables()).to(device), torch.from_numpy(resample_fir_table()).to(device)
        fir = torch.empty(int(ops.lib.wm_attack_fir_table_bytes()) // 2, device=device, dtype=torch.bfloat16)
        ops.check(ops.lib.wm_attack_fir_tables(ptr(lp), LP_TAPS, ptr(fir), torch.cuda.current_stream().cuda_stream),
                  "wm_attack_fir_tables")
        torch.cuda.current_stream().synchronize()
        _tables[key] = (lp, rs, fir)
    return _tables[key]


def _desc(cfg, device, step, precise=False):
    """precise: the two low-pass passes as scalar fp32 loops instead of split-precision MFMA products (1e-7 instead of 4e-6 .. 2e-5
    of full scale each; the fp32 / split compute modes ask for it)."""
    lp, rs, fir = _device_tables(device)
    delay, gain = cfg.echo if cfg.echo is not None else (0, 0.0)
    return WmAttack(cfg.flags, float(cfg.snr_db or 0.0), float(cfg.amp if cfg.amp is not None else 1.0), ptr(lp),
                    LP_TAPS, ptr(rs), 41, rank_seed(cfg.seed), ptr(step), int(delay), float(gain), int(cfg.jitter or 0), 0,
                    ptr(fir) if (cfg.mfma_fir and not precise) else None)


def attack_forward(cfg, wave, step, out=None, sigma=None, precise=False):
    B, L = wave.shape
    out = torch.empty_like(wave) if out is None else out
    if sigma is None:
        sigma = torch.empty(B, device=wave.device, dtype=torch.float32)
    ops.attack_fwd(_desc(cfg, wave.device, step, precise), wave, out, sigma, B, L)
    return out, sigma


def attack_backward(cfg, wave, dout, sigma, step, dwave=None, precise=False):
    B, L = wave.shape
    dwave = torch.empty_like(wave) if dwave is None else dwave
    ops.attack_bwd(_desc(cfg, wave.device, step, precise), wave, dout, sigma, dwave, B, L)
    return dwave
