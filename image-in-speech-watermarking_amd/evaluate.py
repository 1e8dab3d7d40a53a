"""Whole-utterance evaluation path on the device: the ``data_mode='stft'`` and ``'audio'`` branches
of the reference's ``SpeechDataTest`` / ``SpeechDataAudio.prepare_data`` (audio_test.py:299-343, :135-153) and
``reconstruct_audio`` (audio_test.py:528-785), every ``audio_scale`` form included ('0', a factor, the 'lo-hi' range).

Same call surface and return tuple as the reference; what changes is where the work runs:

* the utterance is analysed (``wm_stft255_tiles``), stitched back (``wm_istft255_stitch``) and re-analysed after the
  attack by HIP kernels, the waveform never leaves HBM (the reference round-trips every tile through numpy);
* the tiles of an utterance go through the model as one batch (the reference calls it tile by tile);
* the attacks that run here are the state-free ones (``closed_loop``, ``amplitude_scaling-f``, ``awgn-snr``,
  ``echo_addition``, ``jittering_2-n``, ``requantization``); the filter / codec / stretch attacks of audio_attack.py
  (scipy, librosa, ffmpeg) are CPU tools outside the hot path and raise.
"""
import math

import torch

from . import ops


def prepare_utterance(wave, audio_scale="0", data_min=None, data_max=None, data_mode="stft"):
    """One entry of SpeechDataTest.prepare_data: ``[wave, clips, len_last_clip]`` with clips ``[n, 2, 128, 128]``
    (``data_mode='audio'``: SpeechDataAudio.prepare_data, audio_test.py:135-153 — the waveform zero-padded to a multiple of
    128 * 128 samples and cut into ``[n, 1, 128, 128]`` clips; the pad is a FULL clip when the length already divides).

    ``audio_scale``: the reference's string option; a plain number of more than one character scales the tiles
    (audio_test.py:327-330), '0' leaves them alone, 'lo-hi' maps them from [data_min, data_max] — the training set's
    statistics (dataprep.normalize_tiles returns them; audio_test.py:331-338) — into [lo, hi]."""
    from .dataprep import parse_audio_scale, _stats
    if wave.dim() == 1:
        wave = wave[None]
    if wave.dim() != 2 or wave.shape[0] != 1 or not wave.is_cuda:
        raise ValueError("prepare_utterance takes one CUDA waveform [1, L]")
    if data_mode == "audio":
        L = wave.shape[-1]
        len_pad = 128 * 128 - L % (128 * 128)
        clips = torch.nn.functional.pad(wave.float(), (0, len_pad)).view(-1, 1, 128, 128).contiguous()
        return [wave, clips, 128 * 128 - len_pad]
    if data_mode != "stft":
        raise ValueError("data_mode 'dwt' needs pywt (a CPU library outside the hot path); 'stft' and 'audio' run here")
    mode = parse_audio_scale(audio_scale)
    if mode is not None and mode[0] == "range":
        if data_min is None or data_max is None:
            raise ValueError("range-style audio_scale maps from the training set's [data_min, data_max] (audio_test.py:338): pass them")
        clips, frames = ops.stft255_tiles(wave.contiguous().float(), 1.0)
        clips = ops.range_map(clips, _stats(data_min, data_max, wave.device), mode[1], mode[2])
        return [wave, clips, frames % 128]
    scale = mode[1] if mode is not None else 1.0
    clips, frames = ops.stft255_tiles(wave.contiguous().float(), scale)
    return [wave, clips, frames % 128]


def signaltonoise(a):
    """audio_test.py:522-526: 20 log10 |mean / std| (population std); 0-std input gives -inf as numpy's log10(0)."""
    a = a.reshape(-1).double()
    sd = a.std(unbiased=False)
    m = a.mean()
    r = torch.where(sd == 0, torch.zeros_like(m), m / sd).abs()
    return 20.0 * math.log10(r.item()) if r.item() > 0 else float("-inf")


def _mse(a, b, out=None):
    """mean((a - b)^2) into a device scalar (no host synchronisation here)."""
    loss = torch.zeros((), device=a.device) if out is None else out
    ops.mse_fwd_bwd(a.contiguous().float(), b.contiguous().float(), loss)
    return loss


def wave_lowpass(wave):
    """Butterworth-8 zero-phase low-pass at half band (audio_attack.py:21-30) of [U, L] waveforms of any length."""
    from .attacks import _device_tables, LP_TAPS
    from ._lib import check, lib, ptr
    wave = wave.contiguous().float()
    U, L = wave.shape
    lp, _, _ = _device_tables(wave.device)
    out, tmp = torch.empty_like(wave), torch.empty(U, L + 54, device=wave.device)
    check(lib.wm_wave_lowpass(ptr(wave), ptr(out), ptr(tmp), L, U, ptr(lp), LP_TAPS, torch.cuda.current_stream().cuda_stream),
          "wm_wave_lowpass")
    return out


def wave_resample(wave):
    """16 kHz -> 8 kHz -> 16 kHz (audio_attack.py:71-83) of [U, L] waveforms: polyphase half-band Kaiser filter, scipy
    resample_poly conventions, output length L (the reference's librosa kernel is parity-unpinned)."""
    from .attacks import _device_tables
    from ._lib import check, lib, ptr
    wave = wave.contiguous().float()
    U, L = wave.shape
    _, rs, _ = _device_tables(wave.device)
    out, tmp = torch.empty_like(wave), torch.empty(U, (L + 1) // 2, device=wave.device)
    check(lib.wm_wave_resample(ptr(wave), ptr(out), ptr(tmp), L, U, ptr(rs), 41, torch.cuda.current_stream().cuda_stream),
          "wm_wave_resample")
    return out


def time_stretch(wave, rate, length=None):
    """librosa.effects.time_stretch(y, rate) on [U, L] waveforms: phase vocoder (n_fft 2048, hop 512, Hann); output length
    round(L / rate) unless `length` is given.  librosa semantics restated: parity unpinned."""
    from ._lib import check, lib, ptr
    wave = wave.contiguous().float()
    U, L = wave.shape
    T, Ts = int(lib.wm_pv_frames(L)), int(lib.wm_pv_steps(L, float(rate)))
    Lout = int(round(L / rate)) if length is None else int(length)
    spec = torch.empty(U, 1025, T + 2, 2, device=wave.device)
    st = torch.empty(U, 1025, Ts, 2, device=wave.device)
    frames = torch.empty(U, Ts, 2048, device=wave.device)
    out = torch.empty(U, Lout, device=wave.device)
    check(lib.wm_time_stretch(ptr(wave), L, U, float(rate), ptr(spec), ptr(st), ptr(frames), ptr(out), Lout,
                              torch.cuda.current_stream().cuda_stream), "wm_time_stretch")
    return out


_sinc_tables = {}


def _kaiser_best(device):
    """resampy 'kaiser_best' half filter and its forward differences (64 zero crossings x 512 samples), built with numpy."""
    key = str(device)
    if key not in _sinc_tables:
        import numpy as np
        bits, zeros, rolloff, beta = 512, 64, 0.9475937167399596, 14.769656459379492
        n = bits * zeros
        t = np.linspace(0, zeros, num=n + 1, endpoint=True)
        win = (rolloff * np.sinc(rolloff * t) * np.kaiser(2 * n + 1, beta)[n:]).astype(np.float32)
        delta = np.zeros_like(win)
        delta[:-1] = np.diff(win)
        _sinc_tables[key] = (torch.from_numpy(win).to(device), torch.from_numpy(delta).to(device), bits)
    return _sinc_tables[key]


def pitch_shift(wave, n_steps=-6.0, bins_per_octave=12):
    """librosa.effects.pitch_shift(y, sr, n_steps) on [U, L] waveforms: time stretch by 2^(-n_steps / 12), band-limited
    resampling by the same ratio, length fixed to L.  Parity unpinned (librosa / resampy are not in the reference tree)."""
    from ._lib import check, lib, ptr
    import math
    wave = wave.contiguous().float()
    U, L = wave.shape
    rate = 2.0 ** (-float(n_steps) / bins_per_octave)
    ys = time_stretch(wave, rate)
    win, delta, bits = _kaiser_best(wave.device)
    out = torch.empty(U, L, device=wave.device)                  # fix_length(resampled, L): zero beyond ceil(len * rate)
    n = int(math.ceil(ys.shape[1] * rate))
    check(lib.wm_sinc_resample(ptr(ys), ys.shape[1], U, rate, ptr(win), ptr(delta), win.numel(), bits, ptr(out), L,
                               torch.cuda.current_stream().cuda_stream), "wm_sinc_resample")
    if n < L:
        out[:, n:] = 0
    return out


def _attack(wave, attack, seed):
    name, _, param = (attack or "closed_loop").partition("-")
    if name in ("closed_loop", "no", ""):
        return wave
    if name == "amplitude_scaling":
        return ops.wave_attack(wave, amp=float(param) if param else 0.8)
    if name == "awgn":
        return ops.wave_attack(wave, snr_db=float(param), seed=seed)
    if name == "echo_addition":                                           # audio_attack.py:33-52: td = 0.5 s, AA = 0.2
        return ops.wave_attack(wave, echo=(8000, 0.2))
    if name == "jittering_2":                                             # :176-193, audio_test.py:652
        return ops.wave_attack(wave, jitter=int(param) if param else 1000, seed=seed)
    if name == "requantization":                                          # :85-96 (PCM_U8 round trip)
        return ops.wave_attack(wave, requantize=True)
    if name in ("low_pass", "lowpass", "low_pass_filter"):                # audio_test.py:637 -> audio_attack.py:21-30
        return wave_lowpass(wave)
    if name == "resampling":                                              # audio_test.py:644 -> audio_attack.py:71-83
        return wave_resample(wave)
    if name == "time_scaling":                                            # audio_test.py:654 -> audio_attack.py:195-231
        return time_stretch(wave, float(param))
    if name == "pitch_scaling":                                           # audio_test.py:656 -> audio_attack.py:233-239
        return pitch_shift(wave, -6.0)
    raise NotImplementedError(f"attack '{name}' is a CPU tool of audio_attack.py outside the MI355X path (DESIGN.md, out of scope)")


def _snr_rows(w):
    """signaltonoise of every row of [U, L] as a device tensor of |mean / std| (converted to dB on the host)."""
    w = w.double()
    sd = w.std(dim=1, unbiased=False)
    return torch.where(sd == 0, torch.zeros_like(sd), w.mean(dim=1) / sd).abs()


def _db(r):
    return 20.0 * math.log10(r) if r > 0 else float("-inf")


@torch.no_grad()
def _run_model(model, model_name, clips, msg, batch):
    """audio_test.py:547-556, batched over tiles: (encoded clips, decoded payloads)."""
    enc, dec = [], []
    for i in range(0, clips.shape[0], batch):
        c = clips[i:i + batch].contiguous()
        m = msg.expand(c.shape[0], 1, 32, 32).contiguous()
        if model_name == "uformer":
            audio_clip, _, _, wm_decode = model(c, m)
        else:                                                              # 'modelA' convention: (encoded, extracted)
            audio_clip, wm_decode = model(c, m)
        enc.append(audio_clip)
        dec.append(wm_decode)
    return torch.cat(enc).contiguous(), torch.cat(dec)


def _decode(model, c):
    """audio_test.py:706-708, :763-766: ``wm_decode`` where the model has one, else ``decode``."""
    return model.wm_decode(c) if hasattr(model, "wm_decode") else model.decode(c)


def _reconstruct_audio_mode(audio_datas, watermark, model, attack, model_name, batch, seed, audio_scale="0", data_min=None,
                            data_max=None):
    """``data_mode='audio'`` (audio_test.py:581-583, :601-606, :665-670, :733-772): raw-sample clips [n, 1, 128, 128].  The encoded
    clips go back to the audio value range exactly as in 'stft' mode (audio_test.py:559-571 sits in front of the data_mode
    branches); the attacked samples are handed to the extractor as they are (:733-760 applies no forward scale)."""
    from .dataprep import parse_audio_scale, _stats
    mode = parse_audio_scale(audio_scale)
    if mode is not None and mode[0] == "range" and (data_min is None or data_max is None):
        raise ValueError("range-style audio_scale needs the training set's data_min / data_max (audio_test.py:567-571)")
    U = len(audio_datas)
    L, n, len_last = audio_datas[0][0].shape[-1], audio_datas[0][1].shape[0], int(audio_datas[0][2])
    if any(d[0].shape[-1] != L for d in audio_datas):
        raise ValueError("reconstruct_audio_batch takes utterances of equal length")
    waves = torch.cat([d[0].reshape(1, -1) for d in audio_datas]).contiguous().float()
    msg = watermark.reshape(1, 1, 32, 32).float()
    enc, dec = _run_model(model, model_name, torch.cat([d[1] for d in audio_datas]).contiguous(), msg, batch)
    if mode is not None and mode[0] == "range":                           # :567-571
        enc = ops.range_map(enc, _stats(data_min, data_max, waves.device), mode[1], mode[2], inverse=True)
    elif mode is not None and mode[0] == "factor":                        # :562-564
        scaled = torch.empty_like(enc)
        ops.axpbyz(scaled, 1.0 / mode[1], enc, 0.0, enc, enc.numel())
        enc = scaled
    recon_audio = enc.reshape(U, n * 128 * 128)[:, :(n - 1) * 128 * 128 + len_last].contiguous()       # last clip trimmed
    audio_att = _attack(recon_audio, attack, seed)
    La = audio_att.shape[-1]
    len_pad = 128 * 128 - La % (128 * 128)
    att_clips = torch.nn.functional.pad(audio_att.float(), (0, len_pad)).reshape(-1, 1, 128, 128).contiguous()
    n_att = att_clips.shape[0] // U
    dec_att = torch.cat([_decode(model, att_clips[i:i + batch].contiguous()) for i in range(0, att_clips.shape[0], batch)])
    return waves, recon_audio, audio_att, dec, dec_att, n, n_att


@torch.no_grad()
def reconstruct_audio_batch(audio_datas, watermark, model, n_fft=255, attack=None, data_mode="stft", audio_scale="0",
                            model_name="uformer", batch=96, seed=0, data_min=None, data_max=None):
    """``reconstruct_audio`` for several utterances of EQUAL length in one pass: their tiles share the model batches
    (the model is 1.5 x more efficient at 96 tiles than at 20), the transforms and attacks take [U, L] waveforms, and
    all scalar metrics come back in one host read.  Returns one reference-style 10-tuple per utterance."""
    from .dataprep import parse_audio_scale, _stats
    if data_mode == "audio":
        waves, recon_audio, audio_att, dec, dec_att, n, n_att = _reconstruct_audio_mode(audio_datas, watermark, model, attack,
                                                                                        model_name, batch, seed, audio_scale,
                                                                                        data_min, data_max)
        return _metrics(waves, recon_audio, audio_att, watermark, dec, dec_att, n, n_att)
    if data_mode != "stft" or n_fft != 255:
        raise ValueError("the MI355X evaluation path covers data_mode='stft' (n_fft=255, audio_test.py:594-600) and 'audio'; "
                         "'dwt' needs pywt, a CPU library outside the hot path")
    mode = parse_audio_scale(audio_scale)
    ranged = mode is not None and mode[0] == "range"
    if ranged and (data_min is None or data_max is None):
        raise ValueError("range-style audio_scale needs the training set's data_min / data_max (audio_test.py:567-571)")
    U = len(audio_datas)
    L, n, len_last_clip = audio_datas[0][0].shape[-1], audio_datas[0][1].shape[0], int(audio_datas[0][2])
    if any(d[0].shape[-1] != L or d[1].shape[0] != n or int(d[2]) != len_last_clip for d in audio_datas):
        raise ValueError("reconstruct_audio_batch takes utterances of equal length")
    waves = torch.cat([d[0].reshape(1, -1) for d in audio_datas]).contiguous()
    clips = torch.cat([d[1] for d in audio_datas]).contiguous()
    msg = watermark.reshape(1, 1, 32, 32).float()
    if model_name not in ("uformer", "modelA"):
        raise ValueError(f"unknown model_name '{model_name}'")
    enc, dec = _run_model(model, model_name, clips, msg, batch)
    if ranged:                                                            # :567-571: back to the spectrogram's own range
        mm = _stats(data_min, data_max, waves.device)
        enc = ops.range_map(enc, mm, mode[1], mode[2], inverse=True)
    factor = 1.0 / mode[1] if (mode is not None and mode[0] == "factor") else 1.0     # :559-566
    recon_audio = ops.istft255_stitch(enc, n, len_last_clip, L, factor)   # :594-600, [U, L]
    audio_att = _attack(recon_audio, attack, seed)                        # :628-660
    scale = mode[1] if (mode is not None and mode[0] == "factor") else 1.0
    att_clips, _ = ops.stft255_tiles(audio_att.contiguous(), scale)       # :676-700
    if ranged:
        att_clips = ops.range_map(att_clips, mm, mode[1], mode[2])        # :698-702
    n_att = att_clips.shape[0] // U
    dec_att = torch.cat([model.wm_decode(att_clips[i:i + batch].contiguous()) if model_name == "uformer"
                         else model.decode(att_clips[i:i + batch].contiguous()) for i in range(0, att_clips.shape[0], batch)])
    return _metrics(waves, recon_audio, audio_att, watermark, dec, dec_att, n, n_att)


def _metrics(waves, recon_audio, audio_att, watermark, dec, dec_att, n, n_att):
    """The reference's 10-tuple per utterance; all scalar metrics in one host read."""
    U = waves.shape[0]
    msg = watermark.reshape(1, 1, 32, 32).float()
    scal = torch.zeros(U, 2 + n_att, device=waves.device)                 # mse_loss | wm_loss | wm_losses_att...
    for u in range(U):
        _mse(waves[u:u + 1], recon_audio[u:u + 1], scal[u, 0])            # :618
        _mse(msg, dec[(u + 1) * n - 1:(u + 1) * n], scal[u, 1])           # :625 (the reference keeps the last tile's)
        for j in range(n_att):
            _mse(msg, dec_att[u * n_att + j:u * n_att + j + 1], scal[u, 2 + j])          # :706
    host = torch.cat([scal.double(), _snr_rows(waves)[:, None], _snr_rows(recon_audio)[:, None]], dim=1).tolist()
    out = []
    for u in range(U):
        h = host[u]
        out.append((audio_att[u], recon_audio[u], watermark, [dec[u * n + j:u * n + j + 1] for j in range(n)],
                    [dec_att[u * n_att + j:u * n_att + j + 1] for j in range(n_att)], h[0], h[1],
                    sum(h[2:2 + n_att]) / n_att, _db(h[-2]), _db(h[-1])))
    return out


@torch.no_grad()
def reconstruct_audio(audio_data, watermark, model, n_fft=255, attack=None, data_mode="stft", audio_scale="0",
                      data_min=None, data_max=None, model_name="uformer", batch=32, seed=0):
    """reconstruct_audio of audio_test.py:528-785 for ``data_mode='stft'`` and ``'audio'``.

    Returns ``(audio_att, recon_audio, watermark, wms_decode, wms_att_decode, mse_loss, wm_loss, wm_loss_att,
    snr_ori, snr_recon)``: tensors stay on the device, the per-tile lists hold ``[1, 1, 32, 32]`` tensors."""
    return reconstruct_audio_batch([audio_data], watermark, model, n_fft=n_fft, attack=attack, data_mode=data_mode,
                                   audio_scale=audio_scale, model_name=model_name, batch=batch, seed=seed,
                                   data_min=data_min, data_max=data_max)[0]
