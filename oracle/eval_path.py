"""CPU restatement (PyTorch fp32 / numpy) of the whole-utterance evaluation path, ``data_mode='stft'`` and ``'audio'``.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows /root/reference/uformerWM/audio_test.py:
  prepare_utterance  :299-343  SpeechDataTest.prepare_data — torch.stft(n_fft=255), pad frames to a multiple of 128
                               (always at least one padded column), split into [2,128,128] tiles, scale
  stitch_istft       :594-600  concatenate the encoded tiles (last trimmed to len_last_clip), torch.istft(n_fft=255,
                               length=L)
  signaltonoise      :522-526
  reconstruct_audio  :528-785  the loop itself: encode tile by tile, stitch, attack, re-analyse, decode
  normalize_batch    :33-55    audio_scale forms: '0', a factor, 'lo-hi' (global min / max of the tensor)
  prepare_utterance_audio :135-153  SpeechDataAudio.prepare_data (raw samples, [1,128,128] clips)
The transforms are the same torch.stft / torch.istft calls the reference makes (modern complex API with the rectangular
window explicit, as in oracle/uformer_wm.py); pinned by the round trip test in tests/test_oracle_golden.py.
"""
import numpy as np
import torch
import torch.nn.functional as F

N_FFT = 255


def _stft(wave):
    spec = torch.stft(wave, n_fft=N_FFT, window=torch.ones(N_FFT), return_complex=True)      # hop 63, centre, reflect
    return torch.view_as_real(spec)                                                          # [1, 128, T, 2]


def normalize_batch(image, audio_scale):
    """audio_test.py:33-55 (global min / max; the reference's per-channel view only works for one channel group)."""
    mn, mx = image.min(), image.max()
    if "-" not in audio_scale:
        out = image * float(audio_scale) if len(audio_scale) > 1 else image
    else:
        lo, hi = (float(v) for v in audio_scale.split("-"))
        out = (image - mn) / (mx - mn) * (hi - lo) + lo
    return out, mn, mx


def prepare_utterance(wave, audio_scale="0", data_min=None, data_max=None):
    wave = wave.reshape(1, -1).float()
    stft = _stft(wave)
    len_pad = 128 - stft.shape[2] % 128
    stft_2 = F.pad(stft, (0, 0, 0, len_pad))
    clips = []
    for j in range(stft_2.shape[2] // 128):
        clip = stft_2[:, :, 128 * j:128 * (j + 1), :]
        if "-" not in audio_scale:
            if len(audio_scale) > 1:
                clip = clip * float(audio_scale)
        else:                                                                                # :331-338
            lo, hi = (float(v) for v in audio_scale.split("-"))
            clip = (clip - data_min) / (data_max - data_min) * (hi - lo) + lo
        clips.append(clip.permute(0, 3, 1, 2).squeeze(0))
    return [wave, torch.stack(clips), stft.shape[2] % 128]


def prepare_utterance_audio(wave):
    """audio_test.py:135-153: zero-pad to a multiple of 128 * 128 samples (a full clip when it already divides), cut."""
    wave = wave.reshape(1, -1).float()
    len_pad = 128 ** 2 - wave.shape[1] % 128 ** 2
    padded = F.pad(wave, (0, len_pad))
    clips = [padded[:, 128 ** 2 * i:128 ** 2 * (i + 1)].reshape(1, 128, 128) for i in range(padded.shape[1] // 128 ** 2)]
    return [wave, torch.stack(clips), 128 ** 2 - len_pad]


def stitch_istft(tiles, len_last_clip, L):
    """tiles [n, 2, 128, 128] -> waveform [L] (audio_test.py:594-600)."""
    n = tiles.shape[0]
    preds = [tiles[i:i + 1] for i in range(n - 1)] + [tiles[n - 1:n, :, :, :len_last_clip]]
    spec = torch.cat(preds, dim=3).squeeze(0).permute(1, 2, 0).contiguous()                   # [128, T, 2]
    return torch.istft(torch.view_as_complex(spec), n_fft=N_FFT, window=torch.ones(N_FFT), length=L)


def signaltonoise(a):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    m, sd = a.mean(), a.std()
    with np.errstate(divide="ignore"):
        return 20 * np.log10(abs(np.where(sd == 0, 0, m / sd)))


def reconstruct_audio_raw(audio_data, watermark, model, attack_fn=None):
    """``data_mode='audio'`` (audio_test.py:581-583, :601-606, :665-670, :733-772); model: callable (clip [1,1,128,128], wm) ->
    (encoded, extracted) with .decode."""
    wave, clips, len_last_clip = audio_data
    preds, wms_decode = [], []
    with torch.no_grad():
        for i in range(clips.shape[0]):
            audio_clip, wm_decode = model(clips[i:i + 1], watermark)
            wms_decode.append(wm_decode)
            flat = audio_clip.reshape(audio_clip.shape[0], 1, -1)
            preds.append(flat if i != clips.shape[0] - 1 else flat[:, :, :len_last_clip])
        recon = torch.cat(preds, dim=2).reshape(-1)
        mse_loss = F.mse_loss(wave.reshape(-1), recon).item()
        wm_loss = F.mse_loss(watermark, wms_decode[-1]).item()
        att = recon if attack_fn is None else attack_fn(recon)
        feat = F.pad(att.reshape(-1).float(), (0, 128 ** 2 - att.numel() % 128 ** 2)).reshape(1, 1, -1)
        wms_att, losses_att = [], []
        for j in range(feat.shape[2] // 128 ** 2):
            d = model.decode(feat[:, :, 128 ** 2 * j:128 ** 2 * (j + 1)].reshape(1, 1, 128, 128))
            wms_att.append(d)
            losses_att.append(F.mse_loss(watermark, d).item())
    return (att, recon, watermark, wms_decode, wms_att, mse_loss, wm_loss, float(np.mean(losses_att)),
            signaltonoise(wave.numpy()), signaltonoise(recon.numpy()))


def reconstruct_audio(audio_data, watermark, model, attack_fn=None, audio_scale="0", data_min=None, data_max=None):
    """model: callable (clip [1,2,128,128], wm [1,1,32,32]) -> (stft_new, noise, wm_pred, wm) with .wm_decode;
    attack_fn: waveform [L] -> waveform (None = closed loop)."""
    wave, clips, len_last_clip = audio_data
    L = wave.shape[-1]
    ranged = "-" in audio_scale
    if ranged:
        lo, hi = (float(v) for v in audio_scale.split("-"))
    enc, wms_decode = [], []
    with torch.no_grad():
        for i in range(clips.shape[0]):
            audio_clip, _, _, wm_decode = model(clips[i:i + 1], watermark)
            if ranged:                                                                       # :567-571
                audio_clip = (audio_clip - lo) / (hi - lo) * (data_max - data_min) + data_min
            elif len(audio_scale) > 1:
                audio_clip = audio_clip * (1 / float(audio_scale))
            enc.append(audio_clip)
            wms_decode.append(wm_decode)
        recon = stitch_istft(torch.cat(enc), len_last_clip, L)
        mse_loss = F.mse_loss(wave.reshape(-1), recon).item()
        wm_loss = F.mse_loss(watermark, wms_decode[-1]).item()
        att = recon if attack_fn is None else attack_fn(recon)
        feat = _stft(att.reshape(1, -1).float())
        feat = F.pad(feat, (0, 0, 0, 128 - feat.shape[2] % 128)).squeeze(0).permute(2, 0, 1).unsqueeze(0)
        wms_att, losses_att = [], []
        for j in range(feat.shape[3] // 128):
            clip = feat[:, :, :, 128 * j:128 * (j + 1)].float()
            if ranged:                                                                       # :698-702
                clip = (clip - data_min) / (data_max - data_min) * (hi - lo) + lo
            elif len(audio_scale) > 1:
                clip = clip * float(audio_scale)
            d = model.wm_decode(clip)
            wms_att.append(d)
            losses_att.append(F.mse_loss(watermark, d).item())
    return (att, recon, watermark, wms_decode, wms_att, mse_loss, wm_loss, float(np.mean(losses_att)),
            signaltonoise(wave.numpy()), signaltonoise(recon.numpy()))
