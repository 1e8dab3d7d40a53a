"""Data-side mirror of the reference's dataset preparation and of the config-5 payload tiling.

* ``stft_tiles``: ``SpeechDataTrain.prepare_data`` / ``__getitem__`` (audio_test.py:439-519): utterance ->
  STFT(256/128/256 rectangular, centre, reflect) -> drop bin 128 -> zero-pad frames -> ``[2,128,128]`` tiles, on the
  device (``wm_stft256_tiles``), optionally scaled by ``audio_scale`` (audio_test.py:33-55 applies a global factor).
* ``split_payload`` / ``merge_payload``: config 5 carries a 64x64 grayscale payload.  The model's payload path is
  hard-wired to 32x32 (model.py:2250,2389,2400,1582), so the image is cut into four 32x32 quadrants that ride on
  successive tiles of the same clip (tile t carries quadrant t % 4; a 10 s clip = 10 tiles carries every quadrant at
  least twice).  Pure tiling: every tile stays an ordinary [2,128,128] / [1,32,32] pair the oracle can check.
"""
import torch

from . import ops


def stft_tiles(wave, scale=1.0):
    """wave: float32 CUDA tensor [U, L] (or [L]) of equal-length utterances -> ([U * n, 2, 128, 128], n tiles each)."""
    if wave.dim() == 1:
        wave = wave[None]
    tiles = ops.stft256_tiles(wave.contiguous(), scale)
    return tiles, tiles.shape[0] // wave.shape[0]


def parse_audio_scale(audio_scale):
    """The reference's string option (options.py --audio_scale; audio_test.py:40-47): '0' / one character -> None (leave the
    tiles alone), a plain number of more than one character -> ('factor', f), 'lo-hi' -> ('range', lo, hi)."""
    audio_scale = str(audio_scale)
    if "-" in audio_scale:
        lo, hi = audio_scale.split("-")
        return ("range", float(lo), float(hi))
    return ("factor", float(audio_scale)) if len(audio_scale) > 1 else None


def _stats(data_min, data_max, device):
    """data_min / data_max as the reference passes them around (python floats, 0-d or 1-element tensors) -> [min, max] on device."""
    if torch.is_tensor(data_min) and data_min.numel() == 2 and data_max is None:
        return data_min.to(device).float().contiguous()
    f = lambda v: v.reshape(-1)[:1].to(device).float() if torch.is_tensor(v) else torch.tensor([float(v)], device=device)
    return torch.cat([f(data_min), f(data_max)]).contiguous()


def normalize_tiles(tiles, audio_scale, data_min=None, data_max=None):
    """normalize_batch of audio_test.py:33-55 on the device: returns (tiles', data_min, data_max).  Range form: min / max over
    the WHOLE tensor (unless given: the evaluation sets pass the training set's, audio_test.py:338), then the affine map into
    [lo, hi]; factor form: tiles * f; both statistics come back as 1-element device tensors (no host synchronisation)."""
    mode = parse_audio_scale(audio_scale)
    tiles = tiles.contiguous().float()
    mm = ops.minmax(tiles) if data_min is None else _stats(data_min, data_max, tiles.device)
    if mode is None:
        out = tiles
    elif mode[0] == "factor":
        out = torch.empty_like(tiles)
        ops.axpbyz(out, mode[1], tiles, 0.0, None, tiles.numel())
    else:
        out = ops.range_map(tiles, mm, mode[1], mode[2])
    return out, mm[0:1], mm[1:2]


def split_payload(img):
    """[B, 1, 64, 64] -> [B, 4, 1, 32, 32]: quadrants in row-major order (top-left, top-right, bottom-left, bottom-right)."""
    if img.dim() != 4 or img.shape[1:] != (1, 64, 64):
        raise ValueError("split_payload takes [B,1,64,64] payload images")
    B = img.shape[0]
    return img.reshape(B, 1, 2, 32, 2, 32).permute(0, 2, 4, 1, 3, 5).reshape(B, 4, 1, 32, 32)


def merge_payload(quads):
    """Inverse of split_payload: [B, 4, 1, 32, 32] -> [B, 1, 64, 64]."""
    if quads.dim() != 5 or quads.shape[1:] != (4, 1, 32, 32):
        raise ValueError("merge_payload takes [B,4,1,32,32] quadrants")
    B = quads.shape[0]
    return quads.reshape(B, 2, 2, 1, 32, 32).permute(0, 3, 1, 4, 2, 5).reshape(B, 1, 64, 64)


def payload_for_tiles(img, tiles_per_clip):
    """Per-tile 32x32 payloads of a batch of clips: tile t of a clip carries quadrant t % 4.
    [B,1,64,64] -> [B * tiles_per_clip, 1, 32, 32] (clip-major, the order stft_tiles emits)."""
    q = split_payload(img)
    idx = torch.arange(tiles_per_clip, device=img.device) % 4
    return q[:, idx].reshape(-1, 1, 32, 32)


def payload_from_tiles(decoded, tiles_per_clip):
    """Average the decoded quadrants of each clip's tiles back into [B,1,64,64] (every quadrant that was carried more than
    once is averaged before thresholding, as the reference averages repeated payload copies, audio_test.py:686-712)."""
    B = decoded.shape[0] // tiles_per_clip
    d = decoded.reshape(B, tiles_per_clip, 1, 32, 32)
    quads = torch.stack([d[:, q::4].mean(1) for q in range(4)], 1) if tiles_per_clip >= 4 else None
    if quads is None:
        raise ValueError("a clip needs at least 4 tiles to carry a 64x64 payload")
    return merge_payload(quads)
