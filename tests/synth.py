"""Deterministic synthetic weights / inputs shared by the golden generator and every test.

Weights are *name-keyed*: each tensor of a ``state_dict`` is filled from a numpy Philox stream
seeded by crc32(name), so the reference, the oracle and the HIP product receive bit-identical
values regardless of construction order.  Scales are chosen "busy" (every branch contributes
O(1) to the residual stream) so that an error in any kernel is visible end to end.
"""
import zlib

import numpy as np
import torch


def _rng(name, salt=0):
    return np.random.Generator(np.random.Philox(key=zlib.crc32(name.encode()) + (salt << 32)))


def synth_tensor(name, like):
    shape = tuple(like.shape)
    g = _rng(name)
    parts = name.split(".")
    leaf = parts[-1]
    if leaf in ("relative_position_index", "num_batches_tracked"):
        return like.clone()
    if leaf == "running_mean":
        v = 0.1 * g.standard_normal(shape)
    elif leaf == "running_var":
        v = 1.0 + 0.2 * g.random(shape)
    elif len(shape) == 1 and leaf == "weight":
        v = 1.0 + 0.1 * g.standard_normal(shape)            # LayerNorm / BatchNorm gain
    elif leaf == "relative_position_bias_table" or name.endswith("modulator.weight"):
        v = 0.5 * g.standard_normal(shape)
    elif leaf == "bias":
        v = 0.1 * g.standard_normal(shape)
    else:
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        if "deconv" in name or "t_conv" in name or "embedder_decoder" in name:
            fan_in = shape[0]      # ConvTranspose2d weight is [Cin,Cout,kh,kw]: each output sees Cin taps
        v = g.standard_normal(shape) * (0.8 / np.sqrt(max(fan_in, 1)))
    if name.startswith("output_proj."):
        v = v * 0.015      # keeps ||noise||/B = O(1) so loss4 does not drown the other gradient paths
    return torch.from_numpy(np.asarray(v, dtype=np.float32)).reshape(shape)


def synth_state_dict(module):
    sd = module.state_dict()
    return {k: synth_tensor(k, v) if v.dtype.is_floating_point else v.clone() for k, v in sd.items()}


def load_synth(module):
    module.load_state_dict(synth_state_dict(module))
    return module


def synth_array(tag, shape, scale=1.0):
    return torch.from_numpy((scale * _rng(tag, 1).standard_normal(shape)).astype(np.float32))


def synth_bits(tag, shape):
    return torch.from_numpy(_rng(tag, 2).integers(0, 2, shape).astype(np.float32))


def synth_tile(batch, tag="tile", scale=0.3):
    """Stand-in STFT tile [B,2,128,128] (direct tiles for kernel-level checks)."""
    return synth_array(tag, (batch, 2, 128, 128), scale)


def chirp_wave(length, tag="chirp"):
    """SURVEY 8(d) synthetic speech stand-in: 3 harmonically related chirps + low-level noise."""
    g = _rng(tag, 3)
    t = np.arange(length) / 16000.0
    f0 = 120.0 + 60.0 * g.random()
    w = np.zeros(length)
    for h in (1, 2, 3):
        ph = 2 * np.pi * g.random()
        w += np.sin(2 * np.pi * h * (f0 * t + 40.0 * t * t) + ph) / h
    w += 10 ** (-30 / 20) * g.standard_normal(length)
    return torch.from_numpy((0.5 * w / np.abs(w).max()).astype(np.float32))
