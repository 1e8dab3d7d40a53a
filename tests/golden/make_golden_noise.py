"""Golden vectors for SURVEY 8f row N4: outputs and input gradients of the reference's OWN hidden/noise_layers modules
(imported from /root/reference/hidden in the build container; they are plain torch + numpy and import as they are),
numpy's global generator seeded before every call so that the product's host mirror reproduces the same draws.

  python tests/golden/make_golden_noise.py        ->  tests/golden/g7_noise_layers.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/hidden")
from synth import synth_array  # noqa: E402
from noise_layers.crop import Crop  # noqa: E402
from noise_layers.cropout import Cropout  # noqa: E402
from noise_layers.dropout import Dropout  # noqa: E402
from noise_layers.resize import Resize  # noqa: E402
from noise_layers.quantization import Quantization  # noqa: E402
from noise_layers.jpeg_compression import JpegCompression  # noqa: E402
from noise_layers.noiser import Noiser  # noqa: E402

out = {}
CASES = [  # name, layer factory, channels, H, W, seed
    ("crop", lambda: Crop((0.4, 0.7), (0.5, 0.8)), 2, 40, 56, 11),
    ("cropout", lambda: Cropout((0.3, 0.6), (0.4, 0.6)), 2, 40, 56, 12),
    ("dropout", lambda: Dropout((0.25, 0.45)), 2, 40, 56, 13),
    ("resize_down", lambda: Resize((0.45, 0.75)), 2, 40, 56, 14),
    ("resize_up", lambda: Resize((1.3, 1.7)), 1, 24, 20, 15),
    ("quant", lambda: Quantization(), 2, 40, 56, 16),
    ("jpeg", lambda: JpegCompression(torch.device("cpu")), 3, 40, 52, 17),        # 52 is not a multiple of 8: padding path
    ("jpeg128", lambda: JpegCompression(torch.device("cpu"), yuv_keep_weights=(25, 9, 9)), 3, 128, 128, 18),
]
for name, make, C, H, W, seed in CASES:
    layer = make()
    noised = synth_array(f"g7_{name}_n", (2, C, H, W)).requires_grad_()
    cover = synth_array(f"g7_{name}_c", (2, C, H, W)).requires_grad_()
    np.random.seed(seed)
    if name.startswith("jpeg"):
        # jpeg_compression.py:100 unsqueeze_()s a view in place: autograd of current torch refuses its backward, so only the
        # forward is captured; the map is linear and the product's gradient is held by the adjoint identity in the test
        with torch.no_grad():
            y = layer([noised * 1.0, cover * 1.0])[0]
        out[f"{name}_y"] = y.numpy()
        print(name, tuple(y.shape), float(y.abs().max()))
        continue
    res = layer([noised * 1.0, cover * 1.0])
    y = res[0]
    g = synth_array(f"g7_{name}_g", tuple(y.shape))
    (y * g).sum().backward()
    out[f"{name}_y"] = y.detach().numpy()
    out[f"{name}_dn"] = noised.grad.numpy()
    out[f"{name}_dc"] = cover.grad.numpy() if cover.grad is not None else np.zeros_like(cover.detach().numpy())
    print(name, tuple(y.shape), float(y.abs().max()))
# Noiser: which layer gets drawn (index sequence under a seed) for a 3-layer configuration
np.random.seed(21)
ns = Noiser([Crop((0.4, 0.7), (0.5, 0.8)), "JpegPlaceholder", "QuantizationPlaceholder"], torch.device("cpu"))
picks = []
for _ in range(12):
    layer = np.random.choice(ns.noise_layers, 1)[0]
    picks.append(type(layer).__name__)
out["noiser_picks"] = np.array(picks)
np.savez_compressed(os.path.join(HERE, "g7_noise_layers.npz"), **out)
print("wrote g7_noise_layers.npz", picks)
