"""CPU restatement (plain torch, deterministic: the drawn parameters are arguments) of the reference's HiDDeN noise layers.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows /root/reference/hidden/noise_layers/: crop.py:48-74, cropout.py:6-27,
dropout.py:5-29, resize.py:6-26, quantization.py:17-45, jpeg_compression.py:65-160.  PINNED: tests/test_oracle_golden.py
holds every function to tests/golden/g7_noise_layers.npz, which tests/golden/make_golden_noise.py generated from the
reference's own modules.  Used for shapes the fixtures do not cover.
"""
import numpy as np
import torch
import torch.nn.functional as F


def crop(x, h0, h1, w0, w1):
    return x[:, :, h0:h1, w0:w1].clone()


def cropout(noised, cover, h0, h1, w0, w1):
    m = torch.zeros_like(noised)
    m[:, :, h0:h1, w0:w1] = 1
    return noised * m + cover * (1 - m)


def dropout(noised, cover, mask):
    m = mask.to(noised.dtype).expand_as(noised)
    return noised * m + cover * (1 - m)


def resize(x, ratio):
    return F.interpolate(x, scale_factor=(ratio, ratio), mode="nearest")


def quantization(x):
    def transform(t, lo, hi):
        return (t - t.min()) / (t.max() - t.min()) * (hi - lo) + lo
    n = torch.arange(10, dtype=torch.float64)
    w = (((-1.0) ** (n + 1)) / (np.pi * (n + 1))).float().view(10, 1, 1, 1, 1)
    s = (2 * np.pi * (n + 1)).float().view(10, 1, 1, 1, 1)
    t = transform(x, 0, 255).clamp(0.0, 255.0)
    t = t + torch.sum(w * torch.sin(t * s), dim=0)
    return transform(t, x.min(), x.max())


def _zigzag_mask(keep):
    order = sorted(((a, b) for a in range(8) for b in range(8)), key=lambda p: (p[0] + p[1], -p[1] if (p[0] + p[1]) % 2 else p[1]))
    m = np.zeros((8, 8), dtype=np.float32)
    for i, j in order[:keep]:
        m[i, j] = 1
    return torch.from_numpy(m)


def jpeg(x, keep=(25, 9, 9)):
    B, C, H, W = x.shape
    ph, pw = (8 - H % 8) % 8, (8 - W % 8) % 8
    xp = F.pad(x, (0, pw, 0, ph))
    rgb2yuv = torch.tensor([[0.299, 0.587, 0.114], [-0.14713, -0.28886, 0.436], [0.615, -0.51499, -0.10001]], dtype=x.dtype)
    yuv2rgb = torch.tensor([[1.0, 0.0, 1.13983], [1.0, -0.39465, -0.58060], [1.0, 2.03211, 0.0]], dtype=x.dtype)
    yuv = torch.einsum("cq,bqhw->bchw", rgb2yuv, xp)
    k = torch.arange(8, dtype=torch.float64)
    D = torch.cos(np.pi / 8 * (k[None, :] + 0.5) * k[:, None]).to(x.dtype)                     # D[k][n]
    I = ((torch.cos(np.pi / 8 * (k[:, None] + 0.5) * k[None, :]) - 0.5 * (k[None, :] == 0)) * 0.25).to(x.dtype)   # I[pixel][coef]
    blk = yuv.view(B, 3, (H + ph) // 8, 8, (W + pw) // 8, 8).permute(0, 1, 2, 4, 3, 5)         # [B,3,Hb,Wb,8,8]
    coef = D @ blk @ D.t()
    mask = torch.stack([_zigzag_mask(kc) for kc in keep]).to(x.dtype)[None, :, None, None]
    rec = I @ (coef * mask) @ I.t()
    rec = rec.permute(0, 1, 2, 4, 3, 5).reshape(B, 3, H + ph, W + pw)
    return torch.einsum("cq,bqhw->bchw", yuv2rgb, rec)[:, :, :H, :W]
