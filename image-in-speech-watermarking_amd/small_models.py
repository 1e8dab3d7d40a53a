"""MI355X mirrors of the two small CNN models on the hot path besides UformerAudio.

* ``ModelA`` — ``uformerWM/model.py:3000-3066`` (embedder_encoder / embedder_decoder / detector), the model
  ``train_modelA.py:423-445`` trains with ``MSE(target, encoded) + MSE(extracted, message)``.
* ``HiddenDecoder`` / ``ConvBNRelu`` — ``hidden/model/decoder.py:6-40``, ``hidden/model/conv_bn_relu.py:3-18``;
  ``Noiser`` / ``Identity`` — ``hidden/noise_layers/noiser.py:8-31``, ``identity.py:4-12`` (list-in / list-out);
  ``decoded_ber`` — ``hidden/test_model.py:60-64``.

The torch ``nn`` modules below are only CONTAINERS: they give the mirrors the reference's exact ``state_dict`` keys,
shapes and buffers (checkpoints load both ways).  Their ``forward`` is never called; every layer runs a libwm_hip kernel
(``wm_conv_*``, ``wm_batchnorm2d_*``, ``wm_maxpool_*``, ``wm_dropout``, ``wm_mse_fwd_bwd``) through one autograd node per
layer, so ``loss.backward()`` and any torch optimizer work as in the reference scripts.  NCHW fp32, CUDA tensors only.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import WmKernelError


def _need_cuda(t):
    if not t.is_cuda:
        raise WmKernelError("libwm_hip kernels take device tensors only (got a CPU tensor)")
    return t.contiguous().float()


class _Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, pad, act, transposed):
        x = _need_cuda(x)
        B, Cin, H, W = x.shape
        Cout = w.shape[1] if transposed else w.shape[0]
        d = ops.conv_desc(B, Cin, H, W, Cout, tuple(w.shape[2:]), stride, pad, False, False, act, transposed)
        y = torch.empty(B, Cout, d.OH, d.OW, device=x.device)
        ops.conv_fwd(d, x, w, b, y)
        ctx.d = d
        ctx.save_for_backward(x, w, y)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.zeros_like(w)
        db = torch.zeros(y.shape[1], device=x.device) if ctx.has_bias else None
        ops.conv_bwd(ctx.d, x, w, y, dy.contiguous(), dx, dw, db)
        return dx, dw, db, None, None, None, None


class _BatchNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, eps, momentum, training, act, slope):
        x = _need_cuda(x)
        C = x.shape[1]
        y = torch.empty_like(x)
        mean, invstd = torch.empty(C, device=x.device), torch.empty(C, device=x.device)
        ops.batchnorm2d_fwd(x, gamma, beta, rmean, rvar, y, mean, invstd, eps, momentum, training, act, slope)
        ctx.save_for_backward(x, y, gamma, mean, invstd)
        ctx.cfg = (training, act, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, mean, invstd = ctx.saved_tensors
        training, act, slope = ctx.cfg
        dx = torch.empty_like(x)
        dg, db = torch.zeros_like(gamma), torch.zeros_like(gamma)
        sums = torch.empty(2 * gamma.numel(), device=x.device)
        ops.batchnorm2d_bwd(dy.contiguous(), x, y, gamma, mean, invstd, dx, dg, db, sums, training, act, slope)
        return dx, dg, db, None, None, None, None, None, None, None


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k):
        x = _need_cuda(x)
        B, C, H, W = x.shape
        y = torch.empty(B, C, H // k, W // k, device=x.device)
        ops.maxpool_fwd(x, y, B * C, H, W, k, k)
        ctx.save_for_backward(x)
        ctx.k = k
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        B, C, H, W = x.shape
        dx = torch.empty_like(x)
        ops.maxpool_bwd(x, dy.contiguous(), dx, B * C, H, W, ctx.k, ctx.k)
        return dx, None


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, offset):
        x = _need_cuda(x)
        y = torch.empty_like(x)
        ops.dropout(x, y, p, seed, offset)
        ctx.cfg = (p, seed, offset)
        return y

    @staticmethod
    def backward(ctx, dy):
        dx = torch.empty_like(dy)
        ops.dropout(dy.contiguous(), dx, *ctx.cfg)
        return dx, None, None, None


class _CatChannels(torch.autograd.Function):
    """torch.cat([a, b], 1) for NCHW tensors as two pitched device copies (wm_copy_cols)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _need_cuda(a), _need_cuda(b)
        B, Ca, H, W = a.shape
        Cb = b.shape[1]
        out = torch.empty(B, Ca + Cb, H, W, device=a.device)
        hw = H * W
        ops.copy_cols(a, Ca * hw, 0, out, (Ca + Cb) * hw, 0, B, Ca * hw)
        ops.copy_cols(b, Cb * hw, 0, out, (Ca + Cb) * hw, Ca * hw, B, Cb * hw)
        ctx.dims = (B, Ca, Cb, H, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, Ca, Cb, H, W = ctx.dims
        hw = H * W
        dout = dout.contiguous()
        da, db = torch.empty(B, Ca, H, W, device=dout.device), torch.empty(B, Cb, H, W, device=dout.device)
        ops.copy_cols(dout, (Ca + Cb) * hw, 0, da, Ca * hw, 0, B, Ca * hw)
        ops.copy_cols(dout, (Ca + Cb) * hw, Ca * hw, db, Cb * hw, 0, B, Cb * hw)
        return da, db


class _MSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _need_cuda(a), _need_cuda(b)
        loss = torch.zeros((), device=a.device)
        ctx.save_for_backward(a, b)
        ops.mse_fwd_bwd(a, b, loss)
        return loss

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        ops.mse_fwd_bwd(a, b, None, da, db, float(g))
        return da, db


def mse_loss(a, b):
    """nn.MSELoss()(a, b) on the device (train_modelA.py:435-445)."""
    return _MSE.apply(a, b)


_ACTS = {nn.ReLU: ("relu", 0.0), nn.Sigmoid: ("sigmoid", 0.0)}


def _act_of(m):
    if isinstance(m, nn.LeakyReLU):
        return "leaky", float(m.negative_slope)
    for cls, v in _ACTS.items():
        if isinstance(m, cls):
            return v
    return None


def run_layers(seq, x, training, rng=None):
    """Execute a container ``nn.Sequential`` of Conv2d / ConvTranspose2d / BatchNorm2d / activation / MaxPool2d / Dropout
    (and nested containers) with libwm_hip kernels; an activation directly after a BatchNorm or a convolution is fused."""
    mods = list(seq.children()) if not isinstance(seq, (list, tuple)) else list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        nxt = _act_of(mods[i + 1]) if i + 1 < len(mods) else None
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            tr = isinstance(m, nn.ConvTranspose2d)
            act = "none"
            if nxt is not None and nxt[0] != "leaky":              # the conv kernel's leaky slope is fixed at 0.01
                act, i = nxt[0], i + 1
            x = _Conv.apply(x, m.weight, m.bias, tuple(m.stride), tuple(m.padding), act, tr)
        elif isinstance(m, nn.BatchNorm2d):
            act, slope = ("none", 0.0)
            if nxt is not None:
                (act, slope), i = nxt, i + 1
            use_batch = training or not m.track_running_stats
            if training and m.track_running_stats:
                m.num_batches_tracked += 1
            x = _BatchNormAct.apply(x, m.weight, m.bias, m.running_mean, m.running_var, m.eps, m.momentum, use_batch, act,
                                    slope)
        elif isinstance(m, nn.MaxPool2d):
            k = m.kernel_size if isinstance(m.kernel_size, int) else m.kernel_size[0]
            x = _MaxPool.apply(x, k)
        elif isinstance(m, nn.Dropout):
            if training and m.p > 0:
                seed, off = rng if rng is not None else (int(torch.initial_seed()) & 0xFFFFFFFF, 0)
                x = _Dropout.apply(x, float(m.p), seed, off)
        elif _act_of(m) is not None:
            raise NotImplementedError("a stand-alone activation layer (not after a conv / BatchNorm) is not on this path")
        elif len(list(m.children())):
            x = run_layers(m if not hasattr(m, "layers") else m.layers, x, training, rng)
        else:
            raise NotImplementedError(f"layer {type(m).__name__} is not part of ModelA / the HiDDeN decoder")
        i += 1
    return x


class ModelA(nn.Module):
    """model.py:3000-3066.  forward(stft, wm) -> (encoded [B,2,128,128], extracted [B,1,32,32]); .encode / .decode."""

    def __init__(self):
        super().__init__()
        lr = lambda: nn.LeakyReLU(0.2)
        self.embedder_encoder = nn.Sequential(
            nn.Conv2d(2, 16, 3, 1, 1), nn.BatchNorm2d(16), lr(), nn.MaxPool2d(2, 2),
            nn.Conv2d(16, 32, 3, 1, 1), nn.BatchNorm2d(32), lr(), nn.MaxPool2d(2, 2))
        self.embedder_decoder = nn.Sequential(
            nn.ConvTranspose2d(33, 16, 2, 2), nn.BatchNorm2d(16), nn.ReLU(), nn.Dropout(0.5),
            nn.ConvTranspose2d(16, 2, 2, 2), nn.BatchNorm2d(2), nn.Sigmoid())
        self.detector = nn.Sequential(
            nn.Conv2d(2, 16, 3, padding=1), nn.BatchNorm2d(16), lr(), nn.MaxPool2d(2, 2),
            nn.Conv2d(16, 64, 3, padding=1), nn.BatchNorm2d(64), lr(), nn.MaxPool2d(2, 2),
            nn.Conv2d(64, 1, 3, padding=1), nn.ReLU())
        self._step = 0

    def encode(self, stft, wm):
        if stft.dim() != 4 or stft.shape[1:] != (2, 128, 128) or wm.shape[1:] != (1, 32, 32):
            raise ValueError("ModelA.encode expects stft [B,2,128,128] and wm [B,1,32,32]")
        self._step += 1
        feat = run_layers(self.embedder_encoder, stft, self.training)
        return run_layers(self.embedder_decoder, _CatChannels.apply(feat, wm), self.training,
                          rng=(int(torch.initial_seed()) & 0xFFFFFFFF, self._step))

    def decode(self, x):
        return run_layers(self.detector, x, self.training)

    def forward(self, stft, wm):
        e = self.encode(stft, wm)
        return e, self.decode(e)


class ConvBNRelu(nn.Module):
    """hidden/model/conv_bn_relu.py:3-18."""

    def __init__(self, channels_in, channels_out):
        super().__init__()
        self.layers = nn.Sequential(nn.Conv2d(channels_in, channels_out, 3, 1, padding=1), nn.BatchNorm2d(channels_out), nn.ReLU())

    def forward(self, x):
        return run_layers(self.layers, x, self.training)


class HiddenDecoder(nn.Module):
    """hidden/model/decoder.py:6-40 as configured for audio tiles: [B,1,128,128] -> [B,1,32,32]; ``message_length`` is
    the width of the penultimate ConvBNRelu."""

    def __init__(self, decoder_channels=64, decoder_blocks=7, message_length=30):
        super().__init__()
        seq = [ConvBNRelu(1, decoder_channels)]
        seq += [ConvBNRelu(decoder_channels, decoder_channels) for _ in range(decoder_blocks - 1)]
        seq += [ConvBNRelu(decoder_channels, message_length), nn.MaxPool2d(2, 2), ConvBNRelu(message_length, 1), nn.MaxPool2d(2, 2)]
        self.layers = nn.Sequential(*seq)

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != 1:
            raise ValueError("HiddenDecoder expects [B,1,H,W] tiles")
        return run_layers(self.layers, x, self.training)


def decoded_ber(decoded, message):
    """hidden/test_model.py:60-64: mean |round(clip(decoded, 0, 1)) - message| (host numpy, as the reference does)."""
    rounded = decoded.detach().cpu().numpy().round().clip(0, 1)
    return float(np.mean(np.abs(rounded - message.detach().cpu().numpy())))
