"""Host mirror of the reference's hidden/noise_layers/  on device kernels (csrc/noise.hip).

Same class names, constructor arguments and the ``[noised, cover]`` list-in / list-out convention of
hidden/noise_layers/{crop,cropout,dropout,resize,quantization,jpeg_compression,identity,noiser}.py, plus the
``--noise`` mini-DSL of hidden/noise_argparser.py:12-107 (``parse_noise`` / ``NoiseArgParser``).

Random draws: the reference draws every rectangle, ratio and mask on the HOST with numpy's global generator
(``np.random.rand / randint / uniform / choice``).  The mirror makes the same calls in the same order, so a caller
who seeds numpy gets the reference's draws bit for bit (``Dropout`` then uploads its [H, W] mask: 64 KB per call);
``Dropout(..., rng='device')`` draws the mask on the GPU with Philox instead (its own stream: parity unpinned, like the
AWGN stream).  All layers are differentiable with respect to the noised (and cover) image; the kernels' adjoint /
gradient forms are called from ``torch.autograd.Function``s.  GPU tensors only.
"""
import argparse
import ctypes as C
import re

import numpy as np
import torch
import torch.nn as nn

from ._lib import check, lib, ptr


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _img(t):
    if not t.is_cuda:
        raise RuntimeError("noise layers (MI355X build) run on GPU tensors only")
    if t.dim() != 4:
        raise ValueError("noise layers take [B, C, H, W] images")
    return t.contiguous().float()


# ---------------------------------------------------------------------------------------------- autograd functions
class _Crop(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, h0, h1, w0, w1):
        B, Cc, H, W = x.shape
        y = torch.empty(B, Cc, h1 - h0, w1 - w0, device=x.device)
        check(lib.wm_crop2d(ptr(x), ptr(y), B * Cc, H, W, h0, w0, h1 - h0, w1 - w0, 0, _stream()), "wm_crop2d")
        ctx.geom = (B, Cc, H, W, h0, h1, w0, w1)
        return y

    @staticmethod
    def backward(ctx, g):
        B, Cc, H, W, h0, h1, w0, w1 = ctx.geom
        g = g.contiguous().float()
        dx = torch.empty(B, Cc, H, W, device=g.device)
        check(lib.wm_crop2d(ptr(g), ptr(dx), B * Cc, H, W, h0, w0, h1 - h0, w1 - w0, 1, _stream()), "wm_crop2d(bwd)")
        return dx, None, None, None, None


class _Blend(torch.autograd.Function):
    """out = sel ? noised : cover, sel = rectangle or mask."""

    @staticmethod
    def forward(ctx, a, b, mask, rect):
        B, Cc, H, W = a.shape
        out = torch.empty_like(a)
        h0, h1, w0, w1 = rect if rect is not None else (0, 0, 0, 0)
        check(lib.wm_blend2d(ptr(a), ptr(b), ptr(mask), ptr(out), None, B * Cc, H, W, h0, h1, w0, w1, 0, _stream()), "wm_blend2d")
        ctx.save_for_backward(mask) if mask is not None else None
        ctx.has_mask, ctx.rect, ctx.shape = mask is not None, (h0, h1, w0, w1), (B, Cc, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        B, Cc, H, W = ctx.shape
        mask = ctx.saved_tensors[0] if ctx.has_mask else None
        g = g.contiguous().float()
        da, db = torch.empty_like(g), torch.empty_like(g)
        h0, h1, w0, w1 = ctx.rect
        check(lib.wm_blend2d(ptr(g), None, ptr(mask), ptr(da), ptr(db), B * Cc, H, W, h0, h1, w0, w1, 1, _stream()),
              "wm_blend2d(bwd)")
        return da, db, None, None


class _ResizeNearest(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, OH, OW, sh, sw):
        B, Cc, H, W = x.shape
        y = torch.empty(B, Cc, OH, OW, device=x.device)
        check(lib.wm_resize_nearest(ptr(x), ptr(y), B * Cc, H, W, OH, OW, sh, sw, 0, _stream()), "wm_resize_nearest")
        ctx.geom = (B, Cc, H, W, OH, OW, sh, sw)
        return y

    @staticmethod
    def backward(ctx, g):
        B, Cc, H, W, OH, OW, sh, sw = ctx.geom
        g = g.contiguous().float()
        dx = torch.zeros(B, Cc, H, W, device=g.device)
        check(lib.wm_resize_nearest(ptr(g), ptr(dx), B * Cc, H, W, OH, OW, sh, sw, 1, _stream()), "wm_resize_nearest(bwd)")
        return dx, None, None, None, None


class _Quantize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        f, out = torch.empty_like(x), torch.empty_like(x)
        state = torch.empty(int(lib.wm_quantization_state_bytes()) // 8, device=x.device, dtype=torch.int64)
        check(lib.wm_quantization_fwd(ptr(x), ptr(f), ptr(out), x.numel(), ptr(state), _stream()), "wm_quantization_fwd")
        ctx.save_for_backward(x, f, state)
        return out

    @staticmethod
    def backward(ctx, g):
        x, f, state = ctx.saved_tensors
        g = g.contiguous().float()
        dx = torch.empty_like(x)
        check(lib.wm_quantization_bwd(ptr(g), ptr(x), ptr(f), ptr(dx), x.numel(), ptr(state), _stream()), "wm_quantization_bwd")
        return dx


class _Jpeg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, keep):
        B, Cc, H, W = x.shape
        y = torch.empty_like(x)
        check(lib.wm_jpeg_mask(ptr(x), ptr(y), B, H, W, keep[0], keep[1], keep[2], 0, _stream()), "wm_jpeg_mask")
        ctx.keep = keep
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().float()
        B, Cc, H, W = g.shape
        dx = torch.empty_like(g)
        k = ctx.keep
        check(lib.wm_jpeg_mask(ptr(g), ptr(dx), B, H, W, k[0], k[1], k[2], 1, _stream()), "wm_jpeg_mask(adjoint)")
        return dx, None


# ---------------------------------------------------------------------------------------------- reference surface
def random_float(min, max):
    """crop.py:5-12."""
    return np.random.rand() * (max - min) + min


def get_random_rectangle_inside(image, height_ratio_range, width_ratio_range):
    """crop.py:15-45, numpy draws in the same order (the width ratio is drawn from (w[0], w[0]) there too)."""
    image_height, image_width = image.shape[2], image.shape[3]
    remaining_height = int(np.rint(random_float(height_ratio_range[0], height_ratio_range[1]) * image_height))
    remaining_width = int(np.rint(random_float(width_ratio_range[0], width_ratio_range[0]) * image_width))
    height_start = 0 if remaining_height == image_height else np.random.randint(0, image_height - remaining_height)
    width_start = 0 if remaining_width == image_width else np.random.randint(0, image_width - remaining_width)
    return height_start, height_start + remaining_height, width_start, width_start + remaining_width


class Identity(nn.Module):
    """identity.py:4-12."""

    def forward(self, noised_and_cover):
        return noised_and_cover


class Crop(nn.Module):
    """crop.py:48-74."""

    def __init__(self, height_ratio_range, width_ratio_range):
        super().__init__()
        self.height_ratio_range, self.width_ratio_range = height_ratio_range, width_ratio_range

    def forward(self, noised_and_cover):
        noised = _img(noised_and_cover[0])
        h0, h1, w0, w1 = get_random_rectangle_inside(noised, self.height_ratio_range, self.width_ratio_range)
        noised_and_cover[0] = _Crop.apply(noised, int(h0), int(h1), int(w0), int(w1))
        return noised_and_cover


class Cropout(nn.Module):
    """cropout.py:6-27."""

    def __init__(self, height_ratio_range, width_ratio_range):
        super().__init__()
        self.height_ratio_range, self.width_ratio_range = height_ratio_range, width_ratio_range

    def forward(self, noised_and_cover):
        noised, cover = _img(noised_and_cover[0]), _img(noised_and_cover[1])
        assert noised.shape == cover.shape
        rect = get_random_rectangle_inside(image=noised, height_ratio_range=self.height_ratio_range,
                                           width_ratio_range=self.width_ratio_range)
        noised_and_cover[0] = _Blend.apply(noised, cover, None, tuple(int(v) for v in rect))
        return noised_and_cover


class Dropout(nn.Module):
    """dropout.py:5-29.  rng='numpy' (default): the reference's host draws; rng='device': Philox mask on the GPU."""

    def __init__(self, keep_ratio_range, rng="numpy", seed=0):
        super().__init__()
        self.keep_min, self.keep_max = keep_ratio_range[0], keep_ratio_range[1]
        self.rng, self.seed, self._calls = rng, int(seed), 0

    def forward(self, noised_and_cover):
        noised, cover = _img(noised_and_cover[0]), _img(noised_and_cover[1])
        mask_percent = np.random.uniform(self.keep_min, self.keep_max)
        H, W = noised.shape[2:]
        if self.rng == "numpy":
            mask = np.random.choice([0.0, 1.0], (H, W), p=[1 - mask_percent, mask_percent])
            mask_t = torch.tensor(mask, device=noised.device, dtype=torch.float)
        else:
            mask_t = torch.empty(H, W, device=noised.device)
            self._calls += 1
            check(lib.wm_bernoulli_mask(ptr(mask_t), H * W, float(mask_percent), self.seed, self._calls, _stream()),
                  "wm_bernoulli_mask")
        return [_Blend.apply(noised, cover, mask_t, None), cover]


class Resize(nn.Module):
    """resize.py:6-26 (nearest only, the reference's default)."""

    def __init__(self, resize_ratio_range, interpolation_method="nearest"):
        super().__init__()
        if interpolation_method != "nearest":
            raise ValueError("Resize: only interpolation_method='nearest' (the reference's default) is built")
        self.resize_ratio_min, self.resize_ratio_max = resize_ratio_range[0], resize_ratio_range[1]
        self.interpolation_method = interpolation_method

    def forward(self, noised_and_cover):
        resize_ratio = random_float(self.resize_ratio_min, self.resize_ratio_max)
        noised = _img(noised_and_cover[0])
        H, W = noised.shape[2:]
        OH, OW = int(np.floor(H * resize_ratio)), int(np.floor(W * resize_ratio))     # F.interpolate(scale_factor=...)
        s = float(np.float32(1.0 / resize_ratio))
        noised_and_cover[0] = _ResizeNearest.apply(noised, OH, OW, s, s)
        return noised_and_cover


class Quantization(nn.Module):
    """quantization.py:17-45."""

    def __init__(self, device=None):
        super().__init__()
        self.min_value, self.max_value, self.N = 0.0, 255.0, 10

    def forward(self, noised_and_cover):
        return [_Quantize.apply(_img(noised_and_cover[0])), noised_and_cover[1]]


class JpegCompression(nn.Module):
    """jpeg_compression.py:65-160 ([B, 3, H, W] images: the colour transform indexes three planes)."""

    def __init__(self, device=None, yuv_keep_weights=(25, 9, 9)):
        super().__init__()
        self.device, self.yuv_keep_weighs = device, tuple(int(k) for k in yuv_keep_weights)

    def forward(self, noised_and_cover):
        noised = _img(noised_and_cover[0])
        if noised.shape[1] != 3:
            raise IndexError("JpegCompression works on 3-plane (RGB) images (jpeg_compression.py:50-62)")
        noised_and_cover[0] = _Jpeg.apply(noised, self.yuv_keep_weighs)
        return noised_and_cover


class Noiser(nn.Module):
    """noiser.py:8-31: Identity plus the configured layers; ONE is drawn per call with numpy's global generator."""

    def __init__(self, noise_layers=(), device=None):
        super().__init__()
        self.noise_layers = [Identity()]
        for layer in noise_layers:
            if type(layer) is str:
                if layer == "JpegPlaceholder":
                    self.noise_layers.append(JpegCompression(device))
                elif layer == "QuantizationPlaceholder":
                    self.noise_layers.append(Quantization(device))
                elif layer in ("Identity", "identity"):
                    self.noise_layers.append(Identity())
                else:
                    raise ValueError('Wrong layer placeholder string in Noiser.__init__(). Expected "JpegPlaceholder" or '
                                     f'"QuantizationPlaceholder" but got {layer} instead')
            else:
                self.noise_layers.append(layer)

    def forward(self, encoded_and_cover):
        random_noise_layer = np.random.choice(self.noise_layers, 1)[0]
        return random_noise_layer(encoded_and_cover)


# ---------------------------------------------------------------------------------------------- --noise DSL
_PAIR = r"\((\d+\.*\d*,\d+\.*\d*)\),\((\d+\.*\d*,\d+\.*\d*)\)"


def _pair(groups):
    h, w = groups[0].split(","), groups[1].split(",")
    return (float(h[0]), float(h[1])), (float(w[0]), float(w[1]))


def parse_noise(value):
    """noise_argparser.py:79-107: 'crop((0.2,0.3),(0.4,0.5))+cropout((..),(..))+dropout(0.25,0.35)+resize(0.4,0.6)+jpeg()+quant()'
    -> the list Noiser takes."""
    layers = []
    for command in value.split("+"):
        command = command.replace(" ", "")
        if command[:len("cropout")] == "cropout":
            layers.append(Cropout(*_pair(re.match(r"cropout\(" + _PAIR + r"\)", command).groups())))
        elif command[:len("crop")] == "crop":
            layers.append(Crop(*_pair(re.match(r"crop\(" + _PAIR + r"\)", command).groups())))
        elif command[:len("dropout")] == "dropout":
            r = re.match(r"dropout\((\d+\.*\d*,\d+\.*\d*)\)", command).groups()[0].split(",")
            layers.append(Dropout((float(r[0]), float(r[1]))))
        elif command[:len("resize")] == "resize":
            r = re.match(r"resize\((\d+\.*\d*,\d+\.*\d*)\)", command).groups()[0].split(",")
            layers.append(Resize((float(r[0]), float(r[1]))))
        elif command[:len("jpeg")] == "jpeg":
            layers.append("JpegPlaceholder")
        elif command[:len("quant")] == "quant":
            layers.append("QuantizationPlaceholder")
        elif command[:len("identity")] == "identity":
            pass                                                   # Noiser adds one Identity anyway
        else:
            raise ValueError("Command not recognized: \n{}".format(command))
    return layers


class NoiseArgParser(argparse.Action):
    """argparse action of noise_argparser.py:48-107 (``--noise 'crop(...)+jpeg()'``)."""

    def __call__(self, parser, namespace, values, option_string=None):
        setattr(namespace, self.dest, parse_noise(values[0] if isinstance(values, (list, tuple)) else values))
