"""HiDDeN encoder, discriminator and the adversarial training wrapper on libwm_hip kernels.

The reference configures these networks (``hidden/options.py:19-50``: encoder_blocks / encoder_channels,
discriminator_blocks / discriminator_channels, the three loss weights) and saves / restores them
(``hidden/utils.py:78-108``: ``model.encoder_decoder``, ``model.discriminator`` and their two optimizers; the evaluation
loop calls ``model.encoder_decoder.decoder``, ``hidden/audio_test.py:554``), but ``hidden/model/`` holds only ``decoder.py``
and ``conv_bn_relu.py``.  The modules below are re-derived from the HiDDeN paper (Zhu et al., ECCV 2018, section 3 and
figure 2) with the reference's own ConvBNRelu block and decoder:

* ``HiddenEncoder``: ``encoder_blocks`` ConvBNRelu on the cover, the message replicated over H x W and concatenated with
  those features and the cover, one more ConvBNRelu, a 1 x 1 convolution back to the cover's channels.  The decoder of
  this fork answers with an image [B, 1, H/4, W/4], so besides the paper's bit vector [B, L] the encoder takes a payload
  image [B, L, H/k, W/k], enlarged to H x W (nearest) as its message planes.
* ``HiddenDiscriminator``: ``discriminator_blocks`` ConvBNRelu, global average pool, Linear(channels, 1).
* ``EncoderDecoder``: encoder -> ``Noiser([encoded, cover])[0]`` -> decoder, returning (encoded, noised, decoded).
* ``Hidden``: the two Adam optimizers and ``train_on_batch`` / ``validate_on_batch`` with
  BCE-with-logits adversarial terms (cover = 1, encoded = 0) + ``encoder_loss * MSE(encoded, cover)`` +
  ``decoder_loss * MSE(decoded, message)`` and the bitwise error of ``hidden/test_model.py:60-64``.

There is nothing in the reference to check these against: ``oracle/hidden_nets.py`` restates the same architecture in
plain torch and the GPU tests hold the kernels to it (parity unpinned against the reference).  As everywhere in this
package the ``nn`` modules are parameter containers; every layer runs a libwm_hip kernel and CPU tensors are refused.
"""
import os

import torch
import torch.nn as nn

from . import ops
from .noise_layers import Noiser, _ResizeNearest
from .small_models import ConvBNRelu, HiddenDecoder, _CatChannels, _Conv, _need_cuda, decoded_ber, mse_loss, run_layers


class _MessageCat(torch.autograd.Function):
    """cat([message planes, features, cover], 1) with the message [B, L] replicated over H x W (``wm_plane_fill``) straight
    into the concat buffer, then two pitched copies."""

    @staticmethod
    def forward(ctx, message, feat, cover):
        feat, cover, message = _need_cuda(feat), _need_cuda(cover), _need_cuda(message)
        B, Ce, H, W = feat.shape
        Ci, L, hw = cover.shape[1], message.shape[1], H * W
        Ct = L + Ce + Ci
        out = torch.empty(B, Ct, H, W, device=feat.device)
        ops.plane_fill(message, out, B, L, Ct, 0, hw)
        ops.copy_cols(feat, Ce * hw, 0, out, Ct * hw, L * hw, B, Ce * hw)
        ops.copy_cols(cover, Ci * hw, 0, out, Ct * hw, (L + Ce) * hw, B, Ci * hw)
        ctx.dims = (B, L, Ce, Ci, H, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, L, Ce, Ci, H, W = ctx.dims
        hw, Ct = H * W, L + Ce + Ci
        dout = dout.contiguous()
        dmsg = None
        if ctx.needs_input_grad[0]:
            dmsg = torch.empty(B, L, device=dout.device)
            ops.plane_sum(dout, dmsg, B, L, Ct, 0, hw)
        dfeat, dcover = torch.empty(B, Ce, H, W, device=dout.device), torch.empty(B, Ci, H, W, device=dout.device)
        ops.copy_cols(dout, Ct * hw, L * hw, dfeat, Ce * hw, 0, B, Ce * hw)
        ops.copy_cols(dout, Ct * hw, (L + Ce) * hw, dcover, Ci * hw, 0, B, Ci * hw)
        return dmsg, dfeat, dcover


class _GlobalAvgPool(torch.autograd.Function):
    """nn.AdaptiveAvgPool2d((1, 1)) followed by the squeeze of the two unit dims: [B, C, H, W] -> [B, C]."""

    @staticmethod
    def forward(ctx, x):
        x = _need_cuda(x)
        B, C, H, W = x.shape
        y = torch.empty(B, C, device=x.device)
        ops.plane_sum(x, y, B, C, C, 0, H * W, 1.0 / (H * W))
        ctx.dims = (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W = ctx.dims
        dx = torch.empty(B, C, H, W, device=dy.device)
        ops.plane_fill(dy.contiguous(), dx, B, C, C, 0, H * W, 1.0 / (H * W))
        return dx


class _Linear(torch.autograd.Function):
    """The discriminator's nn.Linear(channels, 1) on the pooled features [M, K] (``wm_linear_small_*``)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = _need_cuda(x)
        y = torch.empty(x.shape[0], w.shape[0], device=x.device)
        ops.linear_small_fwd(x, w, b, y)
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dw, db = torch.zeros_like(w), torch.zeros(w.shape[0], device=x.device)
        ops.linear_small_bwd(dy, x, w, dx, dw, db)
        return dx, dw, db


class _BCELogits(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        logits = _need_cuda(logits)
        loss = torch.zeros((), device=logits.device)
        ops.bce_logits_fwd_bwd(logits, target, loss)
        ctx.save_for_backward(logits)
        ctx.target = target
        return loss

    @staticmethod
    def backward(ctx, g):
        (logits,) = ctx.saved_tensors
        d = torch.empty_like(logits)
        ops.bce_logits_fwd_bwd(logits, ctx.target, None, d, float(g))
        return d, None


def bce_with_logits(logits, target):
    """nn.BCEWithLogitsLoss()(logits, full_like(logits, target)) on the device."""
    return _BCELogits.apply(logits, float(target))


class HiddenEncoder(nn.Module):
    """Cover [B, C, H, W] + message ([B, L] bits, or a payload image [B, L, H/k, W/k]) -> encoded cover [B, C, H, W]."""

    def __init__(self, encoder_channels=64, encoder_blocks=4, message_length=30, image_channels=1):
        super().__init__()
        seq = [ConvBNRelu(image_channels, encoder_channels)]
        seq += [ConvBNRelu(encoder_channels, encoder_channels) for _ in range(encoder_blocks - 1)]
        self.conv_layers = nn.Sequential(*seq)
        self.after_concat_layer = ConvBNRelu(encoder_channels + image_channels + message_length, encoder_channels)
        self.final_layer = nn.Conv2d(encoder_channels, image_channels, kernel_size=1)
        self.message_length = message_length

    def forward(self, image, message):
        if message.shape[1] != self.message_length:
            raise ValueError(f"message has {message.shape[1]} planes, the encoder was built for {self.message_length}")
        feat = run_layers(self.conv_layers, image, self.training)
        if message.dim() == 4:                                      # payload image: every pixel covers a block of the cover
            H, W = image.shape[2:]
            h, w = message.shape[2:]
            if H % h or W % w:
                raise ValueError("the payload image must divide the cover size")
            planes = _ResizeNearest.apply(_need_cuda(message), H, W, h / H, w / W)
            cat = _CatChannels.apply(_CatChannels.apply(planes, feat), image)
        else:
            cat = _MessageCat.apply(message, feat, image)
        y = run_layers(self.after_concat_layer.layers, cat, self.training)
        m = self.final_layer
        return _Conv.apply(y, m.weight, m.bias, (1, 1), (0, 0), "none", False)


class HiddenDiscriminator(nn.Module):
    """Image -> one logit per image (cover vs encoded)."""

    def __init__(self, discriminator_channels=64, discriminator_blocks=3, image_channels=1):
        super().__init__()
        seq = [ConvBNRelu(image_channels, discriminator_channels)]
        seq += [ConvBNRelu(discriminator_channels, discriminator_channels) for _ in range(discriminator_blocks - 1)]
        seq.append(nn.AdaptiveAvgPool2d(output_size=(1, 1)))
        self.before_linear = nn.Sequential(*seq)
        self.linear = nn.Linear(discriminator_channels, 1)

    def forward(self, image):
        x = run_layers(list(self.before_linear.children())[:-1], image, self.training)
        return _Linear.apply(_GlobalAvgPool.apply(x), self.linear.weight, self.linear.bias)


class EncoderDecoder(nn.Module):
    """encoder -> noiser -> decoder; returns (encoded_image, noised_image, decoded_message)."""

    def __init__(self, config, noiser=None):
        super().__init__()
        # this fork's decoder answers with an image [B, 1, H/4, W/4] (hidden/model/decoder.py:23-26), so the payload is an
        # image of that size carried as ONE plane; message_planes = message_length gives the paper's bit-vector encoder
        planes = getattr(config, "message_planes", 1)
        self.encoder = HiddenEncoder(config.encoder_channels, config.encoder_blocks, planes, getattr(config, "image_channels", 1))
        self.noiser = noiser if noiser is not None else Noiser([])
        self.decoder = HiddenDecoder(config.decoder_channels, config.decoder_blocks, config.message_length)

    def forward(self, image, message):
        encoded = self.encoder(image, message)
        noised = self.noiser([encoded, image])[0]
        return encoded, noised, self.decoder(noised)


class Hidden:
    """The training wrapper the reference's checkpoint code expects (``hidden/utils.py:78-108``): ``encoder_decoder``,
    ``discriminator``, ``optimizer_enc_dec``, ``optimizer_discrim``; one discriminator and one encoder / decoder update per
    batch."""

    def __init__(self, config, device="cuda", noiser=None, lr=1e-3):
        self.config = config
        self.encoder_decoder = EncoderDecoder(config, noiser).to(device)
        self.discriminator = HiddenDiscriminator(config.discriminator_channels, config.discriminator_blocks,
                                                 getattr(config, "image_channels", 1)).to(device)
        self.optimizer_enc_dec = torch.optim.Adam(self.encoder_decoder.parameters(), lr=lr)
        self.optimizer_discrim = torch.optim.Adam(self.discriminator.parameters(), lr=lr)
        self.cover_label, self.encoded_label = 1.0, 0.0

    def _terms(self, images, messages, train):
        c = self.config
        d_cover = bce_with_logits(self.discriminator(images), self.cover_label)
        if train:
            d_cover.backward()
        encoded, noised, decoded = self.encoder_decoder(images, messages)
        d_encoded = bce_with_logits(self.discriminator(encoded.detach()), self.encoded_label)
        if train:
            d_encoded.backward()
            self.optimizer_discrim.step()
            self.optimizer_enc_dec.zero_grad()
        g_adv = bce_with_logits(self.discriminator(encoded), self.cover_label)
        g_enc = mse_loss(encoded, images)
        g_dec = mse_loss(decoded, messages.reshape(decoded.shape))
        g_loss = c.adversarial_loss * g_adv + c.encoder_loss * g_enc + c.decoder_loss * g_dec
        if train:
            g_loss.backward()
            self.optimizer_enc_dec.step()
        losses = {"loss": g_loss.item(), "encoder_mse": g_enc.item(), "dec_mse": g_dec.item(),
                  "bitwise-error": decoded_ber(decoded, messages.reshape(decoded.shape)), "adversarial_bce": g_adv.item(),
                  "discr_cover_bce": d_cover.item(), "discr_encod_bce": d_encoded.item()}
        return losses, (encoded, noised, decoded)

    def train_on_batch(self, batch):
        images, messages = batch
        self.encoder_decoder.train()
        self.discriminator.train()
        self.optimizer_discrim.zero_grad()
        return self._terms(images, messages, True)

    def validate_on_batch(self, batch):
        images, messages = batch
        self.encoder_decoder.eval()
        self.discriminator.eval()
        with torch.no_grad():
            return self._terms(images, messages, False)


def save_checkpoint(model, experiment_name, epoch, checkpoint_folder):
    """hidden/utils.py:78-92: '<experiment>--epoch-<n>.pyt' with the two state dicts, the two optimizer states and the epoch."""
    os.makedirs(checkpoint_folder, exist_ok=True)
    name = os.path.join(checkpoint_folder, f"{experiment_name}--epoch-{epoch}.pyt")
    torch.save({"enc-dec-model": model.encoder_decoder.state_dict(), "enc-dec-optim": model.optimizer_enc_dec.state_dict(),
                "discrim-model": model.discriminator.state_dict(), "discrim-optim": model.optimizer_discrim.state_dict(),
                "epoch": epoch}, name)
    return name


def model_from_checkpoint(hidden_net, checkpoint):
    """hidden/utils.py:104-109 (``checkpoint`` is the dict, e.g. ``torch.load(path, weights_only=True)``)."""
    hidden_net.encoder_decoder.load_state_dict(checkpoint["enc-dec-model"])
    hidden_net.optimizer_enc_dec.load_state_dict(checkpoint["enc-dec-optim"])
    hidden_net.discriminator.load_state_dict(checkpoint["discrim-model"])
    hidden_net.optimizer_discrim.load_state_dict(checkpoint["discrim-optim"])
