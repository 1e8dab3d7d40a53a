"""CPU restatement (plain torch) of the HiDDeN encoder, discriminator and adversarial batch step.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: /root/reference configures and checkpoints these networks
(hidden/options.py:19-50, hidden/utils.py:78-108, hidden/audio_test.py:554 uses `model.encoder_decoder.decoder`) but does
not contain them (hidden/model/ holds decoder.py and conv_bn_relu.py only), so there is no reference output or fixture to
hold this file to.  It restates the architecture of the HiDDeN paper (Zhu et al., ECCV 2018, section 3, figure 2) around
the reference's ConvBNRelu block and decoder (oracle/uformer_wm.py: ConvBNRelu, HiddenDecoder, which ARE pinned by golden
G6), and is what tests/test_hidden_nets_gpu.py compares the HIP path with.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .uformer_wm import ConvBNRelu, HiddenDecoder


class HiddenEncoder(nn.Module):
    def __init__(self, encoder_channels=64, encoder_blocks=4, message_length=30, image_channels=1):
        super().__init__()
        seq = [ConvBNRelu(image_channels, encoder_channels)]
        seq += [ConvBNRelu(encoder_channels, encoder_channels) for _ in range(encoder_blocks - 1)]
        self.conv_layers = nn.Sequential(*seq)
        self.after_concat_layer = ConvBNRelu(encoder_channels + image_channels + message_length, encoder_channels)
        self.final_layer = nn.Conv2d(encoder_channels, image_channels, kernel_size=1)

    def forward(self, image, message):
        H, W = image.shape[2:]
        if message.dim() == 2:                                       # bit vector: one constant plane per bit
            planes = message[:, :, None, None].expand(-1, -1, H, W)
        else:                                                        # payload image: each pixel covers a block
            planes = F.interpolate(message, size=(H, W), mode="nearest")
        feat = self.conv_layers(image)
        return self.final_layer(self.after_concat_layer(torch.cat([planes, feat, image], dim=1)))


class HiddenDiscriminator(nn.Module):
    def __init__(self, discriminator_channels=64, discriminator_blocks=3, image_channels=1):
        super().__init__()
        seq = [ConvBNRelu(image_channels, discriminator_channels)]
        seq += [ConvBNRelu(discriminator_channels, discriminator_channels) for _ in range(discriminator_blocks - 1)]
        seq.append(nn.AdaptiveAvgPool2d(output_size=(1, 1)))
        self.before_linear = nn.Sequential(*seq)
        self.linear = nn.Linear(discriminator_channels, 1)

    def forward(self, image):
        return self.linear(self.before_linear(image).flatten(1))


class EncoderDecoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.encoder = HiddenEncoder(config.encoder_channels, config.encoder_blocks, getattr(config, "message_planes", 1),
                                     getattr(config, "image_channels", 1))
        self.decoder = HiddenDecoder(config.decoder_channels, config.decoder_blocks, config.message_length)

    def forward(self, image, message):                              # Identity noiser
        encoded = self.encoder(image, message)
        return encoded, encoded, self.decoder(encoded)


def batch_terms(encoder_decoder, discriminator, config, images, messages):
    """The loss terms of one adversarial batch (no optimizer step): discriminator BCE on cover (1) and on the detached
    encoded image (0); generator loss = adversarial_loss * BCE(D(encoded), 1) + encoder_loss * MSE(encoded, cover) +
    decoder_loss * MSE(decoded, message)."""
    bce = nn.BCEWithLogitsLoss()
    d_cover_logits = discriminator(images)
    d_cover = bce(d_cover_logits, torch.ones_like(d_cover_logits))
    encoded, noised, decoded = encoder_decoder(images, messages)
    d_enc_logits = discriminator(encoded.detach())
    d_encoded = bce(d_enc_logits, torch.zeros_like(d_enc_logits))
    g_logits = discriminator(encoded)
    g_adv = bce(g_logits, torch.ones_like(g_logits))
    g_enc = F.mse_loss(encoded, images)
    g_dec = F.mse_loss(decoded, messages.reshape(decoded.shape))
    g_loss = config.adversarial_loss * g_adv + config.encoder_loss * g_enc + config.decoder_loss * g_dec
    return dict(d_cover=d_cover, d_encoded=d_encoded, g_adv=g_adv, g_enc=g_enc, g_dec=g_dec, g_loss=g_loss,
                encoded=encoded, decoded=decoded)
