"""Plain-PyTorch fp32 CPU restatement of the reference watermark model.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Written from the behaviour of
``/root/reference/uformerWM/model.py``; every class cites the lines it follows.
``state_dict`` keys are identical to the reference's so one synthetic weight
set can be loaded into the reference, this oracle and the HIP product alike.

Differences from the reference are deliberate and behaviour-preserving:
  * modern complex ``torch.stft/istft`` API with the rectangular window made
    explicit (the reference uses the pre-1.8 real-view API, model.py:2458,2463);
  * stochastic depth takes an explicit per-sample keep mask so tests are
    deterministic (timm DropPath draws from the global RNG);
  * the unreachable variants (cross-modulator, conv projection, ffn mlp,
    dilation) are not restated — ``Uformer_audio`` never builds them.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

N_FFT = 255            # model.py:2458  torch.istft(..., n_fft=255)
HOP = N_FFT // 4       # torch default hop = n_fft // 4 = 63
DEPTHS = (1, 2, 8, 8, 2, 8, 8, 2, 1)          # utils/model_utils.py:84-85
HEADS = (1, 2, 4, 8, 16, 16, 8, 4, 2)         # model.py:2228


# --------------------------------------------------------------------------
# STFT round trip (model.py:2458, 2463)
# --------------------------------------------------------------------------
def istft255(y):
    """y: [B,2,128(f),128(t)] -> waveform [B,8002].  model.py:2458."""
    spec = torch.complex(y[:, 0], y[:, 1])
    win = torch.ones(N_FFT, dtype=y.dtype, device=y.device)
    return torch.istft(spec, n_fft=N_FFT, hop_length=HOP, win_length=N_FFT,
                       window=win, center=True, normalized=False, onesided=True)


def stft255(w):
    """w: [B,8002] -> [B,2,128,128].  model.py:2463."""
    win = torch.ones(N_FFT, dtype=w.dtype, device=w.device)
    s = torch.stft(w, n_fft=N_FFT, hop_length=HOP, win_length=N_FFT, window=win,
                   center=True, pad_mode="reflect", normalized=False,
                   onesided=True, return_complex=True)
    return torch.stack((s.real, s.imag), dim=1)


# --------------------------------------------------------------------------
# LeWin block pieces
# --------------------------------------------------------------------------
def rel_pos_index(ws):
    """Pairwise relative-position bin of the ws x ws window (model.py:496-505)."""
    ii = torch.arange(ws)
    gy, gx = torch.meshgrid(ii, ii, indexing="ij")
    p = torch.stack((gy.reshape(-1), gx.reshape(-1)))          # [2, ws*ws]
    d = p[:, :, None] - p[:, None, :] + (ws - 1)               # [2, n, n]
    return d[0] * (2 * ws - 1) + d[1]


def shift_attn_mask(hw, ws, shift, dtype=torch.float32):
    """-100/0 mask of the shifted-window attention (model.py:954-971)."""
    region = torch.zeros(hw, hw, dtype=dtype)
    edges = (slice(0, -ws), slice(-ws, -shift), slice(-shift, None))
    k = 0
    for a in edges:
        for b in edges:
            region[a, b] = k
            k += 1
    r = region.view(hw // ws, ws, hw // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    diff = r[:, None, :] - r[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


class QKV(nn.Module):
    """model.py:450-471 LinearProjection (keys to_q / to_kv)."""

    def __init__(self, dim):
        super().__init__()
        self.to_q = nn.Linear(dim, dim)
        self.to_kv = nn.Linear(dim, 2 * dim)


class WinAttn(nn.Module):
    """model.py:481-551 WindowAttention."""

    def __init__(self, dim, ws, heads):
        super().__init__()
        self.heads, self.ws = heads, ws
        self.scale = (dim // heads) ** -0.5
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))
        self.register_buffer("relative_position_index", rel_pos_index(ws))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        self.qkv = QKV(dim)
        self.proj = nn.Linear(dim, dim)

    def forward(self, xw, mask):
        nb, n, c = xw.shape
        h, d = self.heads, c // self.heads
        q = self.qkv.to_q(xw).view(nb, n, h, d).transpose(1, 2) * self.scale
        kv = self.qkv.to_kv(xw).view(nb, n, 2, h, d)
        k, v = kv[:, :, 0].transpose(1, 2), kv[:, :, 1].transpose(1, 2)
        s = q @ k.transpose(-1, -2)
        bias = self.relative_position_bias_table[self.relative_position_index.reshape(-1)]
        s = s + bias.view(n, n, h).permute(2, 0, 1)
        if mask is not None:
            nw = mask.shape[0]
            s = (s.view(nb // nw, nw, h, n, n) + mask[None, :, None]).view(nb, h, n, n)
        p = s.softmax(-1)
        return self.proj((p @ v).transpose(1, 2).reshape(nb, n, c))


class LeFF(nn.Module):
    """model.py:683-714: Linear-GELU, depthwise 3x3-GELU, Linear."""

    def __init__(self, dim, hidden):
        super().__init__()
        self.linear1 = nn.Sequential(nn.Linear(dim, hidden), nn.GELU())
        self.dwconv = nn.Sequential(nn.Conv2d(hidden, hidden, 3, 1, 1, groups=hidden), nn.GELU())
        self.linear2 = nn.Sequential(nn.Linear(hidden, dim))

    def forward(self, x):
        b, n, _ = x.shape
        s = int(math.sqrt(n))
        t = self.linear1(x)
        t = t.view(b, s, s, -1).permute(0, 3, 1, 2)
        t = self.dwconv(t)
        t = t.permute(0, 2, 3, 1).reshape(b, n, -1)
        return self.linear2(t)


class LeWinBlock(nn.Module):
    """model.py:879-1019 LeWinTransformerBlock (token_mlp='leff', linear projection)."""

    def __init__(self, dim, hw, heads, ws, shift, drop_path, modulator):
        super().__init__()
        if hw <= ws:                     # model.py:892-894
            shift, ws = 0, hw
        self.hw, self.ws, self.shift, self.drop_path_rate = hw, ws, shift, float(drop_path)
        # model.py:898 builds the table from the *constructor* win_size
        self.modulator = nn.Embedding(ws * ws, dim) if modulator else None
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WinAttn(dim, ws, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = LeFF(dim, dim * 4)
        self._mask = shift_attn_mask(hw, ws, shift) if shift > 0 else None

    def _drop(self, t, keep):
        """timm DropPath: scale kept samples by 1/keep_prob (model.py:916,1016-1017)."""
        if keep is None or self.drop_path_rate == 0.0 or not self.training:
            return t
        kp = 1.0 - self.drop_path_rate
        return t * (keep.view(-1, 1, 1).to(t.dtype) / kp)

    def forward(self, x, keep=None):
        b, n, c = x.shape
        hw, ws, sh = self.hw, self.ws, self.shift
        t = self.norm1(x).view(b, hw, hw, c)
        if sh:
            t = torch.roll(t, (-sh, -sh), (1, 2))
        t = t.view(b, hw // ws, ws, hw // ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, c)
        if self.modulator is not None:
            t = t + self.modulator.weight
        mask = self._mask.to(x.dtype) if self._mask is not None else None
        t = self.attn(t, mask)
        t = t.view(b, hw // ws, hw // ws, ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(b, hw, hw, c)
        if sh:
            t = torch.roll(t, (sh, sh), (1, 2))
        k1 = keep[0] if keep is not None else None
        k2 = keep[1] if keep is not None else None
        x = x + self._drop(t.reshape(b, n, c), k1)
        return x + self._drop(self.mlp(self.norm2(x)), k2)


class Stage(nn.Module):
    """model.py:1043-1090 BasicUformerLayer (key ``blocks``)."""

    def __init__(self, dim, hw, depth, heads, ws, dpr, modulator):
        super().__init__()
        self.blocks = nn.ModuleList(
            LeWinBlock(dim, hw, heads, ws, 0 if i % 2 == 0 else ws // 2, dpr[i], modulator)
            for i in range(depth))

    def forward(self, x, keeps=None):
        for i, blk in enumerate(self.blocks):
            x = blk(x, None if keeps is None else keeps[i])
        return x


class Down(nn.Module):
    """model.py:759-775."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(cin, cout, 4, 2, 1))

    def forward(self, x):
        b, n, c = x.shape
        s = int(math.sqrt(n))
        return self.conv(x.transpose(1, 2).reshape(b, c, s, s)).flatten(2).transpose(1, 2)


class Up(nn.Module):
    """model.py:785-800."""

    def __init__(self, cin, cout):
        super().__init__()
        self.deconv = nn.Sequential(nn.ConvTranspose2d(cin, cout, 2, 2))

    def forward(self, x):
        b, n, c = x.shape
        s = int(math.sqrt(n))
        return self.deconv(x.transpose(1, 2).reshape(b, c, s, s)).flatten(2).transpose(1, 2)


class InProj(nn.Module):
    """model.py:810-829 (conv3x3 + LeakyReLU(0.01), tokens out)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.proj = nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1), nn.LeakyReLU(inplace=False))

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class OutProj(nn.Module):
    """model.py:842-865."""

    def __init__(self, cin, cout):
        super().__init__()
        self.proj = nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1))

    def forward(self, x):
        b, n, c = x.shape
        s = int(math.sqrt(n))
        return self.proj(x.transpose(1, 2).reshape(b, c, s, s))


def _enc_dpr(depths, rate):
    n = sum(depths[:4])
    return [x.item() for x in torch.linspace(0, rate, n)]


class EncoderTrunk(nn.Module):
    """Shared body of model.py:1242-1394 (Encoder) and :1428-1583 (EncoderTransformerWM)."""

    def __init__(self, img, in_ch, embed, depths, heads, ws, dp_rate):
        super().__init__()
        dpr = _enc_dpr(depths, dp_rate)
        self.input_proj = InProj(in_ch, embed)          # Encoder owns one but never calls it
        o = 0
        for i in range(4):
            setattr(self, f"encoderlayer_{i}",
                    Stage(embed << i, img >> i, depths[i], heads[i], ws, dpr[o:o + depths[i]], False))
            setattr(self, f"dowsample_{i}", Down(embed << i, embed << (i + 1)))
            o += depths[i]
        self.conv = Stage(embed << 4, img >> 4, depths[4], heads[4], ws, [dp_rate] * depths[4], False)

    def trunk(self, y, keeps=None):
        feats = []
        for i in range(4):
            y = getattr(self, f"encoderlayer_{i}")(y, None if keeps is None else keeps[i])
            feats.append(y)
            y = getattr(self, f"dowsample_{i}")(y)
        feats.append(self.conv(y, None if keeps is None else keeps[4]))
        return feats


class Encoder(EncoderTrunk):
    def forward(self, y, keeps=None):
        return self.trunk(y, keeps)


class EncoderTransformerWM(EncoderTrunk):
    """model.py:1428-1583: trunk + conv2 (k=8, stride (16,8)) on [B,1,64,512] -> [B,4,8,8]."""

    def __init__(self, *a):
        super().__init__(*a)
        self.conv2 = nn.Conv2d(1, 1, 8, stride=(16, 8))

    def forward(self, x, keeps=None):
        c4 = self.trunk(self.input_proj(x), keeps)[4]
        z = self.conv2(c4.unsqueeze(1))
        return z.squeeze(1).reshape(z.shape[0], 4, 8, 8)


class Decoder(nn.Module):
    """model.py:1098-1240.  Note upsample_0 takes embed*32 channels (:1128)."""

    def __init__(self, img, embed, depths, heads, ws, dp_rate, modulator):
        super().__init__()
        dec = _enc_dpr(depths, dp_rate)[::-1]
        o = 0
        cin = [embed * 32, embed * 16, embed * 8, embed * 4]
        for i in range(4):
            dim = embed << (4 - i)
            setattr(self, f"upsample_{i}", Up(cin[i], dim // 2))
            d = depths[5 + i]
            setattr(self, f"decoderlayer_{i}",
                    Stage(dim, img >> (3 - i), d, heads[5 + i], ws, dec[o:o + d], modulator))
            o += d

    def forward(self, feats, keeps=None):
        y = feats[4]
        for i in range(4):
            y = torch.cat([getattr(self, f"upsample_{i}")(y), feats[3 - i]], -1)
            y = getattr(self, f"decoderlayer_{i}")(y, None if keeps is None else keeps[i])
        return y


class ConvAutoencoder(nn.Module):
    """model.py:1695-1748 payload autoencoder."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(1, 16, 3, padding=1)
        self.conv2 = nn.Conv2d(16, 4, 3, padding=1)
        self.t_conv1 = nn.ConvTranspose2d(4, 16, 2, stride=2)
        self.t_conv2 = nn.ConvTranspose2d(16, 1, 2, stride=2)

    def encode(self, m):
        m = F.max_pool2d(F.relu(self.conv1(m)), 2)
        return F.max_pool2d(F.relu(self.conv2(m)), 2)

    def decode(self, f):
        return torch.sigmoid(self.t_conv2(F.relu(self.t_conv1(f))))


class UformerAudio(nn.Module):
    """model.py:2225-2512.  forward -> (stft_new, noise, wm_pred, wm)."""

    def __init__(self, img_size=128, in_chans=2, dd_in=2, embed_dim=32, depths=DEPTHS,
                 num_heads=HEADS, win_size=8, drop_path_rate=0.1, modulator=True, **_):
        super().__init__()
        a = (img_size, dd_in, embed_dim, depths, num_heads, win_size, drop_path_rate)
        self.input_proj = InProj(dd_in, embed_dim)
        self.output_proj = OutProj(2 * embed_dim, in_chans)
        self.encoder = Encoder(*a)
        self.decoder = Decoder(img_size, embed_dim, depths, num_heads, win_size, drop_path_rate, modulator)
        self.encoder_wm = ConvAutoencoder()
        self.decoder_wm = EncoderTransformerWM(*a)
        self.stft_layer = nn.Sequential(nn.Conv2d(in_chans, 4, 3, 1, 1), nn.ReLU(),
                                        nn.Conv2d(4, in_chans, 3, 1, 1))
        self.apply(self._init)

    @staticmethod
    def _init(m):                      # model.py:2325-2332
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.LayerNorm):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)

    def wm_decode(self, y, keeps=None):               # model.py:2379-2382
        return self.encoder_wm.decode(self.decoder_wm(y, keeps))

    def embed(self, x, message, keeps=None):
        """model.py:2386-2421: returns (y, noise, wm_pred)."""
        f0 = self.encoder_wm.encode(message)                       # [B,4,8,8]
        f_exp = f0.flatten(2).repeat(1, 16, 8)                     # [B,64,512]
        feats = self.encoder(self.input_proj(x), None if keeps is None else keeps["enc"])
        pooled = F.max_pool2d(feats[4], (16, 8)).reshape(-1, 4, 8, 8)
        wm_pred = self.encoder_wm.decode(f0 + pooled)
        feats = feats[:4] + [torch.cat([f_exp, feats[4]], 2)]
        noise = self.output_proj(self.decoder(feats, None if keeps is None else keeps["dec"]))
        return x + noise, noise, wm_pred

    def feature_extract(self, x, message):           # model.py:2345-2377 (encoder_wm(message) fwd)
        f0 = self.encoder_wm.encode(message)
        wm_pred = self.encoder_wm.decode(f0)
        f_exp = f0.flatten(2).repeat(1, 16, 8)
        feats = self.encoder(self.input_proj(x))
        feats = feats[:4] + [torch.cat([f_exp, feats[4]], 2)]
        return x + self.output_proj(self.decoder(feats)), wm_pred

    def forward(self, x, message, mask=None, keeps=None, attack=None, extract_from="y"):
        y, noise, wm_pred = self.embed(x, message, keeps)
        wave = istft255(y)                                          # model.py:2458
        if attack is not None:                                      # hook point model.py:2461-2462
            wave = attack(wave)
        spec = stft255(wave)                                        # model.py:2463
        stft_new = self.stft_layer(spec)                            # model.py:2465
        src = y if extract_from == "y" else spec                    # model.py:2508 vs :2507
        wm = self.wm_decode(src, None if keeps is None else keeps["ext"])
        return stft_new, noise, wm_pred, wm


def train_losses(outputs, x, message):
    """4-term loss of audio_uformer_stft.py:463-482.  Returns (total, [l1,l2,l3,l4])."""
    stft_new, noise, wm_pred, wm = outputs
    l1 = F.mse_loss(stft_new, x)
    nn_ = torch.norm(noise) / noise.shape[0]
    l4 = F.mse_loss(nn_, torch.ones_like(nn_))
    l2 = F.mse_loss(wm_pred, message)
    l3 = F.mse_loss(wm, message)
    return l1 + l2 + l3 + l4, [l1, l2, l3, l4]


def payload_bits(wm):
    """hidden/test_model.py:60 rounding rule."""
    return wm.detach().round().clamp(0, 1)


def ber(wm, bits):
    """hidden/test_model.py:60-64: mean(|round(clip(decoded)) - message|)."""
    return (payload_bits(wm) - bits).abs().mean()


def draw_keeps(model, batch, gen):
    """Per-sample stochastic-depth keep masks for every block (train mode).

    Layout mirrors the module tree: {"enc": [stage][block](k1,k2), "dec": ..., "ext": ...}.
    """
    def for_stage(stage):
        out = []
        for blk in stage.blocks:
            kp = 1.0 - blk.drop_path_rate
            out.append(tuple((torch.rand(batch, generator=gen) < kp).float() for _ in range(2)))
        return out

    def for_trunk(t):
        return [for_stage(getattr(t, f"encoderlayer_{i}")) for i in range(4)] + [for_stage(t.conv)]

    return {"enc": for_trunk(model.encoder),
            "dec": [for_stage(getattr(model.decoder, f"decoderlayer_{i}")) for i in range(4)],
            "ext": for_trunk(model.decoder_wm)}


# --------------------------------------------------------------------------
# ModelA (model.py:3000-3066) and HiDDeN decoder (hidden/model/decoder.py:6-40)
# --------------------------------------------------------------------------
class ModelA(nn.Module):
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

    def encode(self, stft, wm):
        return self.embedder_decoder(torch.cat([self.embedder_encoder(stft), wm], 1))

    def decode(self, x):
        return self.detector(x)

    def forward(self, stft, wm):
        e = self.encode(stft, wm)
        return e, self.decode(e)


class ConvBNRelu(nn.Module):
    """hidden/model/conv_bn_relu.py:3-18."""

    def __init__(self, cin, cout):
        super().__init__()
        self.layers = nn.Sequential(nn.Conv2d(cin, cout, 3, 1, padding=1), nn.BatchNorm2d(cout), nn.ReLU())

    def forward(self, x):
        return self.layers(x)


class HiddenDecoder(nn.Module):
    """hidden/model/decoder.py:6-40: 1->C x blocks -> msg_len -> pool -> 1 -> pool."""

    def __init__(self, decoder_channels=64, decoder_blocks=7, message_length=30):
        super().__init__()
        seq = [ConvBNRelu(1, decoder_channels)]
        seq += [ConvBNRelu(decoder_channels, decoder_channels) for _ in range(decoder_blocks - 1)]
        seq += [ConvBNRelu(decoder_channels, message_length), nn.MaxPool2d(2, 2),
                ConvBNRelu(message_length, 1), nn.MaxPool2d(2, 2)]
        self.layers = nn.Sequential(*seq)

    def forward(self, x):
        return self.layers(x)
