"""Generate the committed golden fixtures from the REFERENCE itself (build container only).

Run:  python tests/golden/make_golden.py
Reads /root/reference (read-only) through ``_refload.py``; writes small ``.npz`` files next to this
script.  Inputs and weights come from ``tests/synth.py`` (name-keyed, deterministic), so fixtures
hold only the reference's *outputs* (and the few inputs that are not regenerable).
Goldens: G1 per-op LeWin block, G2 STFT conventions, G3 model eval, G4 model train
(grads + one AdamW step), G5 ModelA, G6 HiDDeN decoder + BER.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from _refload import load_reference_model_module  # noqa: E402
from synth import load_synth, synth_array, synth_bits, synth_tile, chirp_wave  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)
R = load_reference_model_module()
DP = R._DropPathStandIn


def save(name, **arrs):
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def build_ref_model():
    return R.UformerAudio(img_size=128, embed_dim=32, win_size=8, token_projection="linear",
                          token_mlp="leff", depths=[1, 2, 8, 8, 2, 8, 8, 2, 1], modulator=True,
                          dd_in=2, in_chans=2, audio_scale="0")


def ref_losses(out, x, msg):
    """audio_uformer_stft.py:463-482 evaluated with the same torch calls."""
    mse = torch.nn.MSELoss()
    audio, noise, wm_gen, wm_dec = out
    l1 = mse(audio, x)
    nn_ = torch.norm(noise) / noise.shape[0]
    l4 = mse(nn_, torch.ones_like(nn_))
    l2 = mse(wm_gen, msg)
    l3 = mse(wm_dec, msg)
    return l1 + l2 + l3 + l4, torch.stack([l1, l2, l3, l4])


# ---- G2: STFT conventions -------------------------------------------------------------------
y = synth_tile(2, "g2_y", 1.0)
wave = torch.istft(y.permute(0, 2, 3, 1), n_fft=255)            # model.py:2458 via the adapter
spec = torch.stft(wave, n_fft=255).permute(0, 3, 1, 2)          # model.py:2463
save("g2_stft", wave=wave, spec=spec)
w = chirp_wave(48000, "g2_chirp")[None]
st = torch.stft(w, n_fft=256, hop_length=128, win_length=256, return_complex=False)[:, :-1, :]
pad = 128 - st.shape[2] % 128                                     # audio_test.py:476-477
st = torch.nn.functional.pad(st, (0, 0, 0, pad))
tiles = torch.stack([st[:, :, 128 * j:128 * (j + 1), :].permute(0, 3, 1, 2).squeeze(0)
                     for j in range(st.shape[2] // 128)])         # audio_test.py:482-488,507-519
save("g2_dataprep", tiles=tiles)

# ---- G1: LeWin blocks -----------------------------------------------------------------------
for tag, dim, hw, heads, shift, mod in (("a", 32, 16, 1, 0, False), ("b", 64, 16, 2, 4, True)):
    blk = R.LeWinTransformerBlock(dim=dim, input_resolution=(hw, hw), num_heads=heads, win_size=8,
                                  shift_size=shift, token_projection="linear", token_mlp="leff",
                                  modulator=mod)
    load_synth(blk).eval()
    x = synth_array("g1_x" + tag, (2, hw * hw, dim)).requires_grad_()
    dy = synth_array("g1_dy" + tag, (2, hw * hw, dim))
    out = blk(x)
    out.backward(dy)
    g = {("g_" + k.replace(".", "__")): v.grad for k, v in blk.named_parameters()}
    save("g1_block_" + tag, out=out, dx=x.grad, **g)

# ---- G3: whole model, eval ------------------------------------------------------------------
net = load_synth(build_ref_model()).eval()
x, msg = synth_tile(2, "g3_x"), synth_bits("g3_msg", (2, 1, 32, 32))
with torch.no_grad():
    out = net(x, msg)
    tot, terms = ref_losses(out, x, msg)
    wm_dec = net.wm_decode(x)
    y_fe, wm_fe = net.feature_extract(x, msg)
save("g3_model_eval", stft_new=out[0], noise=out[1], wm_pred=out[2], wm=out[3], loss=tot,
     terms=terms, bits=out[3].round().clamp(0, 1).to(torch.uint8), wm_decode_x=wm_dec,
     fe_y=y_fe[:, :, :8, :8], fe_wm=wm_fe)

# ---- G4: whole model, train (stochastic depth replayed), grads + one AdamW step ---------------
net = load_synth(build_ref_model()).train()
x, msg = synth_tile(2, "g4_x"), synth_bits("g4_msg", (2, 1, 32, 32))
n_dp = 2 * sum(1 for m in net.modules() if isinstance(m, DP) and m.drop_prob > 0)  # each block calls it twice
g = torch.Generator().manual_seed(4)
masks = [(torch.rand(2, generator=g) < 0.85).float() for _ in range(n_dp)]
DP.queue = [m.clone() for m in masks]
opt = torch.optim.AdamW(net.parameters(), lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)
opt.zero_grad()
out = net(x, msg)
assert not DP.queue
DP.queue = None
tot, terms = ref_losses(out, x, msg)
tot.backward()
names = [n for n, p in net.named_parameters() if p.grad is not None]
gnorm = np.array([net.get_parameter(n).grad.norm().item() for n in names], dtype=np.float64)
gsum = np.array([net.get_parameter(n).grad.double().sum().item() for n in names], dtype=np.float64)
pick = ["input_proj.proj.0.weight", "encoder.encoderlayer_0.blocks.0.attn.relative_position_bias_table",
        "encoder.encoderlayer_2.blocks.3.attn.qkv.to_kv.weight", "encoder.dowsample_1.conv.0.bias",
        "decoder.upsample_0.deconv.0.bias", "decoder.decoderlayer_3.blocks.0.modulator.weight",
        "decoder.decoderlayer_1.blocks.5.mlp.dwconv.0.weight", "decoder_wm.conv2.weight",
        "encoder_wm.t_conv1.weight", "stft_layer.0.weight", "output_proj.proj.0.weight",
        "decoder_wm.encoderlayer_3.blocks.7.norm2.weight"]
grads = {("g_" + k.replace(".", "__")): net.get_parameter(k).grad.clone() for k in pick}
opt.step()
post = {("p_" + k.replace(".", "__")): net.get_parameter(k).detach().clone() for k in pick}
save("g4_model_train", masks=torch.stack(masks), loss=tot, terms=terms, stft_new=out[0][:, :, :16, :16],
     wm=out[3], names=np.array(names), gnorm=gnorm, gsum=gsum, **grads, **post)

# ---- G5: ModelA -------------------------------------------------------------------------------
ma = load_synth(R.ModelA()).eval()
x, msg = synth_tile(2, "g5_x"), synth_bits("g5_msg", (2, 1, 32, 32))
with torch.no_grad():
    enc, ext = ma(x, msg)
ma.train()
for m in ma.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0                       # dropout off so the train-mode (batch-stat BN) golden is deterministic
enc_t, ext_t = ma(x, msg)
mse = torch.nn.MSELoss()
loss = mse(x, enc_t) + mse(ext_t, msg)                      # train_modelA.py:435-445
loss.backward()
save("g5_modelA", enc=enc, ext=ext, enc_t=enc_t[:, :, :16, :16], ext_t=ext_t, loss=loss,
     g_first=ma.embedder_encoder[0].weight.grad, g_last=ma.detector[8].weight.grad)

# ---- G6: HiDDeN decoder + Noiser/Identity + BER ---------------------------------------------
sys.modules.pop("model", None)              # uformerWM/model.py shadows hidden's ``model`` package
sys.path[:] = [q for q in sys.path if not q.startswith("/root/reference")]
sys.path.insert(0, "/root/reference/hidden")
from options import HiDDenConfiguration  # noqa: E402  (reference hidden/options.py)
from model.decoder import Decoder as HDecoder  # noqa: E402
cfg = HiDDenConfiguration(H=128, W=128, message_length=30, encoder_blocks=4, encoder_channels=64,
                          decoder_blocks=7, decoder_channels=64, use_discriminator=False, use_vgg=False,
                          discriminator_blocks=3, discriminator_channels=64, decoder_loss=1,
                          encoder_loss=0.7, adversarial_loss=1e-3)
hd = load_synth(HDecoder(cfg)).eval()
x = synth_array("g6_x", (4, 1, 128, 128))
with torch.no_grad():
    dec = hd(x)
    mid = hd.layers[:9](x)             # through ConvBNRelu(64->30) and the first pool: [4,30,64,64]
bits = synth_bits("g6_bits", (4, 1, 32, 32))
rounded = dec.numpy().round().clip(0, 1)
err = np.mean(np.abs(rounded - bits.numpy()))                  # hidden/test_model.py:60-64
save("g6_hidden", dec=dec, mid=mid[:, :, :8, :8], ber=err)
print("done")
