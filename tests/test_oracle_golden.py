"""The oracle (oracle/uformer_wm.py) pinned against goldens produced by the reference itself."""
import numpy as np
import torch

from oracle import uformer_wm as O
from synth import load_synth, synth_array, synth_bits, synth_tile, chirp_wave

T = torch.from_numpy


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=rtol, atol=atol)


def test_g2_stft_roundtrip(golden):
    g = golden("g2_stft")
    y = synth_tile(2, "g2_y", 1.0)
    wave = O.istft255(y)
    assert wave.shape == (2, 8002)
    close(wave, g["wave"])
    close(O.stft255(T(g["wave"])), g["spec"], atol=1e-5)


def test_g1_lewin_blocks(golden):
    for tag, dim, hw, heads, shift, mod in (("a", 32, 16, 1, 0, False), ("b", 64, 16, 2, 4, True)):
        g = golden("g1_block_" + tag)
        blk = load_synth(O.LeWinBlock(dim, hw, heads, 8, shift, 0.0, mod)).eval()
        x = synth_array("g1_x" + tag, (2, hw * hw, dim)).requires_grad_()
        out = blk(x)
        out.backward(synth_array("g1_dy" + tag, (2, hw * hw, dim)))
        close(out, g["out"], atol=1e-5)
        close(x.grad, g["dx"], rtol=1e-4, atol=1e-5)
        for k, p in blk.named_parameters():
            close(p.grad, g["g_" + k.replace(".", "__")], rtol=1e-4, atol=1e-4)


def test_g3_model_eval(golden):
    g = golden("g3_model_eval")
    net = load_synth(O.UformerAudio()).eval()
    x, msg = synth_tile(2, "g3_x"), synth_bits("g3_msg", (2, 1, 32, 32))
    with torch.no_grad():
        out = net(x, msg)
        tot, terms = O.train_losses(out, x, msg)
        for o, k in zip(out, ("stft_new", "noise", "wm_pred", "wm")):
            close(o, g[k], atol=1e-6)
        close(tot, g["loss"])
        close(torch.stack(terms), g["terms"])
        assert np.array_equal(O.payload_bits(out[3]).numpy().astype(np.uint8), g["bits"])
        close(net.wm_decode(x), g["wm_decode_x"], atol=1e-6)
        y, wm = net.feature_extract(x, msg)
        close(y[:, :, :8, :8], g["fe_y"], atol=1e-6)
        close(wm, g["fe_wm"], atol=1e-6)


def _keeps_from_masks(net, masks):
    """Re-nest the flat mask list the golden stores (call order; rate-0 blocks draw nothing)."""
    it = iter(masks)

    def stage(st):
        out = []
        for blk in st.blocks:
            if blk.drop_path_rate > 0:
                out.append((next(it), next(it)))
            else:
                out.append(None)
        return out

    def trunk(t):
        return [stage(getattr(t, f"encoderlayer_{i}")) for i in range(4)] + [stage(t.conv)]

    keeps = {"enc": trunk(net.encoder)}
    keeps["dec"] = [stage(getattr(net.decoder, f"decoderlayer_{i}")) for i in range(4)]
    keeps["ext"] = trunk(net.decoder_wm)
    assert next(it, None) is None
    return keeps


def test_g4_model_train_step(golden):
    g = golden("g4_model_train")
    net = load_synth(O.UformerAudio()).train()
    x, msg = synth_tile(2, "g4_x"), synth_bits("g4_msg", (2, 1, 32, 32))
    keeps = _keeps_from_masks(net, [T(m) for m in g["masks"]])
    opt = torch.optim.AdamW(net.parameters(), lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)
    out = net(x, msg, keeps=keeps)
    tot, terms = O.train_losses(out, x, msg)
    close(tot, g["loss"], rtol=1e-5)
    close(torch.stack(terms), g["terms"], rtol=1e-5)
    close(out[0][:, :, :16, :16], g["stft_new"], atol=1e-5)
    tot.backward()
    names = [str(n) for n in g["names"]]
    assert names == [n for n, p in net.named_parameters() if p.grad is not None]
    gn = np.array([net.get_parameter(n).grad.norm().item() for n in names])
    np.testing.assert_allclose(gn, g["gnorm"], rtol=2e-4, atol=1e-6)
    picks = [k[2:].replace("__", ".") for k in g.files if k.startswith("g_")]
    for k in picks:
        close(net.get_parameter(k).grad, g["g_" + k.replace(".", "__")], rtol=1e-3, atol=1e-5)
    opt.step()
    for k in picks:
        close(net.get_parameter(k), g["p_" + k.replace(".", "__")], rtol=1e-5, atol=1e-6)


def test_g5_modelA(golden):
    g = golden("g5_modelA")
    ma = load_synth(O.ModelA()).eval()
    x, msg = synth_tile(2, "g5_x"), synth_bits("g5_msg", (2, 1, 32, 32))
    with torch.no_grad():
        enc, ext = ma(x, msg)
    close(enc, g["enc"], atol=1e-6)
    close(ext, g["ext"], atol=1e-6)
    ma.train()
    ma.embedder_decoder[3].p = 0.0
    enc_t, ext_t = ma(x, msg)
    loss = torch.nn.functional.mse_loss(x, enc_t) + torch.nn.functional.mse_loss(ext_t, msg)
    loss.backward()
    close(loss, g["loss"])
    close(ma.embedder_encoder[0].weight.grad, g["g_first"], rtol=1e-4, atol=1e-6)
    close(ma.detector[8].weight.grad, g["g_last"], rtol=1e-4, atol=1e-6)


def test_g6_hidden_decoder_and_ber(golden):
    g = golden("g6_hidden")
    hd = load_synth(O.HiddenDecoder(64, 7, 30)).eval()
    x = synth_array("g6_x", (4, 1, 128, 128))
    with torch.no_grad():
        dec = hd(x)
        mid = hd.layers[:9](x)
    assert dec.shape == (4, 1, 32, 32)
    close(dec, g["dec"], atol=1e-6)
    close(mid[:, :, :8, :8], g["mid"], atol=1e-5)
    bits = synth_bits("g6_bits", (4, 1, 32, 32))
    assert abs(O.ber(dec, bits).item() - float(g["ber"])) < 1e-7


def test_eval_path_restatement_round_trip_and_tiling():
    """oracle/eval_path.py (audio_test.py:299-343, 594-600): tile count / len_last_clip rules and the exact round trip of
    the rectangular-window STFT pair that pins the stitch against the analysis."""
    from oracle import eval_path as E
    from synth import chirp_wave
    for L, tiles, last in ((1000, 1, 16), (8010, 2, 0), (8064, 2, 0), (8065, 2, 1), (12000, 2, 63), (30000, 4, 93)):
        wave = torch.as_tensor(chirp_wave(L, f"ep{L}"), dtype=torch.float32)
        w, clips, len_last = E.prepare_utterance(wave)
        assert clips.shape == (tiles, 2, 128, 128) and len_last == last == (1 + (L - 1) // 63) % 128
        assert torch.count_nonzero(clips[-1][:, :, len_last:]) == 0
        back = E.stitch_istft(clips, len_last, L)
        assert torch.allclose(back, wave, atol=2e-6)
    short = E.stitch_istft(clips[:1], 10, 2000)                    # length beyond the overlap-add: zero padded
    assert torch.count_nonzero(short[255 + 63 * 9 - 127:]) == 0
    assert abs(E.signaltonoise(np.array([1.0, 2.0, 3.0])) - 20 * np.log10(2 / np.sqrt(2 / 3))) < 1e-12


def test_noise_layer_oracle_matches_reference_golden(golden):
    """oracle/noise_layers.py against outputs of the reference's own hidden/noise_layers modules (G7); the random draws are
    replayed by seeding numpy as the generator script did and making the reference's numpy calls in its order."""
    import numpy as np
    from oracle import noise_layers as N
    from synth import synth_array
    g = golden("g7_noise_layers")

    def rand_rect(H, W, hr, wr):
        rh = int(np.rint((np.random.rand() * (hr[1] - hr[0]) + hr[0]) * H))
        rw = int(np.rint((np.random.rand() * (wr[0] - wr[0]) + wr[0]) * W))          # crop.py:32 draws from (w0, w0)
        h0 = 0 if rh == H else np.random.randint(0, H - rh)
        w0 = 0 if rw == W else np.random.randint(0, W - rw)
        return h0, h0 + rh, w0, w0 + rw
    x = lambda name, C, H, W: (synth_array(f"g7_{name}_n", (2, C, H, W)), synth_array(f"g7_{name}_c", (2, C, H, W)))
    n, c = x("crop", 2, 40, 56)
    np.random.seed(11)
    np.testing.assert_allclose(N.crop(n, *rand_rect(40, 56, (0.4, 0.7), (0.5, 0.8))).numpy(), g["crop_y"], rtol=1e-6)
    n, c = x("cropout", 2, 40, 56)
    np.random.seed(12)
    np.testing.assert_allclose(N.cropout(n, c, *rand_rect(40, 56, (0.3, 0.6), (0.4, 0.6))).numpy(), g["cropout_y"], rtol=1e-6)
    n, c = x("dropout", 2, 40, 56)
    np.random.seed(13)
    p = np.random.uniform(0.25, 0.45)
    mask = torch.tensor(np.random.choice([0.0, 1.0], (40, 56), p=[1 - p, p]), dtype=torch.float)
    np.testing.assert_allclose(N.dropout(n, c, mask).numpy(), g["dropout_y"], rtol=1e-6)
    for name, C, H, W, seed, rr in (("resize_down", 2, 40, 56, 14, (0.45, 0.75)), ("resize_up", 1, 24, 20, 15, (1.3, 1.7))):
        n, c = x(name, C, H, W)
        np.random.seed(seed)
        np.testing.assert_allclose(N.resize(n, np.random.rand() * (rr[1] - rr[0]) + rr[0]).numpy(), g[name + "_y"], rtol=1e-6)
    n, c = x("quant", 2, 40, 56)
    np.testing.assert_allclose(N.quantization(n).numpy(), g["quant_y"], rtol=1e-4, atol=1e-4)
    for name, H, W in (("jpeg", 40, 52), ("jpeg128", 128, 128)):
        n, c = x(name, 3, H, W)
        np.testing.assert_allclose(N.jpeg(n).numpy(), g[name + "_y"], rtol=1e-4, atol=2e-5)
