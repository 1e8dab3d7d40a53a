"""Rows A18 / A19: the MI355X mirrors of ModelA and the HiDDeN decoder against the reference's own numbers (goldens G5, G6
written by tests/golden/make_golden.py from the imported reference) and against the oracle for state_dict compatibility."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from synth import load_synth, synth_array, synth_bits, synth_tile  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wm():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return importlib.import_module("image-in-speech-watermarking_amd")


def close(a, b, rtol=1e-4, atol=1e-5, what=""):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    err = np.abs(a - b).max()
    assert np.allclose(a, b, rtol=rtol, atol=atol), f"{what}: max abs err {err:.3e} (max ref {np.abs(b).max():.3e})"


def test_modelA_state_dict_matches_reference_layout(wm):
    from oracle import uformer_wm as O
    ours, ref = wm.ModelA().state_dict(), O.ModelA().state_dict()
    assert list(ours.keys()) == list(ref.keys())
    assert all(ours[k].shape == ref[k].shape for k in ref)
    ours_h, ref_h = wm.HiddenDecoder(64, 7, 30).state_dict(), O.HiddenDecoder(64, 7, 30).state_dict()
    assert list(ours_h.keys()) == list(ref_h.keys()) and all(ours_h[k].shape == ref_h[k].shape for k in ref_h)


def test_modelA_eval_train_and_gradients_match_golden(wm, golden):
    g = golden("g5_modelA")
    ma = load_synth(wm.ModelA()).cuda().eval()
    x, msg = synth_tile(2, "g5_x").cuda(), synth_bits("g5_msg", (2, 1, 32, 32)).cuda()
    with torch.no_grad():
        enc, ext = ma(x, msg)
    close(enc, g["enc"], what="ModelA eval encoded")
    close(ext, g["ext"], what="ModelA eval extracted")
    ma.train()
    ma.embedder_decoder[3].p = 0.0                       # as the golden: dropout off, BatchNorm on batch statistics
    rm0 = ma.embedder_encoder[1].running_mean.clone()
    enc_t, ext_t = ma(x, msg)
    loss = wm.mse_loss(x, enc_t) + wm.mse_loss(ext_t, msg)               # train_modelA.py:435-445
    loss.backward()
    close(enc_t[:, :, :16, :16], g["enc_t"], what="ModelA train encoded")
    close(ext_t, g["ext_t"], what="ModelA train extracted")
    close(loss, g["loss"], what="ModelA loss")
    close(ma.embedder_encoder[0].weight.grad, g["g_first"], rtol=1e-3, atol=1e-6, what="grad of the first conv")
    close(ma.detector[8].weight.grad, g["g_last"], rtol=1e-3, atol=1e-6, what="grad of the last conv")
    assert not torch.equal(ma.embedder_encoder[1].running_mean, rm0)     # running statistics moved
    assert int(ma.embedder_encoder[1].num_batches_tracked) == 1


def test_modelA_dropout_is_replayed_in_backward(wm):
    ma = load_synth(wm.ModelA()).cuda().train()
    x, msg = synth_tile(2, "g5_x").cuda(), synth_bits("g5_msg", (2, 1, 32, 32)).cuda()
    enc, ext = ma(x, msg)
    (wm.mse_loss(x, enc) + wm.mse_loss(ext, msg)).backward()
    gr = ma.embedder_encoder[0].weight.grad
    assert torch.isfinite(gr).all() and float(gr.abs().sum()) > 0
    d = torch.ones(4, 16, 64, 64, device="cuda")
    y = torch.empty_like(d)
    wm.ops.dropout(d, y, 0.5, 7, 3)
    kept = float((y > 0).float().mean())
    assert abs(kept - 0.5) < 0.01 and float(y.max()) == 2.0


def test_hidden_decoder_and_ber_match_golden(wm, golden):
    g = golden("g6_hidden")
    hd = load_synth(wm.HiddenDecoder(64, 7, 30)).cuda().eval()
    x = synth_array("g6_x", (4, 1, 128, 128)).cuda()
    with torch.no_grad():
        dec = hd(x)
        mid = wm.small_models.run_layers(list(hd.layers.children())[:9], x, False)
    assert dec.shape == (4, 1, 32, 32)
    close(dec, g["dec"], what="HiDDeN decoder output")
    close(mid[:, :, :8, :8], g["mid"], what="HiDDeN decoder mid feature")
    bits = synth_bits("g6_bits", (4, 1, 32, 32))
    assert abs(wm.decoded_ber(dec, bits) - float(g["ber"])) < 1e-6
    noiser = wm.Noiser(["Identity"], "cuda")
    out = noiser([dec, x])
    assert out[0] is dec and out[1] is x
    with pytest.raises(ValueError):
        wm.Noiser(["JpegCompression"], "cuda")


def test_batchnorm_kernel_matches_torch(wm):
    import torch.nn.functional as F
    x = synth_array("bnx", (3, 5, 8, 12)).requires_grad_()
    gam, bet = (1 + 0.1 * synth_array("bng", (5,))).requires_grad_(), (0.1 * synth_array("bnb", (5,))).requires_grad_()
    rm, rv = torch.zeros(5), torch.ones(5)
    y = F.leaky_relu(F.batch_norm(x, rm, rv, gam, bet, True, 0.1, 1e-5), 0.2)
    dy = synth_array("bndy", y.shape)
    y.backward(dy)
    xd, gd, bd = x.detach().cuda(), gam.detach().cuda(), bet.detach().cuda()
    rmd, rvd = torch.zeros(5, device="cuda"), torch.ones(5, device="cuda")
    yo, mean, inv = torch.empty_like(xd), torch.empty(5, device="cuda"), torch.empty(5, device="cuda")
    wm.ops.batchnorm2d_fwd(xd, gd, bd, rmd, rvd, yo, mean, inv, 1e-5, 0.1, True, "leaky", 0.2)
    close(yo, y, what="bn fwd")
    close(rmd, rm, what="running mean")
    close(rvd, rv, what="running var")
    dx, dg, db, sums = torch.empty_like(xd), torch.zeros(5, device="cuda"), torch.zeros(5, device="cuda"), torch.empty(10, device="cuda")
    wm.ops.batchnorm2d_bwd(dy.cuda(), xd, yo, gd, mean, inv, dx, dg, db, sums, True, "leaky", 0.2)
    close(dx, x.grad, what="bn dx")
    close(dg, gam.grad, what="bn dgamma")
    close(db, bet.grad, what="bn dbeta")
