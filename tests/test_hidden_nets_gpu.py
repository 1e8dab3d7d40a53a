"""Row N4, second half: the HiDDeN encoder / discriminator / adversarial batch on libwm_hip kernels against
oracle/hidden_nets.py (a torch restatement of the paper's architecture: the reference configures and checkpoints these
networks but does not contain them, so this parity is unpinned against the reference; the ConvBNRelu block and the decoder
inside them are pinned by golden G6 in test_small_models_gpu.py)."""
import importlib
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from synth import load_synth, synth_array, synth_bits  # noqa: E402

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


def grads_close(ours, ref, tol=5e-3):
    """Every parameter gradient against the oracle's in the 2-norm: ||ours - ref|| <= tol * ||ref|| + 1e-5 * (largest
    gradient norm of the network).  Element-wise bounds do not hold five train-mode BatchNorm + ReLU blocks deep: an
    activation within fp32 rounding of zero may switch sides and moves single entries by O(1) (torch's own fp32 gradients of
    this encoder sit 1e-3 .. 2.6e-3 from its fp64 gradients in this norm); a convolution bias in front of a train-mode
    BatchNorm has a zero gradient and both sides hold rounding noise there."""
    gmax = max(float(q.grad.norm()) for q in ref.parameters())
    for (n, p), (_, q) in zip(ours.named_parameters(), ref.named_parameters()):
        err, nrm = float((p.grad.detach().cpu() - q.grad).norm()), float(q.grad.norm())
        assert err <= tol * nrm + 1e-5 * gmax, f"grad {n}: ||diff|| {err:.3e} vs ||ref|| {nrm:.3e}"


CFG = SimpleNamespace(H=64, W=64, message_length=30, encoder_blocks=4, encoder_channels=32, decoder_blocks=3, decoder_channels=32,
                      use_discriminator=True, use_vgg=False, discriminator_blocks=3, discriminator_channels=32,
                      decoder_loss=1.0, encoder_loss=0.7, adversarial_loss=1e-3)     # hidden/options.py:19-50


def test_plane_and_bce_kernels(wm):
    ops = wm.ops
    B, L, Ct, c0, HW = 3, 5, 9, 2, 77
    v = synth_array("pl_v", (B, L)).cuda()
    dst = torch.zeros(B, Ct, HW, device="cuda")
    ops.plane_fill(v, dst, B, L, Ct, c0, HW, 0.5)
    ref = torch.zeros(B, Ct, HW)
    ref[:, c0:c0 + L] = 0.5 * v.cpu()[:, :, None]
    close(dst, ref, what="plane_fill")
    src = synth_array("pl_s", (B, Ct, HW)).cuda()
    out = torch.empty(B, L, device="cuda")
    ops.plane_sum(src, out, B, L, Ct, c0, HW, 2.0)
    close(out, 2.0 * src.cpu()[:, c0:c0 + L].sum(-1), rtol=1e-5, what="plane_sum")
    x = synth_array("bce_x", (7, 1), 3.0)
    for t in (0.0, 1.0):
        xr = x.clone().requires_grad_(True)
        lr = torch.nn.BCEWithLogitsLoss()(xr, torch.full_like(xr, t))
        lr.backward()
        xg = x.cuda().requires_grad_(True)
        lg = wm.hidden_nets.bce_with_logits(xg, t)
        lg.backward()
        close(lg, lr, rtol=1e-5, what=f"bce loss t={t}")
        close(xg.grad, xr.grad, rtol=1e-5, atol=1e-7, what=f"bce grad t={t}")


@pytest.mark.parametrize("form", ["bits", "payload"])
def test_encoder_matches_oracle_forward_and_gradients(wm, form):
    from oracle import hidden_nets as O
    L = 30 if form == "bits" else 1
    ours, ref = wm.hidden_nets.HiddenEncoder(32, 4, L, 1), O.HiddenEncoder(32, 4, L, 1)
    assert list(ours.state_dict().keys()) == list(ref.state_dict().keys())
    sd = load_synth(ref).state_dict()
    ours.load_state_dict(sd)
    ours.cuda().train()
    ref.train()
    img = synth_array("he_img", (2, 1, 64, 64), 0.5)
    msg = synth_bits("he_msg", (2, 30)) if form == "bits" else synth_bits("he_pay", (2, 1, 16, 16))
    mr = msg.clone().requires_grad_(True)
    yr = ref(img, mr)
    w = synth_array("he_w", tuple(yr.shape))
    (yr * w).sum().backward()
    mg, ig = msg.cuda().requires_grad_(True), img.cuda().requires_grad_(True)
    yg = ours(ig, mg)
    (yg * w.cuda()).sum().backward()
    close(yg, yr, rtol=2e-4, atol=2e-5, what="encoded image")
    close(mg.grad, mr.grad, rtol=2e-3, atol=1e-4, what="message gradient")
    grads_close(ours, ref)
    for (n, p), (_, q) in zip(ours.named_buffers(), ref.named_buffers()):
        close(p, q, rtol=1e-4, atol=1e-6, what=f"buffer {n}")


def test_discriminator_matches_oracle(wm):
    from oracle import hidden_nets as O
    ours, ref = wm.hidden_nets.HiddenDiscriminator(32, 3, 1), O.HiddenDiscriminator(32, 3, 1)
    assert list(ours.state_dict().keys()) == list(ref.state_dict().keys())
    ours.load_state_dict(load_synth(ref).state_dict())
    ours.cuda().train()
    ref.train()
    img = synth_array("hd_img", (3, 1, 64, 64), 0.5)
    lr = ref(img)
    (lr * torch.tensor([[1.0], [-2.0], [0.5]])).sum().backward()
    lg = ours(img.cuda())
    (lg * torch.tensor([[1.0], [-2.0], [0.5]], device="cuda")).sum().backward()
    assert lg.shape == (3, 1)
    close(lg, lr, rtol=2e-4, atol=2e-5, what="logits")
    grads_close(ours, ref)
    ours.eval()
    ref.eval()
    with torch.no_grad():
        close(ours(img.cuda()), ref(img), rtol=2e-4, atol=2e-5, what="eval logits")


def test_adversarial_batch_terms_and_one_update_match_oracle(wm):
    """The loss terms of hidden_nets.Hidden.train_on_batch on identical weights, then the weights after one Adam update of
    both optimizers against the same update made with torch on the oracle."""
    from oracle import hidden_nets as O
    np.random.seed(0)
    hid = wm.hidden_nets.Hidden(CFG, "cuda")
    ref_ed, ref_d = O.EncoderDecoder(CFG), O.HiddenDiscriminator(CFG.discriminator_channels, CFG.discriminator_blocks, 1)
    assert list(hid.encoder_decoder.state_dict().keys()) == list(ref_ed.state_dict().keys())
    hid.encoder_decoder.load_state_dict(load_synth(ref_ed).state_dict())
    hid.discriminator.load_state_dict(load_synth(ref_d).state_dict())
    img, msg = synth_array("hb_img", (2, 1, 64, 64), 0.5), synth_bits("hb_msg", (2, 1, 16, 16))
    ref_ed.train()
    ref_d.train()
    opt_ed, opt_d = torch.optim.Adam(ref_ed.parameters(), lr=1e-3), torch.optim.Adam(ref_d.parameters(), lr=1e-3)
    opt_d.zero_grad()
    bce = torch.nn.BCEWithLogitsLoss()
    lc = ref_d(img)
    d_cover = bce(lc, torch.ones_like(lc))
    d_cover.backward()
    enc, _, dec = ref_ed(img, msg)
    le = ref_d(enc.detach())
    d_enc = bce(le, torch.zeros_like(le))
    d_enc.backward()
    opt_d.step()
    opt_ed.zero_grad()
    lg = ref_d(enc)
    g_adv = bce(lg, torch.ones_like(lg))
    g_enc, g_dec = torch.nn.functional.mse_loss(enc, img), torch.nn.functional.mse_loss(dec, msg)
    g_loss = CFG.adversarial_loss * g_adv + CFG.encoder_loss * g_enc + CFG.decoder_loss * g_dec
    g_loss.backward()
    opt_ed.step()
    losses, (e, n, d) = hid.train_on_batch([img.cuda(), msg.cuda()])
    close(e, enc, rtol=2e-4, atol=2e-5, what="encoded")
    close(d, dec, rtol=5e-4, atol=5e-5, what="decoded")
    for k, v in (("discr_cover_bce", d_cover), ("discr_encod_bce", d_enc), ("adversarial_bce", g_adv), ("encoder_mse", g_enc),
                 ("dec_mse", g_dec), ("loss", g_loss)):
        assert abs(losses[k] - float(v.detach())) <= 2e-4 * abs(float(v.detach())) + 1e-6, (k, losses[k], float(v.detach()))
    assert 0.0 <= losses["bitwise-error"] <= 1.0
    # Adam's first step moves every weight by ~lr * sign(grad): compare where the gradient is well above rounding noise
    for (nme, p), (_, q) in zip(list(hid.discriminator.named_parameters()) + list(hid.encoder_decoder.named_parameters()),
                                list(ref_d.named_parameters()) + list(ref_ed.named_parameters())):
        if nme.endswith("layers.0.bias"):                           # zero gradient (BatchNorm follows): Adam normalises rounding noise
            continue
        g = q.grad
        sel = g.abs() > 1e-3 * g.abs().max()
        close(p.detach().cpu()[sel], q.detach()[sel], rtol=1e-4, atol=2e-5, what=f"updated {nme}")
    v, _ = hid.validate_on_batch([img.cuda(), msg.cuda()])
    assert np.isfinite(v["loss"])
