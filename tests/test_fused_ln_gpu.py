"""LayerNorm fused into the epilogue of the GEMM that produces its input (bf16 mode): norm2 behind the attention projection and
the next block's norm1 (+ window modulator) behind fc2 (reference model.py:1016-1017 -> :982, :996-997).  Against float64 on the
same bf16-rounded operands: the fp32 output 2e-4, the bf16 normalised rows 4e-3 of full scale, statistics 1e-5 / 2e-4; and the
whole model with the fusion against the unfused launch sequence.
"""
import importlib
import math
import os
import zlib

import pytest
import torch
import torch.nn.functional as F

from synth import load_synth, synth_bits, synth_tile

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def wm():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return importlib.import_module("image-in-speech-watermarking_amd")


def rnd(tag, shape, scale=1.0, dtype=BF):
    g = torch.Generator(device="cuda").manual_seed(zlib.crc32(tag.encode()))
    return (torch.randn(shape, device="cuda", generator=g) * scale).to(dtype)


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def window_pos(rows, hw, shift, device):
    t = torch.arange(rows, device=device) % (hw * hw)
    yy, xx = t // hw, t % hw
    i, j = (yy - shift + hw) % hw, (xx - shift + hw) % hw
    return (i & 7) * 8 + (j & 7)


# (B, hw, C, K): M = B hw^2 rows; K = C (proj) or 4 C (fc2)
CASES = [(2, 64, 32, 32), (2, 64, 32, 128), (1, 64, 64, 64), (3, 32, 64, 256), (2, 32, 128, 128), (1, 32, 128, 512),
         (64, 32, 256, 256), (256, 16, 256, 1024)]


@pytest.mark.parametrize("B,hw,C,K", CASES)
@pytest.mark.parametrize("mod,shift", [(False, 0), (True, 4)])
def test_gemm_with_fused_layernorm(wm, B, hw, C, K, mod, shift):
    ops = wm.ops
    M = B * hw * hw
    assert ops.ln_fusable(M, C, K, 1)
    tag = f"{B}{hw}{C}{K}{mod}"
    a, w = rnd("la" + tag, (M, K)), rnd("lw" + tag, (C, K), 1.0 / math.sqrt(K))
    bias, res = rnd("lb" + tag, (C,), 1.0, torch.float32), rnd("lr" + tag, (M, C), 1.0, torch.float32)
    rs = torch.rand(B, device="cuda") + 0.5
    gamma, beta = 1 + 0.1 * rnd("lg" + tag, (C,), 1.0, torch.float32), 0.1 * rnd("lt" + tag, (C,), 1.0, torch.float32)
    modw = rnd("lm" + tag, (64, C), 0.5, torch.float32) if mod else None
    out, y = torch.empty(M, C, device="cuda"), torch.empty(M, C, device="cuda", dtype=BF)
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    ops.linear_fwd(a, w, bias, out, 1, res=res, rowscale=rs, rows_per_scale=hw * hw, ln=(gamma, beta, modw, y, mean, rstd, hw, shift))
    ref = (a.double() @ w.double().t() + bias.double()) * rs.double().repeat_interleave(hw * hw)[:, None] + res.double()
    assert rel(out, ref) <= 2e-4
    mu, var = ref.mean(1), ref.var(1, unbiased=False)
    yref = (ref - mu[:, None]) / torch.sqrt(var[:, None] + 1e-5) * gamma.double() + beta.double()
    if mod:
        yref = yref + modw.double()[window_pos(M, hw, shift, "cuda")]
    assert rel(y, yref) <= 4e-3
    assert rel(mean, mu) <= 1e-5 + 2e-4 and rel(rstd, 1 / torch.sqrt(var + 1e-5)) <= 2e-4
    # the unfused pair gives the same numbers up to summation order
    out2, y2 = torch.empty_like(out), torch.empty_like(y)
    m2, r2 = torch.empty_like(mean), torch.empty_like(rstd)
    ops.linear_fwd(a, w, bias, out2, 1, res=res, rowscale=rs, rows_per_scale=hw * hw)
    ops.layernorm_fwd(out2, gamma, beta, modw, y2, m2, r2, hw, shift)
    assert rel(out, out2) <= 1e-5 and rel(y, y2.float()) <= 8e-3 and rel(rstd, r2) <= 1e-4


def test_model_with_and_without_fused_layernorm(wm, monkeypatch):
    """Whole model forward + backward, bf16: fused epilogues (the default) against the separate LayerNorm launches."""
    x, msg = synth_tile(4, "fl_x").cuda(), synth_bits("fl_m", (4, 1, 32, 32)).cuda()

    def run(fused):
        if not fused:
            monkeypatch.setattr(wm.ops, "ln_fusable", lambda *a, **k: False)
        net = wm.UformerAudio(modulator=True, drop_path_rate=0.0, compute_mode="bf16")
        load_synth(net)
        net = net.cuda().train()
        net._flat_grad.zero_()
        out = net(x, msg)
        (F.mse_loss(out[0], x) + F.mse_loss(out[2], msg) + F.mse_loss(out[3], msg)).backward()
        monkeypatch.undo()
        return [o.detach().clone() for o in out], net._flat_grad.clone()
    of, gf = run(True)
    ou, gu = run(False)
    for a, b, name in zip(of, ou, ("stft_new", "noise", "wm_pred", "wm")):
        assert rel(a, b) <= 2e-2, (name, rel(a, b))
    err = ((gf.double() - gu.double()).norm() / gu.double().norm()).item()
    assert err <= 5e-2, err


def test_declined_fused_layernorm_falls_back_to_the_two_launches(wm):
    """wm_gemm_ln_fusable(M, N, K, mode) says yes for shapes whose launchers then decline the DropPath scale (8 x 8 images:
    64 rows per sample, not a multiple of the 256-row tile at N = 32).  linear_fwd must then produce the same pair of results
    through the plain product + the LayerNorm kernel instead of raising."""
    ops = wm.ops
    B, hw, C, K = 8, 8, 32, 128
    M = B * hw * hw
    assert ops.ln_fusable(M, C, K, 1)
    a, w = rnd("fa", (M, K)), rnd("fw", (C, K), 1.0 / math.sqrt(K))
    bias, res = rnd("fb", (C,), 1.0, torch.float32), rnd("fr", (M, C), 1.0, torch.float32)
    rs = torch.rand(B, device="cuda") + 0.5
    gamma, beta = 1 + 0.1 * rnd("fg", (C,), 1.0, torch.float32), 0.1 * rnd("ft", (C,), 1.0, torch.float32)
    out, y = torch.empty(M, C, device="cuda"), torch.empty(M, C, device="cuda", dtype=BF)
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    ops.linear_fwd(a, w, bias, out, 1, res=res, rowscale=rs, rows_per_scale=hw * hw, ln=(gamma, beta, None, y, mean, rstd, hw, 0))
    ref = (a.double() @ w.double().t() + bias.double()) * rs.double().repeat_interleave(hw * hw)[:, None] + res.double()
    mu, var = ref.mean(1), ref.var(1, unbiased=False)
    yref = (ref - mu[:, None]) / torch.sqrt(var[:, None] + 1e-5) * gamma.double() + beta.double()
    assert rel(out, ref) <= 2e-4 and rel(y, yref) <= 4e-3 and rel(rstd, 1 / torch.sqrt(var + 1e-5)) <= 2e-4
