"""Fused LeFF kernels (csrc/leff_fused.hip) against torch float64 on the SAME bf16-rounded operands.

Reference semantics: LeFF.forward (uformerWM/model.py:695-714) + the DropPath / residual add of
LeWinTransformerBlock.forward (model.py:1016-1017), and their autograd backward.  Shapes: the step's five (hw, C)
stages with C <= 128 plus small ones that exercise image borders in every tile (hw = 8: one tile per image).
Tolerances are bf16 ones: the kernels keep the two pre-activations, gelu(h1) and the hidden gradients as bf16 (as the
un-fused bf16 path does), so a result is held to a few bf16 ulps (2^-8) of its tensor's full scale.
"""
import importlib
import math

import pytest
import torch
import torch.nn.functional as F

from synth import synth_array

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wm():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return importlib.import_module("image-in-speech-watermarking_amd")


def bfr(t):
    return t.to(torch.bfloat16).to(t.dtype)


def dev(t, dtype=torch.float32):
    return t.to(dtype).contiguous().cuda()


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def dwconv(a1, wd, bd, B, hw, Ch):
    y = F.conv2d(a1.reshape(B, hw, hw, Ch).permute(0, 3, 1, 2), wd.reshape(Ch, 1, 3, 3), bd, padding=1, groups=Ch)
    return y.permute(0, 2, 3, 1).reshape(B * hw * hw, Ch)


SHAPES = [(2, 8, 32), (2, 16, 64), (3, 8, 128), (1, 128, 32), (1, 64, 64), (2, 32, 128), (1, 64, 128), (1, 128, 64)]


@pytest.mark.parametrize("B,hw,C", SHAPES)
def test_leff_fused_fwd_bwd(wm, B, hw, C):
    ops = wm.ops
    assert wm.lib.wm_leff_fusable(hw, C)
    bf = torch.bfloat16
    M, Ch = B * hw * hw, 4 * C
    D = torch.float64
    tag = f"{B}_{hw}_{C}"
    xn2 = bfr(synth_array("lx" + tag, (M, C))).to(D)
    x1 = synth_array("lr" + tag, (M, C)).to(D)
    w1 = bfr(synth_array("lw1" + tag, (Ch, C), 1.0 / math.sqrt(C))).to(D)
    b1 = (0.2 * synth_array("lb1" + tag, (Ch,))).to(D)
    wd = (0.3 * synth_array("lwd" + tag, (Ch, 9))).to(D)
    bd = (0.1 * synth_array("lbd" + tag, (Ch,))).to(D)
    w2 = bfr(synth_array("lw2" + tag, (C, Ch), 1.0 / math.sqrt(Ch))).to(D)
    b2 = (0.2 * synth_array("lb2" + tag, (C,))).to(D)
    s2 = (torch.rand(B, generator=torch.Generator().manual_seed(5)) + 0.5).to(D)
    srow = s2.repeat_interleave(hw * hw)[:, None]
    # ---- reference forward (the kernel's storage roundings restated)
    h1 = bfr(xn2 @ w1.t() + b1)
    a1 = bfr(F.gelu(h1))
    h2_exact = dwconv(a1, wd, bd, B, hw, Ch)
    h2 = bfr(h2_exact)
    a2 = bfr(F.gelu(h2_exact))
    x2 = (a2 @ w2.t() + b2) * srow + x1
    # ---- kernels
    x2d = torch.empty(M, C, device="cuda")
    h1d, h2d = torch.empty(M, Ch, device="cuda", dtype=bf), torch.empty(M, Ch, device="cuda", dtype=bf)
    args = (dev(xn2, bf), dev(x1), dev(w1, bf), dev(b1), dev(wd), dev(bd), dev(w2, bf), dev(b2))
    ops.leff_fwd(*args, x2d, B, hw, C, rowscale=dev(s2), h1=h1d, h2=h2d)
    assert rel(h1d, h1) <= 1e-2, f"h1 {rel(h1d, h1):.3e}"
    assert rel(h2d, h2) <= 1.5e-2, f"h2 {rel(h2d, h2):.3e}"
    assert rel(x2d, x2) <= 1e-2, f"x2 {rel(x2d, x2):.3e}"
    # inference form (nothing saved) gives the same x2
    x2e = torch.empty_like(x2d)
    ops.leff_fwd(*args, x2e, B, hw, C, rowscale=dev(s2))
    assert torch.equal(x2e, x2d)
    # ---- backward: from the SAVED bf16 pre-activations, as the kernel does
    h1s, h2s = h1d.double().cpu(), h2d.double().cpu()
    dy = bfr(synth_array("ldy" + tag, (M, C))).to(D)
    g2 = torch.autograd.functional.jvp(F.gelu, h2s, torch.ones_like(h2s))[1]
    dh2 = bfr(((dy * srow) @ w2) * g2)
    h1v = h1s.clone().requires_grad_()
    wdv, bdv = wd.clone().requires_grad_(), bd.clone().requires_grad_()
    dwconv(F.gelu(h1v), wdv, bdv, B, hw, Ch).backward(dh2)
    dh1 = h1v.grad
    dxn2 = bfr(dh1) @ w1
    dh1d, dxd = torch.empty(M, Ch, device="cuda", dtype=bf), torch.empty(M, C, device="cuda", dtype=bf)
    dwd, dbd = torch.zeros(Ch, 9, device="cuda"), torch.zeros(Ch, device="cuda")
    ops.leff_bwd(dev(dy, bf), h1d, h2d, args[2], args[4], args[6], dh1d, dxd, dwd, dbd, B, hw, C, rowscale=dev(s2))
    assert rel(dh1d, dh1) <= 1.5e-2, f"dh1 {rel(dh1d, dh1):.3e}"
    assert rel(dxd, dxn2) <= 1.5e-2, f"dxn2 {rel(dxd, dxn2):.3e}"
    assert rel(dwd, wdv.grad) <= 1e-2, f"dwd {rel(dwd, wdv.grad):.3e}"
    assert rel(dbd, bdv.grad) <= 1e-2, f"dbd {rel(dbd, bdv.grad):.3e}"
    # fp32 incoming gradient (first block behind a down / up-sampling layer), strip sums through scratch + deferred reduction
    scr = torch.empty(int(wm.lib.wm_leff_bwd_scratch_floats(B, hw, C)), device="cuda")
    jobs = ops.ReduceJobs()
    dh1e, dxe = torch.empty_like(dh1d), torch.empty_like(dxd)
    dwe, dbe = torch.zeros_like(dwd), torch.zeros_like(dbd)
    ops.leff_bwd(dev(dy), h1d, h2d, args[2], args[4], args[6], dh1e, dxe, dwe, dbe, B, hw, C, rowscale=dev(s2), scratch=scr,
                 defer=jobs.take(2))
    ops.reduce_jobs(jobs)
    assert torch.equal(dh1e, dh1d) and torch.equal(dxe, dxd)
    assert rel(dwe, wdv.grad) <= 1e-2 and rel(dbe, bdv.grad) <= 1e-2
    # no DropPath scale
    ops.leff_bwd(dev(dy, bf), h1d, h2d, args[2], args[4], args[6], dh1e, dxe, dwe, dbe, B, hw, C, scratch=scr)
    dh2n = bfr((dy @ w2) * g2)
    h1n = h1s.clone().requires_grad_()
    dwconv(F.gelu(h1n), wd, bd, B, hw, Ch).backward(dh2n)
    assert rel(dh1e, h1n.grad) <= 1.5e-2


def test_leff_fused_rejects_what_it_does_not_take(wm):
    ops = wm.ops
    f = wm.lib.wm_leff_fusable
    assert not f(16, 256) and not f(12, 64) and not f(4, 32) and f(8, 32) and f(64, 128)
    bf = torch.bfloat16
    x = torch.zeros(64, 256, device="cuda", dtype=bf)
    with pytest.raises(wm._lib.WmKernelError):
        ops.leff_fwd(x, x.float(), x, x.float(), x.float(), x.float(), x, x.float(), x.float(), 1, 8, 256)


def test_model_with_and_without_fused_leff(wm, monkeypatch):
    """Whole model, bf16, train mode with DropPath masks: the fused LeFF launches (WM_LEFF_FUSE=1: every block with C <= 128)
    against the separate fc1 / depthwise / fc2 launches — outputs and the whole gradient arena."""
    from synth import load_synth, synth_bits, synth_tile
    B = 3
    x, msg = synth_tile(B, "lf_x").cuda(), synth_bits("lf_m", (B, 1, 32, 32)).cuda()

    def run(fused):
        monkeypatch.setenv("WM_LEFF_FUSE", "1" if fused else "0")
        net = wm.UformerAudio(modulator=True, drop_path_rate=0.1, compute_mode="bf16")
        load_synth(net)
        net = net.cuda().train()
        net._ensure_runtime(torch.device("cuda"))
        rates = net._rates_host
        scales = torch.ones(len(rates), B)
        for s_, r in enumerate(rates):
            if r > 0:
                for b in range(B):
                    scales[s_, b] = 0.0 if (5 * s_ + 3 * b) % 4 == 0 else 1.0 / (1.0 - r)
        net._flat_grad.zero_()
        calls = []
        orig = wm.ops.leff_fwd
        monkeypatch.setattr(wm.ops, "leff_fwd", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
        out = net(x, msg, keep_scales=scales.cuda())
        (F.mse_loss(out[0], x) + F.mse_loss(out[2], msg) + F.mse_loss(out[3], msg) + out[1].square().mean()).backward()
        torch.cuda.synchronize()
        monkeypatch.undo()
        return [o.detach().clone() for o in out], net._flat_grad.double().cpu(), len(calls)
    of, gf, nf = run(True)
    ou, gu, nu = run(False)
    assert nf == 25 and nu == 0                                   # 2 trunks x (1 + 2 + 8) blocks + 2 + 1 decoder blocks with C <= 128
    for a, b, name in zip(of, ou, ("stft_new", "noise", "wm_pred", "wm")):
        assert rel(a, b) <= 2e-2, (name, rel(a, b))
    err = ((gf - gu).norm() / gu.norm()).item()
    cos = (gf @ gu / (gf.norm() * gu.norm())).item()
    # two bf16 paths with different roundings (gelu(h1) as bf16 in LDS, dh2 rounded before the transposed conv): measured 7.8e-2 /
    # cosine 0.9970 over the whole arena, where each path alone is 6e-2 / 0.998 from fp32 mode in the extractor segment
    # (test_bf16_droppath_backward_against_fp32_mode); a mis-routed DropPath factor or weight slice shows as >= 0.3
    assert err <= 0.12 and cos >= 0.992, (err, cos)
