"""Per-kernel parity: every C-ABI entry point of libwm_hip.so against plain torch fp32 (CPU) references.

Runs on the MI355X only (-m gpu).  fp32 "parity mode" kernels are held to 1e-4 relative (north_star),
bf16 MFMA mode to bf16 rounding of the operands.
"""
import importlib
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import uformer_wm as O
from oracle import attacks as OA
from synth import synth_array

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wm():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return importlib.import_module("image-in-speech-watermarking_amd")


def dev(t):
    return t.contiguous().cuda()


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def assert_close(a, b, tol, what=""):
    e = rel_err(a, b)
    assert e <= tol, f"{what}: max rel err {e:.3e} > {tol:.1e}"


TOL = {0: 1e-4, 1: 3e-2}


# ---------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("M,N,K", [(300, 96, 64), (257, 32, 32), (130, 64, 128), (512, 200, 36), (64, 2048, 512)])
def test_gemm_forward_epilogues(wm, mode, M, N, K):
    ops = wm.ops
    x, w = synth_array("gx", (M, K)), synth_array("gw", (N, K), 1.0 / math.sqrt(K))
    bias, res = synth_array("gb", (N,)), synth_array("gr", (M, N))
    rows_per = 10
    rs = torch.rand((M + rows_per - 1) // rows_per) + 0.5
    out = torch.empty(M, N, device="cuda")
    ops.linear_fwd(dev(x), dev(w), dev(bias), out, mode, res=dev(res), rowscale=dev(rs), rows_per_scale=rows_per)
    ref = (x @ w.t() + bias) * rs.repeat_interleave(rows_per)[:M, None] + res
    assert_close(out, ref, TOL[mode], "linear_fwd")
    # GELU applied to the A operand while staging
    out2 = torch.empty(M, N, device="cuda")
    ops.linear_fwd(dev(x), dev(w), dev(bias), out2, mode, a_op=1)
    assert_close(out2, F.gelu(x) @ w.t() + bias, TOL[mode], "linear_fwd(gelu A)")


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("M,N,K", [(320, 96, 64), (200, 32, 128), (192, 512, 40)])
def test_gemm_backward_layouts(wm, mode, M, N, K):
    """dX = dY W (B row-contiguous) and dW/db = dY^T [X|1] (both row-contiguous, atomics, split-K)."""
    ops = wm.ops
    dy, w, x = synth_array("bdy", (M, N)), synth_array("bw", (N, K), 1.0 / math.sqrt(N)), synth_array("bx", (M, K))
    aux = synth_array("baux", (M, K))
    rows_per = 64
    rs = torch.rand((M + rows_per - 1) // rows_per) + 0.5
    dx = torch.empty(M, K, device="cuda")
    ops.linear_bwd_input(dev(dy), dev(w), dx, mode, rowscale=dev(rs), rows_per_scale=rows_per, gaux=dev(aux))
    g = torch.autograd.functional.jvp(F.gelu, aux, torch.ones_like(aux))[1]
    ref = (dy @ w) * rs.repeat_interleave(rows_per)[:M, None] * g
    assert_close(dx, ref, TOL[mode], "linear_bwd_input")
    ops.linear_bwd_input(dev(dy), dev(w), dx, mode, rowscale=dev(rs), rows_per_scale=rows_per, gaux=dev(aux), gaux_raw=True)
    assert_close(dx, (dy @ w) * rs.repeat_interleave(rows_per)[:M, None] * aux, TOL[mode], "linear_bwd_input raw aux")
    for ks in (0, 1, 3):
        dw = torch.zeros(N, K, device="cuda")
        db = torch.zeros(N, device="cuda")
        A, Bm = dev(dy), dev(x)
        ops.gemm(A, Bm, dw, N, K, M, N, K, K, 0, 0, b_op=1, kscale=dev(rs), k_per_scale=rows_per, b_ones=True,
                 C2=db, atomic_out=True, ksplit=ks, mode=mode)
        dys = dy * rs.repeat_interleave(rows_per)[:M, None]
        assert_close(dw, dys.t() @ F.gelu(x), TOL[mode] * (3 if mode else 1), f"dW ksplit={ks}")
        assert_close(db, dys.sum(0), TOL[mode] * (3 if mode else 1), f"db ksplit={ks}")
    # split-K through a scratch buffer (partial tiles + reduce kernel) instead of atomics; gradients ACCUMULATE into dw
    scr = torch.empty(int(wm.lib.wm_gemm_scratch_floats()), device="cuda")
    for ks in (0, 1, 3):
        dw = torch.ones(N, K, device="cuda")
        db = torch.zeros(N, device="cuda")
        ops.gemm(dev(dy), dev(x), dw, N, K, M, N, K, K, 0, 0, b_op=1, kscale=dev(rs), k_per_scale=rows_per, b_ones=True,
                 C2=db, atomic_out=True, ksplit=ks, mode=mode, scratch=scr)
        assert_close(dw - 1.0, dys.t() @ F.gelu(x), TOL[mode] * (3 if mode else 1), f"dW scratch ksplit={ks}")
        assert_close(db, dys.sum(0), TOL[mode] * (3 if mode else 1), f"db scratch ksplit={ks}")


def test_gemm_rejects_bad_args(wm):
    ops = wm.ops
    a = torch.zeros(8, 6, device="cuda")
    with pytest.raises(wm.WmKernelError):
        ops.gemm(a, a, a, 8, 8, 6, 6, 6, 8)          # K not a multiple of 4


# ---------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("C", [32, 64, 128, 256, 512])
def test_layernorm_fwd_bwd(wm, C):
    ops = wm.ops
    B, hw, shift = 2, 16, 4
    rows = B * hw * hw
    x = synth_array("lnx", (rows, C)).requires_grad_()
    gamma = (1 + 0.1 * synth_array("lng", (C,))).requires_grad_()
    beta = (0.1 * synth_array("lnb", (C,))).requires_grad_()
    use_mod = C >= 64
    mod = synth_array("lnm", (64, C)).requires_grad_() if use_mod else None
    y = F.layer_norm(x, (C,), gamma, beta)
    if use_mod:
        t = torch.roll(y.view(B, hw, hw, C), (-shift, -shift), (1, 2))
        t = t.view(B, hw // 8, 8, hw // 8, 8, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, 64, C) + mod
        t = t.view(B, hw // 8, hw // 8, 8, 8, C).permute(0, 1, 3, 2, 4, 5).reshape(B, hw, hw, C)
        y = torch.roll(t, (shift, shift), (1, 2)).reshape(rows, C)
    dy, dres = synth_array("lndy", (rows, C)), synth_array("lndr", (rows, C))
    y.backward(dy)
    yo = torch.empty(rows, C, device="cuda")
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    xd, gd = dev(x.detach()), dev(gamma.detach())
    ops.layernorm_fwd(xd, gd, dev(beta.detach()), dev(mod.detach()) if use_mod else None, yo, mean, rstd, hw, shift)
    assert_close(yo, y.detach(), 1e-5, "ln fwd")
    dx = torch.empty(rows, C, device="cuda")
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    dm = torch.zeros(64, C, device="cuda") if use_mod else None
    ops.layernorm_bwd(dev(dy), xd, gd, mean, rstd, dev(dres), dx, dg, db, dm, hw, shift)
    assert_close(dx, x.grad + dres, 1e-4, "ln dx")
    assert_close(dg, gamma.grad, 1e-4, "ln dgamma")
    assert_close(db, beta.grad, 1e-4, "ln dbeta")
    if use_mod:
        assert_close(dm, mod.grad, 1e-4, "ln dmod")
    # with a scratch buffer dgamma / dbeta are reduced by a second kernel instead of global atomics
    scr = torch.empty(int(wm.lib.wm_layernorm_bwd_scratch_floats(C)), device="cuda")
    dg2, db2 = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    dxh = torch.empty(rows, C, device="cuda", dtype=torch.bfloat16)      # optional bf16 copy of dx for the next GEMMs
    ops.layernorm_bwd(dev(dy), xd, gd, mean, rstd, dev(dres), dx, dg2, db2, None, hw, shift, scratch=scr, dx_bf16=dxh)
    assert torch.equal(dxh, dx.to(torch.bfloat16))
    assert_close(dg2, gamma.grad, 1e-4, "ln dgamma (scratch)")
    assert_close(db2, beta.grad, 1e-4, "ln dbeta (scratch)")


# ----------------------------------------------------------------------------- window attention
def test_layernorm_bwd_scaled_bf16_copy(wm):
    """dx_bf16 = bf16(dx * copy_scale[sample]): the gradient copy the next dX / dW products read already carries the DropPath
    factor of their branch (model.py:1016-1017), so a sample that dropped the branch gets an all-zero copy while dx itself stays."""
    ops = wm.ops
    B, hw, C = 3, 8, 64
    rows, N = B * hw * hw, hw * hw
    x, gamma, beta = synth_array("cs_x", (rows, C)), 1 + 0.1 * synth_array("cs_g", (C,)), 0.1 * synth_array("cs_b", (C,))
    dy, dres = synth_array("cs_dy", (rows, C)), synth_array("cs_dr", (rows, C))
    xg = x.clone().requires_grad_()
    (F.layer_norm(xg, (C,), gamma, beta) * dy).sum().backward()
    want = xg.grad + dres
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    y = torch.empty(rows, C, device="cuda")
    ops.layernorm_fwd(dev(x), dev(gamma), dev(beta), None, y, mean, rstd, hw, 0)
    scale = torch.tensor([0.0, 1.0 / 0.9, 1.0 / 0.7])
    dx, dxh = torch.empty(rows, C, device="cuda"), torch.empty(rows, C, device="cuda", dtype=torch.bfloat16)
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    ops.layernorm_bwd(dev(dy), dev(x), dev(gamma), mean, rstd, dev(dres), dx, dg, db, None, hw, 0, dx_bf16=dxh,
                      copy_scale=dev(scale), rows_per_scale=N)
    assert_close(dx, want, 1e-4, "ln bwd dx")
    ref_copy = (want * scale.repeat_interleave(N)[:, None]).to(torch.bfloat16).float()
    assert_close(dxh.float(), ref_copy, 8e-3, "ln bwd scaled bf16 copy")
    assert torch.count_nonzero(dxh[:N]) == 0 and torch.count_nonzero(dxh[N:]) > 0
    # without a scale the copy is plain bf16(dx)
    ops.layernorm_bwd(dev(dy), dev(x), dev(gamma), mean, rstd, dev(dres), dx, dg, db, None, hw, 0, dx_bf16=dxh)
    assert_close(dxh.float(), want.to(torch.bfloat16).float(), 8e-3, "ln bwd bf16 copy")


def ref_winattn(qkv, table, B, hw, C, shift):
    """Reference restated with torch ops on natural-order tokens (model.py:523-551, 954-1013)."""
    heads = C // 32
    t = qkv.view(B, hw, hw, 3 * C)
    if shift:
        t = torch.roll(t, (-shift, -shift), (1, 2))
    t = t.view(B, hw // 8, 8, hw // 8, 8, 3 * C).permute(0, 1, 3, 2, 4, 5).reshape(-1, 64, 3, heads, 32)
    q, k, v = (t[:, :, i].transpose(1, 2) for i in range(3))
    s = (q * 32 ** -0.5) @ k.transpose(-1, -2)
    s = s + table[O.rel_pos_index(8).reshape(-1)].view(64, 64, heads).permute(2, 0, 1)
    if shift:
        m = O.shift_attn_mask(hw, 8, shift)
        nw = m.shape[0]
        s = (s.view(B, nw, heads, 64, 64) + m[None, :, None]).view(-1, heads, 64, 64)
    o = (s.softmax(-1) @ v).transpose(1, 2).reshape(-1, 64, C)
    o = o.view(B, hw // 8, hw // 8, 8, 8, C).permute(0, 1, 3, 2, 4, 5).reshape(B, hw, hw, C)
    if shift:
        o = torch.roll(o, (shift, shift), (1, 2))
    return o.reshape(B * hw * hw, C)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("hw,C,shift", [(16, 64, 0), (16, 64, 4), (8, 128, 0), (32, 32, 4)])
def test_winattn_fwd_bwd(wm, mode, hw, C, shift):
    ops = wm.ops
    B, heads = 2, C // 32
    qkv = synth_array("aqkv", (B * hw * hw, 3 * C)).requires_grad_()
    table = (0.5 * synth_array("atab", (225, heads))).requires_grad_()
    ref = ref_winattn(qkv, table, B, hw, C, shift)
    do = synth_array("ado", ref.shape)
    ref.backward(do)
    out = torch.empty(B * hw * hw, C, device="cuda")
    qd, td = dev(qkv.detach()), dev(table.detach())
    lse = torch.empty(B * hw * hw, heads, device="cuda")
    ops.winattn_fwd(qd, td, out, B, hw, C, shift, mode, lse=lse)
    assert_close(out, ref.detach(), TOL[mode], "attn fwd")
    dqkv = torch.empty_like(qd)
    dtab = torch.zeros_like(td)
    ops.winattn_bwd(qd, td, dev(do), dqkv, dtab, B, hw, C, shift, mode)
    assert_close(dqkv, qkv.grad, TOL[mode] * (2 if mode else 1), "attn dqkv")
    assert_close(dtab, table.grad, TOL[mode] * (2 if mode else 1), "attn dtable")
    if mode == 1:       # register-resident backward: needs the forward's output and row log-sum-exp
        dq2, dt2 = torch.empty_like(qd), torch.zeros_like(td)
        ops.winattn_bwd(qd, td, dev(do), dq2, dt2, B, hw, C, shift, mode, out=out, lse=lse)
        assert_close(dq2, qkv.grad, TOL[mode] * 2, "attn dqkv (fast)")
        assert_close(dt2, table.grad, TOL[mode] * 2, "attn dtable (fast)")


@pytest.mark.parametrize("hw,C,shift,B", [(16, 64, 0, 2), (16, 64, 4, 2), (8, 128, 0, 3), (32, 32, 4, 2), (64, 32, 4, 1), (8, 512, 0, 5)])
def test_winattn_split_precision(wm, hw, C, shift, B):
    """compute_mode "split" (mode 2, attn_split.hip): fp32 storage, every product as three bf16 MFMAs of hi / lo halves.
    Against the float64 restatement at north_star's 1e-4 (measured ~1e-6), with and without a scratch / deferred bias-table sum,
    and against the exact fp32 kernels (mode 0)."""
    ops = wm.ops
    heads = C // 32
    qkv = synth_array(f"sqkv{hw}{C}", (B * hw * hw, 3 * C)).double().requires_grad_()
    table = (0.5 * synth_array(f"stab{C}", (225, heads))).double().requires_grad_()
    ref = ref_winattn(qkv, table, B, hw, C, shift)
    do = synth_array(f"sdo{hw}{C}", ref.shape).double()
    ref.backward(do)
    qd, td, dd = dev(qkv.detach().float()), dev(table.detach().float()), dev(do.float())
    out, lse = torch.empty(B * hw * hw, C, device="cuda"), torch.empty(B * hw * hw, heads, device="cuda")
    ops.winattn_fwd(qd, td, out, B, hw, C, shift, 2, lse=lse)
    assert_close(out, ref.detach(), 2e-5, "split attn fwd")
    out0 = torch.empty_like(out)
    ops.winattn_fwd(qd, td, out0, B, hw, C, shift, 0)
    assert_close(out, out0, 2e-5, "split attn fwd vs the fp32 kernel")
    dqkv, dtab = torch.full_like(qd, float("nan")), torch.zeros_like(td)
    ops.winattn_bwd(qd, td, dd, dqkv, dtab, B, hw, C, shift, 2, out=out, lse=lse)
    assert_close(dqkv, qkv.grad, 1e-4, "split attn dqkv")
    assert_close(dtab, table.grad, 1e-4, "split attn dtable")
    # per-block bins through the scratch buffer, summed by the deferred-reduction launch
    scr = torch.empty(int(wm.lib.wm_winattn_bwd_scratch_floats()), device="cuda")
    jobs = ops.ReduceJobs()
    dq2, dt2 = torch.empty_like(qd), torch.zeros_like(td)
    ops.winattn_bwd(qd, td, dd, dq2, dt2, B, hw, C, shift, 2, out=out, lse=lse, scratch=scr, defer=jobs.take(1))
    ops.reduce_jobs(jobs)
    assert torch.equal(dq2, dqkv)
    assert_close(dt2, table.grad, 1e-4, "split attn dtable (deferred)")


# ------------------------------------------------------------------------------------ dwconv
@pytest.mark.parametrize("hw,Ch", [(16, 128), (8, 64), (32, 256), (64, 64), (128, 64)])   # hw >= 64: strip-walking forward
def test_dwconv_fwd_bwd(wm, hw, Ch):
    ops = wm.ops
    B = 2
    h1 = synth_array("dh1", (B, hw, hw, Ch)).requires_grad_()
    w = (0.3 * synth_array("dw", (Ch, 1, 3, 3))).requires_grad_()
    b = (0.1 * synth_array("db", (Ch,))).requires_grad_()
    ref = F.conv2d(F.gelu(h1).permute(0, 3, 1, 2), w, b, padding=1, groups=Ch).permute(0, 2, 3, 1)
    g = synth_array("dg", ref.shape)
    ref.backward(g)
    h2 = torch.empty(B, hw, hw, Ch, device="cuda")
    hd, wd = dev(h1.detach()), dev(w.detach())
    a2, g2 = torch.empty_like(h2), torch.empty_like(h2)
    ops.dwconv_fwd(hd, wd, dev(b.detach()), h2, B, hw, Ch, a2=a2, g2=g2)
    assert_close(h2, ref.detach(), 1e-5, "dwconv fwd")
    assert_close(a2, F.gelu(ref.detach()), 1e-5, "dwconv fwd gelu output")
    gref = torch.autograd.functional.jvp(F.gelu, ref.detach(), torch.ones_like(ref.detach()))[1]
    assert_close(g2, gref, 1e-5, "dwconv fwd gelu' output")
    dh1 = torch.empty_like(hd)
    dwg, dbg = torch.zeros(Ch, 9, device="cuda"), torch.zeros(Ch, device="cuda")
    ops.dwconv_bwd(hd, dev(g), wd, dh1, dwg, dbg, B, hw, Ch)
    assert_close(dh1, h1.grad, 1e-4, "dwconv dh1")
    assert_close(dwg, w.grad.reshape(Ch, 9), 1e-4, "dwconv dw")
    assert_close(dbg, b.grad, 1e-4, "dwconv db")
    # with a scratch buffer the weight / bias gradients are reduced by a second kernel instead of global atomics
    scr = torch.empty(int(wm.lib.wm_dwconv_bwd_scratch_floats(B, hw, Ch)), device="cuda")
    dwg2, dbg2 = torch.zeros_like(dwg), torch.zeros_like(dbg)
    ops.dwconv_bwd(hd, dev(g), wd, dh1, dwg2, dbg2, B, hw, Ch, scratch=scr)
    assert_close(dwg2, w.grad.reshape(Ch, 9), 1e-4, "dwconv dw (scratch)")
    assert_close(dbg2, b.grad, 1e-4, "dwconv db (scratch)")


# --------------------------------------------------------------- down / up sampling data movement
def test_im2col_col2im_match_conv4x4s2(wm):
    ops = wm.ops
    B, hw, C, Co = 2, 16, 32, 64
    x = synth_array("ix", (B, hw, hw, C)).requires_grad_()
    w = synth_array("iw", (Co, C, 4, 4), 0.05)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, None, stride=2, padding=1).permute(0, 2, 3, 1)
    g = synth_array("ig", ref.shape)
    ref.backward(g)
    cols = torch.empty(B * (hw // 2) ** 2, 16 * C, device="cuda")
    ops.im2col(dev(x.detach()), cols, B, hw, C)
    w2 = w.permute(0, 2, 3, 1).reshape(Co, 16 * C)
    assert_close(cols.cpu() @ w2.t(), ref.detach().reshape(-1, Co), 1e-5, "im2col")
    dcols = dev(g.reshape(-1, Co) @ w2)
    dx = torch.empty(B, hw, hw, C, device="cuda")
    ops.col2im(dcols, dx, B, hw, C)
    assert_close(dx, x.grad, 1e-5, "col2im")
    # weight re-layout [o][c][tap] -> [o][tap][c] and back (atomic accumulate)
    w2d = torch.empty(Co, 16, C, device="cuda")
    ops.strided3(dev(w), w2d, (Co, 16, C), (C * 16, 1, 16), (16 * C, C, 1))
    assert torch.equal(w2d.cpu().reshape(Co, 16 * C), w2)
    back = torch.ones(Co, C, 16, device="cuda")
    ops.strided3(w2d, back, (Co, 16, C), (16 * C, C, 1), (C * 16, 1, 16), atomic=True)
    assert_close(back, w.reshape(Co, C, 16) + 1, 1e-6, "strided3 atomic")


def test_pixel_shuffle_matches_convtranspose(wm):
    ops = wm.ops
    B, hw, Ci, Co = 2, 8, 64, 32
    x = synth_array("px", (B, hw, hw, Ci))
    w = synth_array("pw", (Ci, Co, 2, 2), 0.1)
    bias = synth_array("pb", (Co,))
    ref = F.conv_transpose2d(x.permute(0, 3, 1, 2), w, bias, stride=2).permute(0, 2, 3, 1)
    wr = w.permute(2, 3, 1, 0).reshape(4 * Co, Ci)              # rows (di,dj,o)
    y4 = dev(x.reshape(-1, Ci) @ wr.t())
    dst = torch.full((B, 2 * hw, 2 * hw, 2 * Co), 7.0, device="cuda")
    ops.pixel_shuffle(y4, dev(bias), dst, B, hw, Co, 2 * Co, Co)
    assert_close(dst[..., Co:], ref, 1e-6, "pixel_shuffle")
    assert torch.all(dst[..., :Co] == 7.0)
    back = torch.empty(B * hw * hw, 4 * Co, device="cuda")
    ops.pixel_unshuffle(dst, back, B, hw, Co, 2 * Co, Co)
    assert_close(back, y4.cpu() + bias.repeat(4), 1e-6, "pixel_unshuffle")
    src = dev(synth_array("cc", (50, 96)))
    d2 = torch.zeros(50, 64, device="cuda")
    ops.copy_cols(src, 96, 32, d2, 64, 16, 50, 32)
    ops.copy_cols(src, 96, 32, d2, 64, 16, 50, 32, accumulate=True)
    assert torch.equal(d2[:, 16:48].cpu(), 2 * src[:, 32:64].cpu()) and d2[:, :16].abs().sum() == 0


def test_msg_expand(wm):
    ops = wm.ops
    B = 3
    f = synth_array("mf", (B, 4, 64)).requires_grad_()
    ref = f.repeat(1, 16, 8)
    g = synth_array("mg", (B, 64, 1024))
    ref.backward(g[..., :512])
    dst = torch.zeros(B, 64, 1024, device="cuda")
    ops.msg_expand_fwd(dev(f.detach()), dst, B)
    assert torch.equal(dst[..., :512].cpu(), ref.detach())
    df = torch.ones(B, 256, device="cuda")
    ops.msg_expand_bwd(dev(g), df, B)
    assert_close(df, f.grad.reshape(B, 256) + 1, 1e-5, "msg_expand_bwd")


# ---------------------------------------------------------------------------------- small convs
CONVS = [
    # Cin, H, W, Cout, k, stride, pad, in_nhwc, out_nhwc, act, transposed
    (2, 32, 32, 32, (3, 3), (1, 1), (1, 1), False, True, "leaky", False),      # InputProj
    (64, 16, 16, 2, (3, 3), (1, 1), (1, 1), True, False, "none", False),       # OutputProj
    (2, 16, 48, 64, (3, 3), (1, 1), (1, 1), False, True, "leaky", False),      # fan-out kernel, 64 channels, several tiles
    (32, 24, 16, 2, (3, 3), (1, 1), (1, 1), True, False, "relu", False),       # fan-in kernel, 32 channels, activation
    (2, 24, 24, 4, (3, 3), (1, 1), (1, 1), False, False, "relu", False),       # stft_layer.0
    (1, 32, 32, 16, (3, 3), (1, 1), (1, 1), False, False, "relu", False),      # ConvAutoencoder.conv1
    (1, 64, 512, 1, (8, 8), (16, 8), (0, 0), False, False, "none", False),     # EncoderTransformerWM.conv2
    (4, 8, 8, 16, (2, 2), (2, 2), (0, 0), False, False, "relu", True),         # t_conv1
    (16, 16, 16, 1, (2, 2), (2, 2), (0, 0), False, False, "sigmoid", True),    # t_conv2
]


@pytest.mark.parametrize("cfg", CONVS)
def test_small_conv_fwd_bwd(wm, cfg):
    ops = wm.ops
    Cin, H, W, Cout, k, st, pad, inl, outl, act, tr = cfg
    B = 2
    x = synth_array("cx", (B, Cin, H, W)).requires_grad_()
    wshape = (Cin, Cout) + k if tr else (Cout, Cin) + k
    w = synth_array("cw", wshape, 1.0 / math.sqrt(Cin * k[0] * k[1])).requires_grad_()
    b = (0.1 * synth_array("cb", (Cout,))).requires_grad_()
    pre = F.conv_transpose2d(x, w, b, stride=st) if tr else F.conv2d(x, w, b, stride=st, padding=pad)
    y = {"none": lambda t: t, "relu": F.relu, "leaky": lambda t: F.leaky_relu(t, 0.01), "sigmoid": torch.sigmoid}[act](pre)
    g = synth_array("cg", y.shape)
    y.backward(g)
    d = ops.conv_desc(B, Cin, H, W, Cout, k, st, pad, inl, outl, act, tr)
    xin = x.detach().permute(0, 2, 3, 1) if inl else x.detach()
    yo = torch.empty((B, d.OH, d.OW, Cout) if outl else (B, Cout, d.OH, d.OW), device="cuda")
    xd, wd = dev(xin), dev(w.detach())
    ops.conv_fwd(d, xd, wd, dev(b.detach()), yo)
    yref = y.detach().permute(0, 2, 3, 1) if outl else y.detach()
    assert_close(yo, yref, 1e-5, "conv fwd")
    gd = dev(g.permute(0, 2, 3, 1) if outl else g)
    dx = torch.empty_like(xd)
    dw, db = torch.zeros_like(wd), torch.zeros(Cout, device="cuda")
    ops.conv_bwd(d, xd, wd, yo, gd, dx, dw, db)
    assert_close(dx, x.grad.permute(0, 2, 3, 1) if inl else x.grad, 1e-4, "conv dx")
    assert_close(dw, w.grad, 1e-4, "conv dw")
    assert_close(db, b.grad, 1e-4, "conv db")


@pytest.mark.parametrize("planes,H,W,kh,kw", [(8, 32, 32, 2, 2), (3, 64, 512, 16, 8)])
def test_maxpool(wm, planes, H, W, kh, kw):
    ops = wm.ops
    x = synth_array("mpx", (1, planes, H, W)).requires_grad_()
    y = F.max_pool2d(x, (kh, kw))
    g = synth_array("mpg", y.shape)
    y.backward(g)
    yo = torch.empty(planes, H // kh, W // kw, device="cuda")
    xd = dev(x.detach()[0])
    ops.maxpool_fwd(xd, yo, planes, H, W, kh, kw)
    assert torch.equal(yo.cpu(), y.detach()[0])
    dx = torch.ones_like(xd)
    ops.maxpool_bwd(xd, dev(g[0]), dx, planes, H, W, kh, kw, accumulate=True)
    assert_close(dx, x.grad[0] + 1, 1e-6, "maxpool bwd")


# ------------------------------------------------------------------------------- STFT round trip
@pytest.mark.parametrize("terms", [2, 3])
def test_stft_istft_fwd_bwd(wm, golden, terms):
    """terms = 2: hi + lo operands (bf16 compute mode); 3: hi + mid + lo (fp32 / split modes), same bounds against the fp32 oracle
    here — test_stft_three_terms_are_fp32_grade holds the tighter one against float64."""
    ops = wm.ops
    B = 3
    spec = synth_array("sspec", (B, 2, 128, 128)).requires_grad_()
    wave_ref = O.istft255(spec)
    gw = synth_array("sgw", (B, 8002))
    wave_ref.backward(gw)
    wave = torch.empty(B, 8002, device="cuda")
    ops.istft_fwd(dev(spec.detach()), wave, B, terms)
    assert_close(wave, wave_ref.detach(), 1e-4, "istft fwd")
    dspec = torch.empty(B, 2, 128, 128, device="cuda")
    ops.istft_bwd(dev(gw), dspec, B, terms)
    assert_close(dspec, spec.grad, 1e-4, "istft bwd")
    w = (0.2 * synth_array("sw", (B, 8002))).requires_grad_()
    s_ref = O.stft255(w)
    gs = synth_array("sgs", s_ref.shape)
    s_ref.backward(gs)
    s = torch.empty(B, 2, 128, 128, device="cuda")
    ops.stft_fwd(dev(w.detach()), s, B, terms)
    assert_close(s, s_ref.detach(), 1e-4, "stft fwd")
    dw = torch.empty(B, 8002, device="cuda")
    ops.stft_bwd(dev(gs), dw, B, terms)
    assert_close(dw, w.grad, 1e-4, "stft bwd")
    # golden produced by the reference's own torch.istft/torch.stft calls (model.py:2458,2463)
    g2 = golden("g2_stft")
    from synth import synth_tile
    y = synth_tile(2, "g2_y", 1.0)
    wv = torch.empty(2, 8002, device="cuda")
    ops.istft_fwd(dev(y), wv, 2, terms)
    assert_close(wv, torch.from_numpy(g2["wave"]), 1e-4, "istft vs golden")
    sp = torch.empty(2, 2, 128, 128, device="cuda")
    ops.stft_fwd(dev(torch.from_numpy(g2["wave"])), sp, 2, terms)
    assert_close(sp, torch.from_numpy(g2["spec"]), 1e-4, "stft vs golden")


def test_stft_three_terms_are_fp32_grade(wm):
    """The fp32 / split compute modes' transforms (operands as three bf16 terms, six products) against float64 torch.stft / istft:
    5e-7 of full scale where the two-term form measures 2e-6 .. 4e-6; adjoint identities at 1e-6 of the norms; a batch with several
    items per persistent workgroup and padding items (B = 70 on 256 workgroups: 280 analysis items, 360 synthesis items of which 10
    are padding), results bit-identical across launches."""
    ops = wm.ops
    B = 70
    g = torch.Generator().manual_seed(5)
    w = (0.2 * torch.randn(B, 8002, generator=g)).cuda()
    sp = torch.randn(B, 2, 128, 128, generator=g).cuda()
    win = torch.ones(255, dtype=torch.float64)
    st = torch.view_as_real(torch.stft(w.double().cpu(), n_fft=255, window=win, return_complex=True)).permute(0, 3, 1, 2)
    wv = torch.istft(torch.complex(sp[:, 0].double().cpu(), sp[:, 1].double().cpu()), n_fft=255, window=win)
    outs = {}
    for terms in (3, 2, 3):
        s, x = torch.empty(B, 2, 128, 128, device="cuda"), torch.empty(B, 8002, device="cuda")
        ds, dx = torch.empty(B, 2, 128, 128, device="cuda"), torch.empty(B, 8002, device="cuda")
        ops.stft_fwd(w, s, B, terms)
        ops.istft_fwd(sp, x, B, terms)
        ops.istft_bwd(w, ds, B, terms)
        ops.stft_bwd(sp, dx, B, terms)
        if terms in outs:
            for a, b_ in zip((s, x, ds, dx), outs[terms]):
                assert torch.equal(a, b_)
        outs[terms] = (s, x, ds, dx)
    s3, x3, ds3, dx3 = outs[3]
    s2, x2, _, _ = outs[2]

    def err(a, ref):
        return ((a.double().cpu() - ref).abs().max() / ref.abs().max()).item()
    e3, e2 = (err(s3, st), err(x3, wv)), (err(s2, st), err(x2, wv))
    assert max(e3) <= 5e-7, e3
    assert max(e2) <= 1e-5 and min(e2) >= 2 * max(e3), (e2, e3)      # the two-term form is the coarser one, as documented
    for b in (0, B // 2, B - 1):
        assert err(s3[b], st[b]) <= 5e-7 and err(x3[b], wv[b]) <= 5e-7
    lhs, rhs = (x3.double() * w.double()).sum().item(), (sp.double() * ds3.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-6 * (x3.double().norm() * w.double().norm()).item(), (lhs, rhs)
    lhs, rhs = (s3.double() * sp.double()).sum().item(), (w.double() * dx3.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-6 * (s3.double().norm() * sp.double().norm()).item(), (lhs, rhs)
    with pytest.raises(wm._lib.WmKernelError):
        ops.check(wm.lib.wm_stft255_fwd(ops.ptr(w), ops.ptr(s3), ops.ptr(ops.stft_tables(w.device, 3)), 4, B, None), "wm_stft255_fwd")


@pytest.mark.parametrize("B", [96, 131, 260])
def test_stft_round_trip_is_deterministic_under_load(wm, B):
    """The four n_fft = 255 transforms at bench-like batches (several workgroups resident per CU), launched repeatedly:
    bit-identical results every time and within 1e-4 of torch.stft / torch.istft (float64) for every tile.  A build of
    these kernels with packed-fp32 folds failed exactly this on MI355X (wrong sums in one half-wave under load).
    B = 96: one (tile, frame group) item per persistent workgroup; 131: two or three items each, a batch that is no multiple of 8
    (padding items in the synthesis grid); 260: beyond the bench batch."""
    ops = wm.ops
    g = torch.Generator().manual_seed(3)
    w = (0.2 * torch.randn(B, 8002, generator=g)).cuda()
    sp = torch.randn(B, 2, 128, 128, generator=g).cuda()
    win = torch.ones(255, dtype=torch.float64)
    st = torch.view_as_real(torch.stft(w.double().cpu(), n_fft=255, window=win, return_complex=True)).permute(0, 3, 1, 2)
    wv = torch.istft(torch.complex(sp[:, 0].double().cpu(), sp[:, 1].double().cpu()), n_fft=255, window=win)
    outs = None
    for rep in range(12 if B == 96 else 4):
        s, x = torch.empty(B, 2, 128, 128, device="cuda"), torch.empty(B, 8002, device="cuda")
        ds, dx = torch.empty(B, 2, 128, 128, device="cuda"), torch.empty(B, 8002, device="cuda")
        ops.stft_fwd(w, s, B)
        ops.istft_fwd(sp, x, B)
        ops.istft_bwd(w, ds, B)
        ops.stft_bwd(sp, dx, B)
        cur = (s, x, ds, dx)
        if outs is None:
            outs = cur
            assert_close(s, st, 1e-4, f"stft at B = {B}")
            assert_close(x, wv, 1e-4, f"istft at B = {B}")
            # per tile, not only over the batch: a wrong tile index in the persistent loop would hide behind the batch maximum
            for b in (0, B // 2, B - 1):
                assert_close(s[b], st[b], 1e-4, f"stft tile {b}")
                assert_close(x[b], wv[b], 1e-4, f"istft tile {b}")
            # adjoint identities tie the backward kernels to the forward ones
            # (the inner products of random operands nearly cancel: the bound is relative to the norms, not to the sum)
            lhs, rhs = (x.double() * w.double()).sum().item(), (sp.double() * ds.double()).sum().item()
            assert abs(lhs - rhs) <= 1e-5 * (x.double().norm() * w.double().norm()).item(), (lhs, rhs)
            lhs, rhs = (s.double() * sp.double()).sum().item(), (w.double() * dx.double()).sum().item()
            assert abs(lhs - rhs) <= 1e-5 * (s.double().norm() * sp.double().norm()).item(), (lhs, rhs)
        else:
            for a, b_, name in zip(cur, outs, ("stft_fwd", "istft_fwd", "istft_bwd", "stft_bwd")):
                assert torch.equal(a, b_), f"{name}: launch {rep} differs from launch 0"


# ---------------------------------------------------------------------------------- attacks
def test_attacks_fwd_and_adjoint(wm):
    att = wm.attacks
    ops = wm.ops
    B, L = 3, 8002
    x = 0.1 * synth_array("atx", (B, L))
    xd = dev(x)
    step = torch.zeros(1, dtype=torch.int64, device="cuda")
    # 1) noise extraction: AWGN only
    cfg = att.AttackConfig(resample=False, snr_db=20.0, lowpass=False, amp=None, seed=7)
    out, sigma = att.attack_forward(cfg, xd, step)
    sig_ref = OA.awgn_sigma(x.numpy(), 20.0)
    assert_close(sigma.cpu()[:, None], torch.from_numpy(sig_ref), 1e-4, "awgn sigma")
    z = ((out.cpu() - x) / sigma.cpu()[:, None]).double().numpy()
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02
    snr = 10 * np.log10((x.numpy() ** 2).mean(-1) / ((out.cpu().numpy() - x.numpy()) ** 2).mean(-1))
    assert np.all(np.abs(snr - 20.0) < 0.3)
    # 2) each deterministic stage and the full chain against the oracle (noise replayed through z)
    for kw in (dict(resample=True), dict(lowpass=True), dict(amp=0.8), dict(resample=True, lowpass=True, amp=0.8)):
        c = att.AttackConfig(resample=kw.get("resample", False), snr_db=None, lowpass=kw.get("lowpass", False),
                             amp=kw.get("amp"), seed=7)
        o, _ = att.attack_forward(c, xd, step)
        ref = OA.attack_chain(x.numpy(), resample=c.resample, lowpass=c.lowpass, amp=c.amp)
        assert_close(o, torch.from_numpy(ref), 1e-4, f"attack {kw}")
    full = att.AttackConfig(resample=True, snr_db=20.0, lowpass=True, amp=0.8, seed=7)
    o, sg = att.attack_forward(full, xd, step)
    rs = OA.resampling(x.numpy())
    o1, _ = att.attack_forward(att.AttackConfig(True, 20.0, False, None, 7), xd, step)
    zz = (o1.cpu().double().numpy() - rs) / OA.awgn_sigma(rs, 20.0)
    ref = OA.attack_chain(x.numpy(), resample=True, snr_db=20.0, z=zz, lowpass=True, amp=0.8)
    assert_close(o, torch.from_numpy(ref), 2e-4, "full attack chain")
    # 3) backward = exact adjoint of the linear stages: <A x, g> == <x, A^T g>
    lin = att.AttackConfig(resample=True, snr_db=None, lowpass=True, amp=0.8, seed=7)
    g = synth_array("atg", (B, L))
    ax, _ = att.attack_forward(lin, xd, step)
    atg = att.attack_backward(lin, xd, dev(g), None, step)
    lhs = (ax.double().cpu() * g.double()).sum(-1)
    rhs = (x.double() * atg.double().cpu()).sum(-1)
    assert_close(lhs, rhs, 1e-4, "attack adjoint")
    # 4) AWGN backward (sigma depends on the signal): directional derivative check in float64 on the oracle
    c = att.AttackConfig(resample=False, snr_db=10.0, lowpass=False, amp=None, seed=7)
    o, sg = att.attack_forward(c, xd, step)
    zz = ((o.cpu() - x) / sg.cpu()[:, None]).double().numpy()
    dgr = att.attack_backward(c, xd, dev(g), sg, step).cpu().double().numpy()
    v = synth_array("atv", (B, L)).double().numpy()
    eps = 1e-6
    f = lambda xx: (OA.awgn(xx, 10.0, zz) * g.double().numpy()).sum()
    fd = (f(x.double().numpy() + eps * v) - f(x.double().numpy() - eps * v)) / (2 * eps)
    assert abs(fd - (dgr * v).sum()) <= 2e-3 * abs(fd) + 1e-6


def test_attack_lowpass_on_matrix_cores_equals_scalar_loops(wm):
    """The MFMA form of the two 128-tap FIR passes (Toeplitz operand tables, split-precision bf16) against the fp32 scalar
    loops of the same kernel, forward and backward, full stack incl. AWGN, at a batch that fills every CU."""
    att = wm.attacks
    B, L = 300, 8002
    g = torch.Generator().manual_seed(11)
    x = (0.1 * torch.randn(B, L, generator=g)).cuda()
    gy = torch.randn(B, L, generator=g).cuda()
    step = torch.full((1,), 3, dtype=torch.int64, device="cuda")
    res = {}
    for mfma in (True, False):
        cfg = att.AttackConfig(resample=True, snr_db=20.0, lowpass=True, amp=0.8, seed=7, mfma_fir=mfma)
        y, sg = att.attack_forward(cfg, x, step)
        dx = att.attack_backward(cfg, x, gy, sg, step)
        res[mfma] = (y, dx)
    assert_close(res[True][0], res[False][0], 2e-5, "attack forward: MFMA FIR vs scalar FIR")
    assert_close(res[True][1], res[False][1], 2e-5, "attack backward: MFMA FIR vs scalar FIR")
    y2, _ = att.attack_forward(att.AttackConfig.full_stack(), x, step)
    assert torch.equal(y2, res[True][0])                                     # deterministic
    # odd and short rows through the edge paths of the half-band resampler, against the oracle
    for L2 in (801, 64, 8001):
        xs = 0.1 * torch.randn(2, L2, generator=g)
        c = att.AttackConfig(resample=True, lowpass=True, amp=0.8)
        o, _ = att.attack_forward(c, xs.cuda(), step)
        assert_close(o, torch.from_numpy(OA.attack_chain(xs.numpy(), resample=True, lowpass=True, amp=0.8)), 1e-4, f"L = {L2}")
        gg = torch.randn(2, L2, generator=g)
        atg = att.attack_backward(c, xs.cuda(), gg.cuda(), None, step)
        lhs, rhs = (o.double().cpu() * gg.double()).sum(-1), (xs.double() * atg.double().cpu()).sum(-1)
        assert_close(lhs, rhs, 1e-4, f"adjoint identity, L = {L2}")


def test_attack_row_n3_stages_fwd_and_adjoint(wm):
    """In-loop echo / jittering_2 / requantisation behind the four BASELINE attacks."""
    att = wm.attacks
    B, L = 3, 8002
    x = 0.4 * synth_array("n3x", (B, L))
    xd = dev(x)
    step = torch.full((1,), 5, dtype=torch.int64, device="cuda")
    # echo after low-pass + amplitude scale, against the oracle chain
    c = att.AttackConfig(lowpass=True, amp=0.8, echo=(3000, 0.2))
    o, _ = att.attack_forward(c, xd, step)
    assert_close(o, torch.from_numpy(OA.attack_chain(x.numpy(), lowpass=True, amp=0.8, echo=(3000, 0.2))), 1e-4, "echo")
    # requantisation alone: bit exact against the PCM_U8 rule
    big = 3.0 * x
    o, _ = att.attack_forward(att.AttackConfig(requantize=True), dev(big), step)
    assert np.array_equal(o.cpu().numpy(), OA.requantization(big.numpy()))
    # jitter: positions read back from the output, replayed through the oracle; deterministic in (seed, step)
    cj = att.AttackConfig(amp=0.5, jitter=300, seed=11)
    o, _ = att.attack_forward(cj, xd, step)
    hit = (o.cpu() == 0) & (x != 0)
    counts = hit.sum(1)
    assert ((counts >= 280) & (counts <= 300)).all(), counts
    idx = [torch.nonzero(h)[:, 0].numpy() for h in hit]
    assert_close(o, torch.from_numpy(OA.attack_chain(x.numpy(), amp=0.5, jitter_idx=idx)), 1e-6, "jitter")
    assert not np.array_equal(idx[0], idx[1])                                   # rows draw their own positions
    o2, _ = att.attack_forward(cj, xd, step)
    assert torch.equal(o, o2)
    o3, _ = att.attack_forward(cj, xd, step + 1)
    assert not torch.equal(o, o3)
    # all three together with the full BASELINE stack in front: runs, finite, 8-bit grid
    full = att.AttackConfig(True, 20.0, True, 0.8, 7, echo=(4000, 0.2), jitter=100, requantize=True)
    o, sg = att.attack_forward(full, xd, step)
    assert torch.isfinite(o).all() and torch.equal(o * 128, (o * 128).round())
    # backward = exact adjoint of the linear stages (echo + jitter mask behind resample / low-pass / amp)
    lin = att.AttackConfig(resample=True, lowpass=True, amp=0.8, echo=(3000, 0.2), jitter=200, seed=3)
    g = synth_array("n3g", (B, L))
    ax, _ = att.attack_forward(lin, xd, step)
    atg = att.attack_backward(lin, xd, dev(g), None, step)
    lhs = (ax.double().cpu() * g.double()).sum(-1)
    rhs = (x.double() * atg.double().cpu()).sum(-1)
    assert_close(lhs, rhs, 1e-4, "row-N3 adjoint")
    # requantisation passes the gradient straight through
    ga = att.attack_backward(att.AttackConfig(amp=0.8, requantize=True), xd, dev(g), None, step)
    assert_close(ga, 0.8 * g, 1e-6, "requantisation straight-through")
    assert att.AttackConfig.from_string("echo_addition+jittering_2-50+requantization").flags == 16 | 32 | 64


# ------------------------------------------------------------------------------- loss, AdamW, rng
def test_loss_fwd_bwd(wm):
    ops = wm.ops
    B = 3
    x = synth_array("lx", (B, 2, 128, 128))
    ts = [synth_array("l" + n, s).requires_grad_() for n, s in
          (("a", (B, 2, 128, 128)), ("n", (B, 2, 128, 128)), ("p", (B, 1, 32, 32)), ("w", (B, 1, 32, 32)))]
    msg = (synth_array("lm", (B, 1, 32, 32)) > 0).float()
    tot, terms = O.train_losses(ts, x, msg)
    tot.backward()
    losses = torch.zeros(8, device="cuda")
    grads = [torch.empty_like(dev(t.detach())) for t in ts]
    ops.loss_fwd_bwd(*[dev(t.detach()) for t in (ts[0],)], dev(x), dev(ts[1].detach()), dev(ts[2].detach()),
                     dev(ts[3].detach()), dev(msg), losses, grads[0], grads[1], grads[2], grads[3], B,
                     torch.zeros(8, device="cuda"))
    assert_close(losses[:4], torch.stack(terms).detach(), 1e-5, "loss terms")
    assert_close(losses[4], tot.detach(), 1e-5, "loss total")
    for gk, t in zip(grads, ts):
        assert_close(gk, t.grad, 1e-4, "loss grad")


def test_adamw_matches_torch(wm):
    ops = wm.ops
    n = 10007
    p0 = synth_array("op", (n,))
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p], lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)
    pd, m, v = dev(p0), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    step = torch.zeros(1, dtype=torch.int64, device="cuda")
    lr = torch.tensor([2e-4], device="cuda")
    for it in range(3):
        g = synth_array(f"og{it}", (n,))
        p.grad = g.clone()
        opt.step()
        ops.step_inc(step)
        ops.adamw(pd, dev(g), m, v, n, lr, 0.9, 0.999, 1e-8, 0.02, step)
    assert int(step.item()) == 3
    assert_close(pd, p.detach(), 1e-6, "adamw")


def test_droppath_scales(wm):
    ops = wm.ops
    nslots, B = 8, 4096
    rates = torch.tensor([0.0, 0.1, 0.1, 0.5, 0.0, 0.25, 0.1, 0.1], device="cuda")
    s = torch.empty(nslots, B, device="cuda")
    step = torch.zeros(1, dtype=torch.int64, device="cuda")
    ops.droppath_scales(s, rates, nslots, B, 1234, step)
    sc = s.cpu()
    for i, r in enumerate(rates.cpu().tolist()):
        vals = sc[i].unique().numpy()
        if r == 0:
            assert np.array_equal(vals, [1.0])
        else:
            assert all(np.isclose(v, 0.0) or np.isclose(v, 1 / (1 - r), rtol=1e-5) for v in vals)
            assert abs((sc[i] > 0).float().mean().item() - (1 - r)) < 0.03
    s2 = torch.empty_like(s)
    ops.droppath_scales(s2, rates, nslots, B, 1234, step)
    assert torch.equal(s, s2)                       # pure function of (seed, step)
    ops.step_inc(step)
    ops.droppath_scales(s2, rates, nslots, B, 1234, step)
    assert not torch.equal(s, s2)


# ------------------------------------------------------------------ bf16 storage of the wide activations
def bfr(t):
    """Round-trip through bf16 (what a bf16-stored tensor holds)."""
    return t.to(torch.bfloat16).float()


@pytest.mark.parametrize("a_bf,b_bf,c_bf", [(1, 1, 1), (1, 1, 0), (0, 1, 1), (1, 0, 0)])
def test_gemm_bf16_storage(wm, a_bf, b_bf, c_bf):
    ops = wm.ops
    M, N, K = 300, 200, 192
    x, w = synth_array("hx", (M, K)), synth_array("hw", (N, K), 1.0 / math.sqrt(K))
    bias, aux = synth_array("hb", (N,)), synth_array("haux", (M, N))
    xs = dev(x).bfloat16() if a_bf else dev(x)
    ws = dev(w).bfloat16() if b_bf else dev(w)
    xr, wr = (bfr(x) if a_bf else x), (bfr(w) if b_bf else w)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16 if c_bf else torch.float32)
    ops.linear_fwd(xs, ws, dev(bias), out, 1)
    assert_close(out.float(), xr @ wr.t() + bias, 2e-2, "fwd")
    ops.linear_fwd(xs, ws, dev(bias), out, 1, a_op=1)
    assert_close(out.float(), F.gelu(xr) @ wr.t() + bias, 2e-2, "fwd gelu(A)")
    # dX: B row-contiguous ([N_out=K', K_in=N'] weight), gelu' operand stored like C
    dy = synth_array("hdy", (M, K))
    w2 = synth_array("hw2", (K, N), 1.0 / math.sqrt(K))            # [out=K, in=N]
    dys = dev(dy).bfloat16() if a_bf else dev(dy)
    w2s = dev(w2).bfloat16() if b_bf else dev(w2)
    auxs = dev(aux).bfloat16() if c_bf else dev(aux)
    dx = torch.empty(M, N, device="cuda", dtype=torch.bfloat16 if c_bf else torch.float32)
    ops.linear_bwd_input(dys, w2s, dx, 1, gaux=auxs)
    g = torch.autograd.functional.jvp(F.gelu, bfr(aux) if c_bf else aux, torch.ones_like(aux))[1]
    assert_close(dx.float(), ((bfr(dy) if a_bf else dy) @ (bfr(w2) if b_bf else w2)) * g, 2e-2, "dX")
    # dW: both operands row-contiguous, ones row, kscale, atomics into fp32
    act = synth_array("hact", (M, N))
    acts = dev(act).bfloat16() if b_bf else dev(act)
    rs = torch.rand((M + 63) // 64) + 0.5
    dw, db = torch.zeros(K, N, device="cuda"), torch.zeros(K, device="cuda")
    ops.linear_bwd_weight(dys, acts, dw, db, 1, x_op=1, kscale=dev(rs), k_per_scale=64)
    dyr = (bfr(dy) if a_bf else dy) * rs.repeat_interleave(64)[:M, None]
    assert_close(dw, dyr.t() @ F.gelu(bfr(act) if b_bf else act), 3e-2, "dW")
    assert_close(db, dyr.sum(0), 3e-2, "db")


def test_elementwise_kernels_bf16_storage(wm):
    ops = wm.ops
    B, hw, C = 2, 16, 64
    rows = B * hw * hw
    bf = torch.bfloat16
    # LayerNorm: bf16 output / bf16 incoming gradient
    x, gamma, beta = synth_array("ex", (rows, C)), 1 + 0.1 * synth_array("eg", (C,)), 0.1 * synth_array("eb", (C,))
    y = torch.empty(rows, C, device="cuda", dtype=bf)
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    ops.layernorm_fwd(dev(x), dev(gamma), dev(beta), None, y, mean, rstd, hw, 0)
    assert_close(y.float(), bfr(F.layer_norm(x, (C,), gamma, beta)), 1e-2, "ln fwd bf16")
    xg = x.clone().requires_grad_()
    dy = bfr(synth_array("edy", (rows, C)))
    F.layer_norm(xg, (C,), gamma, beta).backward(dy)
    dx = torch.empty(rows, C, device="cuda")
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    ops.layernorm_bwd(dev(dy).to(bf), dev(x), dev(gamma), mean, rstd, None, dx, dg, db, None, hw, 0)
    assert_close(dx, xg.grad, 1e-4, "ln bwd bf16 dy")
    # window attention with bf16 q|k|v, output and gradients
    heads = C // 32
    qkv = bfr(synth_array("eqkv", (rows, 3 * C))).requires_grad_()
    table = (0.5 * synth_array("etab", (225, heads))).requires_grad_()
    ref = ref_winattn(qkv, table, B, hw, C, 4)
    do = bfr(synth_array("edo", ref.shape))
    ref.backward(do)
    out = torch.empty(rows, C, device="cuda", dtype=bf)
    qd = dev(qkv.detach()).to(bf)
    lse = torch.empty(B * hw * hw, C // 32, device="cuda")
    ops.winattn_fwd(qd, dev(table.detach()), out, B, hw, C, 4, 1, lse=lse)
    assert_close(out.float(), ref.detach(), 3e-2, "attn fwd bf16 io")
    dqkv, dtab = torch.empty_like(qd), torch.zeros(225, heads, device="cuda")
    ops.winattn_bwd(qd, dev(table.detach()), dev(do).to(bf), dqkv, dtab, B, hw, C, 4, 1, out=out, lse=lse)
    assert_close(dqkv.float(), qkv.grad, 5e-2, "attn bwd bf16 io")
    # depthwise conv
    Ch = 128
    h1 = bfr(synth_array("eh1", (B, hw, hw, Ch))).requires_grad_()
    w = (0.3 * synth_array("ew", (Ch, 1, 3, 3))).requires_grad_()
    b = 0.1 * synth_array("ebb", (Ch,))
    refc = F.conv2d(F.gelu(h1).permute(0, 3, 1, 2), w, b, padding=1, groups=Ch).permute(0, 2, 3, 1)
    g = bfr(synth_array("egc", refc.shape))
    refc.backward(g)
    h2 = torch.empty(B, hw, hw, Ch, device="cuda", dtype=bf)
    hd = dev(h1.detach()).to(bf)
    ops.dwconv_fwd(hd, dev(w.detach()), dev(b), h2, B, hw, Ch)
    assert_close(h2.float(), refc.detach(), 1e-2, "dwconv fwd bf16")
    dh1 = torch.empty_like(hd)
    dwg, dbg = torch.zeros(Ch, 9, device="cuda"), torch.zeros(Ch, device="cuda")
    ops.dwconv_bwd(hd, dev(g).to(bf), dev(w.detach()), dh1, dwg, dbg, B, hw, Ch)
    assert_close(dh1.float(), h1.grad, 1e-2, "dwconv bwd bf16")
    assert_close(dwg, w.grad.reshape(Ch, 9), 1e-3, "dwconv dw bf16")
    # the strip-walking forward (hw >= 64) in bf16 io: h2, gelu(h2) and gelu'(h2) against torch on the bf16-rounded input
    hs, Cs_ = 64, 64
    h1s = bfr(synth_array("es1", (1, hs, hs, Cs_)))
    ws, bs = 0.3 * synth_array("esw", (Cs_, 1, 3, 3)), 0.1 * synth_array("esb", (Cs_,))
    refs = F.conv2d(F.gelu(h1s).permute(0, 3, 1, 2), ws, bs, padding=1, groups=Cs_).permute(0, 2, 3, 1)
    h2s = torch.empty(1, hs, hs, Cs_, device="cuda", dtype=bf)
    a2s, g2s = torch.empty_like(h2s), torch.empty_like(h2s)
    ops.dwconv_fwd(dev(h1s).to(bf), dev(ws), dev(bs), h2s, 1, hs, Cs_, a2=a2s, g2=g2s)
    assert_close(h2s.float(), refs, 1e-2, "dwconv strip fwd bf16")
    assert_close(a2s.float(), F.gelu(refs), 1e-2, "dwconv strip gelu bf16")
    grefs = torch.autograd.functional.jvp(F.gelu, refs, torch.ones_like(refs))[1]
    assert_close(g2s.float(), grefs, 1e-2, "dwconv strip gelu' bf16")
    # im2col / col2im / pixel (un)shuffle / strided3 / cast
    xt = synth_array("ei", (B, hw, hw, 32))
    cols32 = torch.empty(B * 64, 16 * 32, device="cuda")
    cols16 = torch.empty(B * 64, 16 * 32, device="cuda", dtype=bf)
    ops.im2col(dev(xt), cols32, B, hw, 32)
    ops.im2col(dev(xt), cols16, B, hw, 32)
    assert torch.equal(cols16.float(), bfr(cols32.cpu()).cuda())
    d32, d16 = torch.empty(B, hw, hw, 32, device="cuda"), torch.empty(B, hw, hw, 32, device="cuda")
    ops.col2im(cols16.float(), d32, B, hw, 32)
    ops.col2im(cols16, d16, B, hw, 32)
    assert_close(d16, d32, 1e-6, "col2im bf16")
    y4 = synth_array("ey4", (B * 64, 4 * 32))
    a32, a16 = torch.zeros(B, 16, 16, 32, device="cuda"), torch.zeros(B, 16, 16, 32, device="cuda")
    ops.pixel_shuffle(dev(bfr(y4)), None, a32, B, 8, 32, 32, 0)
    ops.pixel_shuffle(dev(y4).to(bf), None, a16, B, 8, 32, 32, 0)
    assert torch.equal(a32, a16)
    u16 = torch.empty(B * 64, 4 * 32, device="cuda", dtype=bf)
    ops.pixel_unshuffle(a32, u16, B, 8, 32, 32, 0)
    assert torch.equal(u16.float().cpu(), bfr(y4))
    src = dev(synth_array("es3", (6, 5, 8)))
    o16 = torch.empty(6, 8, 5, device="cuda", dtype=bf)
    ops.strided3(src, o16, (6, 8, 5), (40, 1, 8), (40, 5, 1))
    assert torch.equal(o16.float(), src.permute(0, 2, 1).to(bf).float())
    c16 = torch.empty(src.numel(), device="cuda", dtype=bf)
    ops.cast_bf16(src, c16, src.numel())
    assert torch.equal(c16, src.flatten().to(bf))


# ------------------------------------------------------------------------ data-prep STFT (row A1) + payload tiling (A14)
def test_dataprep_stft_tiles_match_reference_golden(wm, golden):
    """wm_stft256_tiles against the tiles the reference's own torch.stft(256/128) + pad + split produced (G2)."""
    from synth import chirp_wave
    ref = torch.from_numpy(golden("g2_dataprep")["tiles"])
    w = chirp_wave(48000, "g2_chirp")[None].cuda()
    tiles, n = wm.dataprep.stft_tiles(w)
    assert n == 3 and tiles.shape == (3, 2, 128, 128)
    assert_close(tiles, ref, 1e-5, "data-prep STFT tiles")
    # two utterances at once + audio_scale; a length whose frame count is a multiple of 128 gets one all-padding tile
    w2 = torch.stack([chirp_wave(16256, "dp_a"), chirp_wave(16256, "dp_b")]).cuda()
    t2, n2 = wm.dataprep.stft_tiles(w2, scale=0.5)
    assert n2 == 2 and float(t2[1].abs().max()) == 0.0 and float(t2[3].abs().max()) == 0.0
    st = torch.stft(w2.cpu(), n_fft=256, hop_length=128, win_length=256, window=torch.ones(256), return_complex=True)
    want = torch.view_as_real(st)[:, :-1].permute(0, 3, 1, 2) * 0.5
    assert_close(t2[0::2], want, 1e-5, "data-prep STFT, batch of utterances")


def test_payload_quadrant_tiling_roundtrip(wm):
    dp = wm.dataprep
    img = torch.rand(3, 1, 64, 64, device="cuda")
    q = dp.split_payload(img)
    assert torch.equal(q[:, 1, 0], img[:, 0, :32, 32:]) and torch.equal(q[:, 2, 0], img[:, 0, 32:, :32])
    assert torch.equal(dp.merge_payload(q), img)
    per_tile = dp.payload_for_tiles(img, 10)                         # 10 s clip = 10 tiles
    assert per_tile.shape == (30, 1, 32, 32) and torch.equal(per_tile[4], q[0, 0]) and torch.equal(per_tile[13], q[1, 3])
    assert torch.allclose(dp.payload_from_tiles(per_tile, 10), img)


# ------------------------------------------------------------------------------------------- deferred reductions
def test_sum_bf16_rows(wm):
    """fp32 accumulation of the ranks' bf16 gradient buckets (train.GradSync(exchange='bf16')): exact on bf16 inputs."""
    ops = wm.ops
    rows, n = 5, 8 * 1031
    src = torch.randn(rows, n, device="cuda").bfloat16()
    dst = torch.empty(n, device="cuda")
    ops.sum_bf16_rows(src, dst, rows, n)
    want = torch.zeros(n, device="cuda")
    for r in range(rows):
        want += src[r].float()
    assert torch.equal(dst, want)
    with pytest.raises(wm._lib.WmKernelError):
        ops.sum_bf16_rows(src, dst, rows, n - 3)


def test_deferred_reductions_match_immediate_ones(wm):
    """wm_reduce_jobs: LayerNorm / depthwise / bias-table / split-K producers defer their partial-sum reductions into one
    launch; gradients must equal the immediate (scratch + own reduce kernel) results."""
    ops = wm.ops
    B, hw, C = 2, 16, 64
    rows, Ch, heads = B * hw * hw, 4 * 64, 2
    bf = torch.bfloat16
    x = dev(synth_array("jx", (rows, C)))
    dyb = dev(synth_array("jdy", (rows, C))).to(bf)
    gam = dev(1 + 0.1 * synth_array("jg", (C,)))
    mean, rstd, xn = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda"), torch.empty(rows, C, device="cuda", dtype=bf)
    ops.layernorm_fwd(x, gam, torch.zeros_like(gam), None, xn, mean, rstd, hw, 0)
    h1 = dev(synth_array("jh1", (B, hw, hw, Ch))).to(bf)
    dh2 = dev(synth_array("jdh2", (B, hw, hw, Ch))).to(bf)
    wdw = dev(0.3 * synth_array("jw", (Ch, 1, 3, 3)))
    qkv = dev(synth_array("jqkv", (rows, 3 * C))).to(bf)
    table = dev(0.5 * synth_array("jtab", (225, heads)))
    o, lse = torch.empty(rows, C, device="cuda", dtype=bf), torch.empty(rows, heads, device="cuda")
    ops.winattn_fwd(qkv, table, o, B, hw, C, 4, 1, lse=lse)
    do = dev(synth_array("jdo", (rows, C))).to(bf)
    dyg, xg = dev(synth_array("jgy", (rows, 96))).to(bf), dev(synth_array("jgx", (rows, C))).to(bf)

    def run(deferred):
        jobs = ops.ReduceJobs() if deferred else None
        take = (lambda k: jobs.take(k)) if deferred else (lambda k: None)
        scr = [torch.empty(int(n), device="cuda") for n in (wm.lib.wm_layernorm_bwd_scratch_floats(C), wm.lib.wm_dwconv_bwd_scratch_floats(B, hw, Ch),
                                                            wm.lib.wm_winattn_bwd_scratch_floats(), wm.lib.wm_gemm_scratch_floats())]
        dx, dg, db = torch.empty(rows, C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        ops.layernorm_bwd(dyb, x, gam, mean, rstd, None, dx, dg, db, None, hw, 0, scratch=scr[0], defer=take(2))
        dh1, dw, dbw = torch.empty_like(h1), torch.zeros(Ch, 9, device="cuda"), torch.zeros(Ch, device="cuda")
        ops.dwconv_bwd(h1, dh2, wdw, dh1, dw, dbw, B, hw, Ch, scratch=scr[1], defer=take(2))
        dqkv, dt = torch.empty_like(qkv), torch.zeros_like(table)
        ops.winattn_bwd(qkv, table, do, dqkv, dt, B, hw, C, 4, 1, out=o, lse=lse, scratch=scr[2], defer=take(1))
        gw, gb = torch.zeros(96, C, device="cuda"), torch.zeros(96, device="cuda")
        ops.linear_bwd_weight(dyg, xg, gw, gb, 1, scratch=scr[3], defer=take(1))
        if deferred:
            assert jobs.n == 6
            ops.reduce_jobs(jobs)
            assert jobs.n == 0
        return dg, db, dw, dbw, dt, gw, gb

    for name, a, b in zip(("ln dgamma", "ln dbeta", "dw", "db", "bias table", "gemm dW", "gemm db"), run(False), run(True)):
        assert_close(b, a, 1e-5, f"deferred {name}")
