"""Parity of the kernel INSTANCES the benchmark runs (bf16 mode, B = 96): every tile choice of dispatch_nt / dispatch_tn in
csrc/gemm_fast.hip at bench-like shapes, each epilogue specialisation, the large-K split-K weight gradients, and the
strip / persistent-grid kernels at B > 2.

Reference: torch (rocBLAS / ATen on the GPU, float64) on the SAME bf16-rounded operands, i.e. exact products of what
the kernel reads, fp32-accumulation-free; so the only differences are fp32 accumulation order and the rounding of a bf16
output: tolerances 2e-4 for fp32 outputs and 4e-3 (half a bf16 ulp at full scale) for bf16 outputs, of max |ref|.
Replaces: nn.Linear forward / backward of LinearProjection (model.py:460-471), WindowAttention.proj (:549), LeFF (:700,711).
"""
import importlib
import math
import zlib

import pytest
import torch
import torch.nn.functional as F

from synth import synth_array

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
TOL_F32, TOL_BF16 = 2e-4, 4e-3


@pytest.fixture(scope="module")
def wm():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return importlib.import_module("image-in-speech-watermarking_amd")


def rnd(tag, shape, scale=1.0, dtype=BF):
    """Deterministic device tensor; the bf16 tensor IS the operand, its float64 upcast the reference operand."""
    g = torch.Generator(device="cuda").manual_seed(zlib.crc32(tag.encode()))
    return (torch.randn(shape, device="cuda", generator=g) * scale).to(dtype)


def rel(a, b):
    return ((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def check(a, b, tol, what):
    e = rel(a, b)
    assert e <= tol, f"{what}: max rel err {e:.3e} > {tol:.1e}"


# (M, N, K, note): forward GEMM shapes of the B = 96 step and the tile each one selects
NT_SHAPES = [
    (49152, 128, 32, "fc1 C=32 rows: 128x128 tiles"),
    (98304 + 40, 96, 32, "qkv C=32: 128x128 with a column-edge tile, ragged M"),
    (16384, 192, 64, "qkv C=64: three 64-wide tiles (N == 192 path)"),
    (49152, 32, 128, "fc2 C=32: 256x32 tiles"),
    (32768, 64, 64, "proj C=64: 128x64 tiles"),
    (6144, 512, 512, "proj C=512: 128x128 at K=512"),
    (6144 + 8, 2048, 512, "fc1 C=512: 128x128, 16 column tiles, ragged M"),
    (1536, 256, 1024, "coarse stage: 64x64 tiles"),
    (1000, 72, 200, "64x64 tiles, 128-deep k tiles: one full tile + a 72-wide tail, ragged M, column-edge tile"),
    (520, 64, 72, "64x64 tiles: K below one 128-deep tile"),
    (2048, 136, 328, "64x64 tiles: two full k tiles + a 72-wide tail, three column tiles"),
]


@pytest.mark.parametrize("M,N,K,note", NT_SHAPES)
def test_nt_forward_instances(wm, M, N, K, note):
    """C = epi(A W^T): EPI 0 (bf16 out, bias), EPI 1 (fp32 out, bias, DropPath row scale, residual), GELU-on-load A."""
    ops = wm.ops
    a, w = rnd("fa" + note, (M, K)), rnd("fw" + note, (N, K), 1.0 / math.sqrt(K))
    bias = rnd("fb" + note, (N,), 1.0, torch.float32)
    ref = a.double() @ w.double().t() + bias.double()
    out = torch.empty(M, N, device="cuda", dtype=BF)
    ops.linear_fwd(a, w, bias, out, 1)
    check(out, ref, TOL_BF16, f"EPI0 {note}")
    rows_per = 128 if M % 128 == 0 else 8
    nsc = (M + rows_per - 1) // rows_per
    rs = (torch.rand(nsc, device="cuda") + 0.5)
    res = rnd("fr" + note, (M, N), 1.0, torch.float32)
    out32 = torch.empty(M, N, device="cuda")
    ops.linear_fwd(a, w, bias, out32, 1, res=res, rowscale=rs, rows_per_scale=rows_per)
    ref1 = ref * rs.double().repeat_interleave(rows_per)[:M, None] + res.double()
    check(out32, ref1, TOL_F32, f"EPI1 {note}")
    if K <= 128:                                                  # fc2 reads gelu(h2) in the fp32 fallback only; cheap shapes
        ops.linear_fwd(a, w, bias, out, 1, a_op=1)
        check(out, F.gelu(a.float()).to(BF).double() @ w.double().t() + bias.double(), TOL_BF16, f"gelu(A) {note}")
    # fp32 activations as the A operand (first block of a level: the residual stream itself)
    a32 = a.float() + 1e-3
    ops.linear_fwd(a32, w, bias, out, 1)
    check(out, a32.to(BF).double() @ w.double().t() + bias.double(), TOL_BF16, f"fp32 A {note}")


@pytest.mark.parametrize("M,N,K,note", NT_SHAPES)
def test_nt_input_gradient_instances(wm, M, N, K, note):
    """dX[M,K] = epi(dY[M,N] W[N,K]) (B row-contiguous, ds_read_b64_tr_b16): EPI 0 plain and with the DropPath scale,
    EPI 2 (times the stored gelu' factor)."""
    ops = wm.ops
    if K % 8:
        pytest.skip("dX needs K % 8 == 0")
    dy, w = rnd("bdy" + note, (M, N)), rnd("bw" + note, (N, K), 1.0 / math.sqrt(N))
    ref = dy.double() @ w.double()
    dx = torch.empty(M, K, device="cuda", dtype=BF)
    ops.linear_bwd_input(dy, w, dx, 1)
    check(dx, ref, TOL_BF16, f"dX {note}")
    rows_per = 128 if M % 128 == 0 else 8
    rs = torch.rand((M + rows_per - 1) // rows_per, device="cuda") + 0.5
    sc = rs.double().repeat_interleave(rows_per)[:M, None]
    ops.linear_bwd_input(dy, w, dx, 1, rowscale=rs, rows_per_scale=rows_per)
    check(dx, ref * sc, TOL_BF16, f"dX rowscale {note}")
    gaux = rnd("bg" + note, (M, K))
    ops.linear_bwd_input(dy, w, dx, 1, rowscale=rs, rows_per_scale=rows_per, gaux=gaux, gaux_raw=True)
    check(dx, ref * sc * gaux.double(), TOL_BF16, f"dX EPI2 {note}")
    # fp32 incoming gradient (first block of a level)
    ops.linear_bwd_input(dy.float(), w, dx, 1)
    check(dx, ref, TOL_BF16, f"dX fp32 dY {note}")


TN_SHAPES = [
    # tokens, N_out, K_in
    (393216, 96, 32, "qkv C=32 at 393k tokens"),
    (393216 + 72, 128, 32, "fc1 C=32, ragged token count"),
    (196608, 32, 128, "fc2 C=32"),
    (98304, 192, 64, "qkv C=64"),
    (24576, 512, 128, "fc1 C=128"),
    (6144, 1536, 512, "qkv C=512 (dy row stride 3072 B)"),
    (6144, 2048, 512, "fc1 C=512"),
    (6144, 512, 2048, "fc2 C=512"),
]


@pytest.mark.parametrize("T,N,K,note", TN_SHAPES)
def test_tn_weight_gradient_instances(wm, T, N, K, note):
    """dW[N,K] += dY^T X, db += column sums: atomics, scratch split-K + its reduce kernel, the deferred reduction and the
    DropPath k-scale, at bench-like token counts (K of the GEMM = tokens)."""
    ops = wm.ops
    dy, x = rnd("tdy" + note, (T, N), 0.05), rnd("tx" + note, (T, K))
    ref, refb = dy.double().t() @ x.double(), dy.double().sum(0)
    scr = torch.empty(int(wm.lib.wm_gemm_scratch_floats()), device="cuda")
    tol = 3e-4                                                     # fp32 accumulation over up to 393 k terms, split in slices
    for name, kw in (("atomics", {}), ("scratch", dict(scratch=scr))):
        dw, db = torch.ones(N, K, device="cuda"), torch.zeros(N, device="cuda")
        ops.linear_bwd_weight(dy, x, dw, db, 1, **kw)
        check(dw - 1.0, ref, tol, f"dW {name} {note}")
        check(db, refb, tol, f"db {name} {note}")
    jobs = ops.ReduceJobs()
    dw, db = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
    ops.linear_bwd_weight(dy, x, dw, db, 1, scratch=scr, defer=jobs.take(1))
    ops.reduce_jobs(jobs)
    check(dw, ref, tol, f"dW deferred {note}")
    if T % 64 == 0:
        kps = 16384 if T % 16384 == 0 else 64
        ks = torch.rand(T // kps, device="cuda") + 0.5
        dw, db = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
        ops.linear_bwd_weight(dy, x, dw, db, 1, kscale=ks, k_per_scale=kps, scratch=scr)
        # the kernel scales dY while staging it and re-rounds the product to bf16: the reference operand does the same
        dys = (dy.float() * ks.repeat_interleave(kps)[:, None]).to(BF).double()
        check(dw, dys.t() @ x.double(), tol, f"dW kscale {note}")
        check(db, dys.sum(0), tol, f"db kscale {note}")
    # gelu on the activation operand (the fp32-mode layout of fc2's weight gradient) and fp32 operands
    if K <= 128:
        dw, db = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
        ops.linear_bwd_weight(dy.float(), x.float(), dw, db, 1, x_op=1, scratch=scr)
        check(dw, dy.double().t() @ F.gelu(x.float()).to(BF).double(), 3e-3, f"dW gelu(X) fp32 operands {note}")


@pytest.mark.parametrize("B,hw,Ch", [(6, 64, 128), (3, 128, 128), (12, 32, 512)])
def test_dwconv_strip_kernels_many_tiles(wm, B, hw, Ch):
    """LeFF depthwise 3x3 (model.py:703-709) in bf16 storage at B > 2: the strip-walking forward (hw >= 64), the halo
    handling across many tiles per row, and the backward with its scratch-reduced weight gradient."""
    ops = wm.ops
    h1 = rnd(f"sh{B}{hw}", (B, hw, hw, Ch))
    w = rnd(f"sw{B}{hw}", (Ch, 1, 3, 3), 0.3, torch.float32)
    b = rnd(f"sb{B}{hw}", (Ch,), 0.1, torch.float32)
    h1d = h1.double().requires_grad_()
    ref = F.conv2d(F.gelu(h1d).permute(0, 3, 1, 2), w.double(), b.double(), padding=1, groups=Ch).permute(0, 2, 3, 1)
    h2, a2, g2 = (torch.empty(B, hw, hw, Ch, device="cuda", dtype=BF) for _ in range(3))
    ops.dwconv_fwd(h1, w, b, h2, B, hw, Ch, a2=a2, g2=g2)
    check(h2, ref.detach(), TOL_BF16, "dwconv h2")
    check(a2, F.gelu(ref.detach()), TOL_BF16, "dwconv gelu(h2)")
    gp = torch.autograd.functional.jvp(F.gelu, ref.detach(), torch.ones_like(ref.detach()))[1]
    check(g2, gp, TOL_BF16 * 2, "dwconv gelu'(h2)")
    dh2 = rnd(f"sg{B}{hw}", (B, hw, hw, Ch))
    ref.backward(dh2.double())
    dh1 = torch.empty_like(h1)
    dw, db = torch.zeros(Ch, 9, device="cuda"), torch.zeros(Ch, device="cuda")
    scr = torch.empty(int(wm.lib.wm_dwconv_bwd_scratch_floats(B, hw, Ch)), device="cuda")
    ops.dwconv_bwd(h1, dh2, w, dh1, dw, db, B, hw, Ch, scratch=scr)
    check(dh1, h1d.grad, TOL_BF16, "dwconv dh1")
    # weight gradient: correlate gelu(h1) with dh2 in float64
    xg = F.gelu(h1.double()).permute(0, 3, 1, 2)
    wd = w.double().requires_grad_()
    F.conv2d(xg, wd, None, padding=1, groups=Ch).backward(dh2.double().permute(0, 3, 1, 2))
    check(dw, wd.grad.reshape(Ch, 9), 5e-4, "dwconv dw")
    check(db, dh2.double().sum((0, 1, 2)), 5e-4, "dwconv db")


@pytest.mark.parametrize("B,hw,C,shift", [(8, 64, 64, 4), (24, 16, 256, 0), (3, 128, 32, 4)])
def test_window_attention_persistent_grid_many_windows(wm, B, hw, C, shift):
    """WindowAttention core (model.py:523-551) in bf16 storage with more windows than the persistent grid holds at once."""
    from test_kernels_gpu import ref_winattn
    ops = wm.ops
    heads, rows = C // 32, B * hw * hw
    qkv = rnd(f"pq{B}{hw}", (rows, 3 * C))
    table = rnd(f"pt{B}{hw}", (225, heads), 0.5, torch.float32)
    q64 = qkv.double().cpu().requires_grad_()
    t64 = table.double().cpu().requires_grad_()
    ref = ref_winattn(q64, t64, B, hw, C, shift)
    do = rnd(f"pd{B}{hw}", (rows, C))
    ref.backward(do.double().cpu())
    out, lse = torch.empty(rows, C, device="cuda", dtype=BF), torch.empty(rows, heads, device="cuda")
    ops.winattn_fwd(qkv, table, out, B, hw, C, shift, 1, lse=lse)
    check(out.cpu(), ref.detach(), 1e-2, "attention forward (bf16 probabilities)")
    dqkv, dtab = torch.empty_like(qkv), torch.zeros_like(table)
    scr = torch.empty(int(wm.lib.wm_winattn_bwd_scratch_floats()), device="cuda")
    ops.winattn_bwd(qkv, table, do, dqkv, dtab, B, hw, C, shift, 1, out=out, lse=lse, scratch=scr)
    check(dqkv.cpu(), q64.grad, 2e-2, "attention dqkv")
    check(dtab.cpu(), t64.grad, 2e-2, "attention dtable")


@pytest.mark.parametrize("rows,C", [(393216, 32), (98304, 64), (6144, 512)])
def test_layernorm_bench_rows(wm, rows, C):
    """LayerNorm (model.py:982,1017) forward with bf16 output and backward with the bf16 copy of dx at bench row counts."""
    ops = wm.ops
    hw = int(round(math.sqrt(rows / 24)))
    x = rnd(f"lx{rows}", (rows, C), 1.0, torch.float32)
    gamma = 1 + 0.1 * rnd(f"lg{rows}", (C,), 1.0, torch.float32)
    beta = 0.1 * rnd(f"lb{rows}", (C,), 1.0, torch.float32)
    xd = x.double().requires_grad_()
    ref = F.layer_norm(xd, (C,), gamma.double(), beta.double())
    y = torch.empty(rows, C, device="cuda", dtype=BF)
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    ops.layernorm_fwd(x, gamma, beta, None, y, mean, rstd, hw, 0)
    check(y, ref.detach(), TOL_BF16, "ln fwd")
    dy, dres = rnd(f"ld{rows}", (rows, C)), rnd(f"lr{rows}", (rows, C), 1.0, torch.float32)
    ref.backward(dy.double())
    dx, dxh = torch.empty(rows, C, device="cuda"), torch.empty(rows, C, device="cuda", dtype=BF)
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    scr = torch.empty(int(wm.lib.wm_layernorm_bwd_scratch_floats(C)), device="cuda")
    ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres, dx, dg, db, None, hw, 0, scratch=scr, dx_bf16=dxh)
    check(dx, xd.grad + dres.double(), 1e-4, "ln dx")
    assert torch.equal(dxh, dx.to(BF))
    gfull = torch.autograd.grad(F.layer_norm(x.double(), (C,), g2 := gamma.double().requires_grad_(), b2 := beta.double().requires_grad_()),
                                [g2, b2], dy.double())
    check(dg, gfull[0], 3e-4, "ln dgamma")
    check(db, gfull[1], 3e-4, "ln dbeta")
