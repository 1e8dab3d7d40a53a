"""Tensor-level wrappers over the C ABI (include/wm_kernels.h).

Each function takes contiguous fp32 CUDA tensors, launches on torch's current stream and returns
nothing (outputs are passed in) — allocation policy belongs to the caller (the engine), so the same
calls are hipGraph-capturable.  Nothing here has a PyTorch fallback.
"""
import ctypes as C

import torch

from ._lib import lib, check, ptr, WmGemm, WmConv, WmAttack, WmReduceJob, WmWaveAttack, WmKernelError

MODE_FP32, MODE_BF16, MODE_SPLIT = 0, 1, 2      # SPLIT: fp32 storage, GEMMs as three bf16 MFMA products (hi*hi + hi*lo + lo*hi)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _bf(t):
    """1 if the tensor is stored as bf16 (wide activations / weight copies in bf16 mode), else 0 (fp32)."""
    if t is None:
        return 0
    if t.dtype == torch.bfloat16:
        return 1
    if t.dtype != torch.float32:
        raise ValueError(f"libwm_hip ops take float32 or bfloat16 tensors, got {t.dtype}")
    return 0


def gemm(A, B, Cout, M, N, K, lda, ldb, ldc, a_kc=1, b_kc=1, a_op=0, b_op=0, bias=None, res=None, ldr=0,
         rowscale=None, rows_per_scale=0, kscale=None, k_per_scale=0, gaux=None, ldg=0, b_ones=False, C2=None,
         atomic_out=False, ksplit=0, mode=MODE_FP32, gaux_raw=False, scratch=None, defer=None, ln=None,
         a_grid_n=0, a_jump_k=0, a_jump=0, b_seg_cols=0, b_seg_rows=(0, 0, 0, 0)):
    lnf = (None, None, None, None, None, None, 0, 0)
    if ln is not None:
        gamma, beta, mod, out, mean, rstd, hw, shift = ln
        if out.dtype != torch.bfloat16:
            raise ValueError("fused LayerNorm output is bf16 (bf16 mode only)")
        lnf = (ptr(gamma), ptr(beta), ptr(mod), ptr(out), ptr(mean), ptr(rstd), int(hw), int(shift))
    g = WmGemm(ptr(A), ptr(B), ptr(Cout), M, N, K, lda, ldb, ldc, a_kc, b_kc, a_op, b_op, ptr(bias), ptr(res), ldr,
               ptr(rowscale), rows_per_scale, ptr(kscale), k_per_scale, ptr(gaux), ldg, int(b_ones), ptr(C2),
               int(atomic_out), ksplit, mode, _bf(A), _bf(B), _bf(Cout), _bf(gaux), int(gaux_raw),
               ptr(scratch), scratch.numel() if scratch is not None else 0, defer, *lnf,
               int(a_grid_n), int(a_jump_k), int(a_jump), int(b_seg_cols), (C.c_int32 * 4)(*[int(v) for v in b_seg_rows]))
    check(lib.wm_gemm(C.byref(g), _stream()), f"wm_gemm(M={M},N={N},K={K})")


def ln_fusable(M, N, K, mode, a_bf16=True):
    """Can linear_fwd(..., ln=...) produce the LayerNorm of its output rows in the same launch?  (WM_NO_LN_FUSE=1: never; a comma
    list of widths: not at those — the A/B switch of the benchmarks.)"""
    import os
    off = os.environ.get("WM_NO_LN_FUSE", "")
    if off == "1" or (off and off != "0" and str(N) in off.split(",")):     # "1": never; "32,64": not at these widths
        return False
    return bool(lib.wm_gemm_ln_fusable(M, N, K, mode, int(a_bf16)))


def linear_fwd(x, w, bias, out, mode, a_op=0, res=None, rowscale=None, rows_per_scale=0, ln=None):
    """out[M,N] = (op(x)[M,K] @ w[N,K]^T + bias) * rowscale + res;  ln = (gamma, beta, modulator or None, y_bf16, mean, rstd, hw,
    shift): y = LayerNorm(out rows) * gamma + beta (+ modulator[window position]) written by the same launch."""
    M, K = x.shape
    N = w.shape[0]
    if ln is not None:
        # wm_gemm_ln_fusable answers for (M, N, K, mode) only; the launchers also want rows_per_scale / hw * hw to be multiples of the
        # tile height when a DropPath scale / the modulator ride along.  A launch they decline (WM_EINVAL: nothing was written) runs as
        # the plain product followed by the LayerNorm kernel: same result, one more launch
        try:
            gemm(x, w, out, M, N, K, K, K, N, 1, 1, a_op=a_op, bias=bias, res=res, ldr=N, rowscale=rowscale,
                 rows_per_scale=rows_per_scale, mode=mode, ln=ln)
            return
        except WmKernelError as e:
            if e.code != 22:
                raise
        gamma, beta, mod, y, mean, rstd, hw, shift = ln
        gemm(x, w, out, M, N, K, K, K, N, 1, 1, a_op=a_op, bias=bias, res=res, ldr=N, rowscale=rowscale,
             rows_per_scale=rows_per_scale, mode=mode)
        layernorm_fwd(out, gamma, beta, mod, y, mean, rstd, hw, shift)
        return
    gemm(x, w, out, M, N, K, K, K, N, 1, 1, a_op=a_op, bias=bias, res=res, ldr=N, rowscale=rowscale,
         rows_per_scale=rows_per_scale, mode=mode)


def linear_bwd_input(dy, w, dx, mode, rowscale=None, rows_per_scale=0, gaux=None, gaux_raw=False, wT=None):
    """dx[M,K] = ((dy[M,N] @ w[N,K]) * rowscale) * (gaux if gaux_raw else gelu'(gaux)).
    wT (optional): the transposed copy w^T [K,N] (wm_transpose_cast_jobs); the product then runs as a K-contiguous x
    K-contiguous GEMM like the forward, which is the form the 256 x 256 tile kernel takes."""
    M, N = dy.shape
    K = w.shape[1]
    if wT is not None:
        gemm(dy, wT, dx, M, K, N, N, N, K, 1, 1, rowscale=rowscale, rows_per_scale=rows_per_scale, gaux=gaux, ldg=K,
             mode=mode, gaux_raw=gaux_raw)
        return
    gemm(dy, w, dx, M, K, N, N, K, K, 1, 0, rowscale=rowscale, rows_per_scale=rows_per_scale, gaux=gaux, ldg=K,
         mode=mode, gaux_raw=gaux_raw)


def linear_bwd_weight(dy, x, dw, db, mode, x_op=0, kscale=None, k_per_scale=0, scratch=None, defer=None):
    """dw[N,K] += (dy*kscale)^T[N,M] @ op(x)[M,K];  db[N] += column sums of dy*kscale"""
    M, N = dy.shape
    K = x.shape[1]
    gemm(dy, x, dw, N, K, M, N, K, K, 0, 0, b_op=x_op, kscale=kscale, k_per_scale=k_per_scale,
         b_ones=db is not None, C2=db, atomic_out=True, mode=mode, scratch=scratch, defer=defer)


def layernorm_fwd(x, gamma, beta, mod, y, mean, rstd, hw, shift):
    rows, Cc = x.shape
    check(lib.wm_layernorm_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(mod), ptr(y), _bf(y), ptr(mean), ptr(rstd), rows, Cc,
                               hw, shift, _stream()), "wm_layernorm_fwd")


def layernorm_bwd(dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, dmod, hw, shift, scratch=None, defer=None, dx_bf16=None,
                  copy_scale=None, rows_per_scale=0):
    rows, Cc = x.shape
    if scratch is not None and scratch.numel() < lib.wm_layernorm_bwd_scratch_floats(Cc):
        raise ValueError("layernorm_bwd: scratch buffer too small")
    if dx_bf16 is not None and dx_bf16.dtype != torch.bfloat16:
        raise ValueError("layernorm_bwd: dx_bf16 must be a bfloat16 tensor")
    check(lib.wm_layernorm_bwd(ptr(dy), _bf(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(dx_bf16),
                               ptr(dgamma), ptr(dbeta), ptr(dmod), ptr(scratch), defer, rows, Cc, hw, shift, ptr(copy_scale),
                               rows_per_scale, _stream()),
          "wm_layernorm_bwd")


def winattn_fwd(qkv, table, out, B, hw, Cc, shift, mode, lse=None):
    if _bf(qkv) != _bf(out):
        raise ValueError("winattn: qkv and out must share a storage type")
    check(lib.wm_winattn_fwd(ptr(qkv), ptr(table), ptr(out), ptr(lse), _bf(qkv), B, hw, Cc, shift, mode, _stream()),
          "wm_winattn_fwd")


def winattn_bwd(qkv, table, dout, dqkv, dtable, B, hw, Cc, shift, mode, out=None, lse=None, scratch=None, defer=None):
    if scratch is not None and scratch.numel() < lib.wm_winattn_bwd_scratch_floats():
        raise ValueError("winattn_bwd: scratch buffer too small")
    if not (_bf(qkv) == _bf(dout) == _bf(dqkv)) or (out is not None and _bf(out) != _bf(qkv)):
        raise ValueError("winattn_bwd: qkv, out, dout and dqkv must share a storage type")
    check(lib.wm_winattn_bwd(ptr(qkv), ptr(table), ptr(out), ptr(lse), ptr(dout), ptr(dqkv), ptr(dtable), ptr(scratch),
                             defer, _bf(qkv), B, hw, Cc, shift, mode, _stream()), "wm_winattn_bwd")


def dwconv_fwd(h1, w, b, h2, B, hw, Ch, a2=None, g2=None):
    """h2 = dwconv3x3(gelu(h1)) + b (optional output), a2 = gelu(h2), g2 = gelu'(h2)"""
    for t in (h2, a2, g2):
        if t is not None and _bf(t) != _bf(h1):
            raise ValueError("dwconv: all tensors must share a storage type")
    check(lib.wm_dwconv_fwd(ptr(h1), ptr(w), ptr(b), ptr(h2), ptr(a2), ptr(g2), _bf(h1), B, hw, Ch, _stream()),
          "wm_dwconv_fwd")


def dwconv_bwd(h1, dh2, w, dh1, dw, db, B, hw, Ch, scratch=None, defer=None):
    if not (_bf(h1) == _bf(dh2) == _bf(dh1)):
        raise ValueError("dwconv_bwd: h1, dh2 and dh1 must share a storage type")
    if scratch is not None and scratch.numel() < lib.wm_dwconv_bwd_scratch_floats(B, hw, Ch):
        raise ValueError("dwconv_bwd: scratch buffer too small")
    check(lib.wm_dwconv_bwd(ptr(h1), ptr(dh2), ptr(w), ptr(dh1), ptr(dw), ptr(db), ptr(scratch), defer, _bf(h1), B, hw, Ch,
                            _stream()), "wm_dwconv_bwd")


def attn_half_fusable(hw, Cc):
    """Does the model route this block's attention half through the fused window-local kernel (csrc/attn_fused.hip)?
    (WM_NO_ATTN_FUSE=1: never — the A/B switch of the benchmarks.)"""
    import os
    if os.environ.get("WM_NO_ATTN_FUSE", "0") == "1":
        return False
    return bool(lib.wm_attn_half_fusable(hw, Cc))


def attn_half_fwd(x, g1, b1, mod, wqkv, bqkv, table, wproj, bproj, g2, b2, x1, xn2, mean1, rstd1, mean2, rstd2, lse, B, hw, Cc, shift,
                  rowscale=None, xn1=None, qkv=None, o=None):
    """x1 = x + rowscale * proj(window_attention(norm1(x) (+ mod))), xn2 = norm2(x1) in ONE launch (csrc/attn_fused.hip).
    xn1 / qkv / o (bf16, all or none): also written, for the un-fused backward launches."""
    for t in (wqkv, wproj, xn2, xn1, qkv, o):
        if t is not None and t.dtype != torch.bfloat16:
            raise ValueError("attn_half_fwd: wqkv, wproj, xn2, xn1, qkv, o are bfloat16 tensors (bf16 mode only)")
    check(lib.wm_attn_half_fwd(ptr(x), ptr(g1), ptr(b1), ptr(mod), ptr(wqkv), ptr(bqkv), ptr(table), ptr(wproj), ptr(bproj),
                               ptr(rowscale), ptr(g2), ptr(b2), ptr(x1), ptr(xn2), ptr(mean1), ptr(rstd1), ptr(mean2), ptr(rstd2),
                               ptr(lse), ptr(xn1), ptr(qkv), ptr(o), B, hw, Cc, shift, _stream()), "wm_attn_half_fwd")


def attn_half_bwd(dxn2, dy, x1, mean2, rstd2, g2, x, mean1, rstd1, g1, qkv, o, lse, table, wproj, wqkvT, dx1h, dqkv, dx, dg2, db2, dg1,
                  db1, dtable, B, hw, Cc, shift, scratch, rowscale=None, copy_scale=None, dxh=None, dmod=None, defer=None):
    """The residual-stream backward of the attention half in ONE launch (csrc/attn_fused.hip): norm2 backward -> projection dX ->
    window-attention backward -> q | k | v dX -> norm1 backward.  wqkvT: the transposed [C, 3C] bf16 copy of the q | k | v weight."""
    for t in (dxn2, qkv, o, wproj, wqkvT, dx1h, dqkv, dxh):
        if t is not None and t.dtype != torch.bfloat16:
            raise ValueError("attn_half_bwd: dxn2, qkv, o, wproj, wqkvT, dx1h, dqkv, dxh are bfloat16 tensors (bf16 mode only)")
    if scratch.numel() < lib.wm_attn_half_bwd_scratch_floats(B, hw, Cc, int(dmod is not None)):
        raise ValueError("attn_half_bwd: scratch buffer too small")
    check(lib.wm_attn_half_bwd(ptr(dxn2), ptr(dy), ptr(x1), ptr(mean2), ptr(rstd2), ptr(g2), ptr(rowscale), ptr(x), ptr(mean1),
                               ptr(rstd1), ptr(g1), ptr(qkv), ptr(o), ptr(lse), ptr(table), ptr(wproj), ptr(wqkvT), ptr(copy_scale),
                               ptr(dx1h), ptr(dqkv), ptr(dx), ptr(dxh), ptr(dg2), ptr(db2), ptr(dg1), ptr(db1), ptr(dmod), ptr(dtable),
                               ptr(scratch), defer, B, hw, Cc, shift, _stream()), "wm_attn_half_bwd")


def leff_fusable(hw, Cc):
    """Does the model route this block's LeFF through the fused kernels (csrc/leff_fused.hip)?  They are parity-green at every
    (hw, C <= 128) of the step but, measured on MI355X at B = 256 (scripts/leff_fused_bench.py, profiles/r04_leff_fused_bench_b256.txt),
    the forward kernel is 8 % ahead of fc1 + depthwise + fc2 and the backward kernel 35 % behind its three launches (C = 128: one
    workgroup per CU), 176.1 against 170.5 ms per step: the step keeps the separate launches unless WM_LEFF_FUSE=1."""
    import os
    if os.environ.get("WM_LEFF_FUSE", "0") != "1":
        return False
    return bool(lib.wm_leff_fusable(hw, Cc))


def leff_fwd(xn2, x1, w1, b1, wd, bd, w2, b2, x2, B, hw, Cc, rowscale=None, h1=None, h2=None):
    """x2 = (gelu(dwconv3x3(gelu(xn2 @ w1^T + b1)) + bd) @ w2^T + b2) * rowscale[sample] + x1 in one launch; h1 / h2 (bf16,
    both or neither): the two pre-activations, kept for leff_bwd."""
    for t in (xn2, w1, w2, h1, h2):
        if t is not None and t.dtype != torch.bfloat16:
            raise ValueError("leff_fwd: xn2, w1, w2, h1, h2 are bfloat16 tensors (bf16 mode only)")
    check(lib.wm_leff_fwd(ptr(xn2), ptr(x1), ptr(w1), ptr(b1), ptr(wd), ptr(bd), ptr(w2), ptr(b2), ptr(rowscale), ptr(x2),
                          ptr(h1), ptr(h2), B, hw, Cc, _stream()), "wm_leff_fwd")


def leff_bwd(dy, h1, h2, w1, wd, w2, dh1, dxn2, dwd, dbd, B, hw, Cc, rowscale=None, scratch=None, defer=None):
    """dh1 = gelu'(h1) * dwconv^T(((dy * rowscale) @ w2) * gelu'(h2)), dxn2 = dh1 @ w1; dwd / dbd accumulated."""
    for t in (h1, h2, w1, w2, dh1, dxn2):
        if t.dtype != torch.bfloat16:
            raise ValueError("leff_bwd: h1, h2, w1, w2, dh1, dxn2 are bfloat16 tensors (bf16 mode only)")
    if scratch is not None and scratch.numel() < lib.wm_leff_bwd_scratch_floats(B, hw, Cc):
        raise ValueError("leff_bwd: scratch buffer too small")
    check(lib.wm_leff_bwd(ptr(dy), _bf(dy), ptr(rowscale), ptr(h1), ptr(h2), ptr(w1), ptr(wd), ptr(w2), ptr(dh1), ptr(dxn2),
                          ptr(dwd), ptr(dbd), ptr(scratch), defer, B, hw, Cc, _stream()), "wm_leff_bwd")


def im2col(x, cols, B, hw, Cc):
    check(lib.wm_im2col_4x4s2(ptr(x), ptr(cols), _bf(cols), B, hw, Cc, _stream()), "wm_im2col_4x4s2")


def col2im(dcols, dx, B, hw, Cc):
    check(lib.wm_col2im_4x4s2(ptr(dcols), _bf(dcols), ptr(dx), B, hw, Cc, _stream()), "wm_col2im_4x4s2")


def pixel_shuffle(y4, bias, dst, B, hw, Co, ldd, c0):
    check(lib.wm_pixel_shuffle2(ptr(y4), _bf(y4), ptr(bias), ptr(dst), B, hw, Co, ldd, c0, _stream()), "wm_pixel_shuffle2")


def pixel_unshuffle(dsrc, dy4, B, hw, Co, lds, c0):
    check(lib.wm_pixel_unshuffle2(ptr(dsrc), ptr(dy4), _bf(dy4), B, hw, Co, lds, c0, _stream()), "wm_pixel_unshuffle2")


def copy_cols(src, lds, s0, dst, ldd, c0, rows, Cc, accumulate=False):
    check(lib.wm_copy_cols(ptr(src), lds, s0, ptr(dst), ldd, c0, rows, Cc, int(accumulate), _stream()), "wm_copy_cols")


def strided3(src, dst, n, si, so, atomic=False):
    check(lib.wm_strided3(ptr(src), ptr(dst), _bf(dst), n[0], n[1], n[2], si[0], si[1], si[2], so[0], so[1], so[2],
                          int(atomic), _stream()), "wm_strided3")


def cast_bf16(src, dst, n):
    check(lib.wm_cast_bf16(ptr(src), ptr(dst), n, _stream()), "wm_cast_bf16")


def sum_bf16_rows(src, dst, rows, n):
    """dst[i] (fp32) = sum over the `rows` bf16 rows of src[r, i]."""
    check(lib.wm_sum_bf16_rows(ptr(src), ptr(dst), rows, n, _stream()), "wm_sum_bf16_rows")


def down_direct_ok(hw, Cc):
    """Downsample without the im2col image (csrc/conv_grid.hip + grid addressing of wm_gemm): bf16 mode, C a multiple of 32."""
    import os
    return os.environ.get("WM_DOWN_DIRECT", "1") != "0" and hw % 2 == 0 and hw >= 4 and Cc % 32 == 0


def conv_weights(w, Cc):
    """weight [2C, C, 4, 4] fp32 -> (w_fwd [2C, 16C], w_dx [4C, 8C]) bf16 in 2 x 2-block order."""
    w_fwd = torch.empty(2 * Cc, 16 * Cc, device=w.device, dtype=torch.bfloat16)
    w_dx = torch.empty(4 * Cc, 8 * Cc, device=w.device, dtype=torch.bfloat16)
    check(lib.wm_conv4x4s2_weights(ptr(w), ptr(w_fwd), ptr(w_dx), Cc, _stream()), "wm_conv4x4s2_weights")
    return w_fwd, w_dx


def conv_x_to_grid(x, xg, B, hw, Cc):
    check(lib.wm_conv4x4s2_x_to_grid(ptr(x), ptr(xg), B, hw, Cc, _stream()), "wm_conv4x4s2_x_to_grid")


def conv_dy_to_grid(dy, dyg, B, hw, Cc):
    check(lib.wm_conv4x4s2_dy_to_grid(ptr(dy), _bf(dy), ptr(dyg), B, hw, Cc, _stream()), "wm_conv4x4s2_dy_to_grid")


def conv_grid_to_dx(dxg, dx, B, hw, Cc):
    check(lib.wm_conv4x4s2_grid_to_dx(ptr(dxg), ptr(dx), B, hw, Cc, _stream()), "wm_conv4x4s2_grid_to_dx")


def down_direct_fwd(x, w_fwd, bias, y, B, hw, Cc):
    """y [B (hw/2)^2, 2C] = Conv2d(4, 2, 1)(x [B, hw, hw, C]) + bias; returns the grid image of x (kept for the weight gradient)."""
    n, Wg = hw // 2, hw // 2 + 2
    xg = torch.empty(int(lib.wm_conv4x4s2_xg_rows(B, hw)), 4 * Cc, device=x.device, dtype=torch.bfloat16)
    conv_x_to_grid(x, xg, B, hw, Cc)
    gemm(xg, w_fwd, y, B * n * n, 2 * Cc, 16 * Cc, 4 * Cc, 16 * Cc, 2 * Cc, 1, 1, bias=bias, mode=MODE_BF16,
         a_grid_n=n, a_jump_k=8 * Cc, a_jump=(Wg - 2) * 4 * Cc)
    return xg


def down_direct_bwd(dy, xg, w_dx, dx, dw_fwd_order, db, B, hw, Cc, scratch=None):
    """dx [B, hw, hw, C] fp32 (written), dw_fwd_order [2C, 16C] fp32 += , db [2C] += from dy [B (hw/2)^2, 2C] (fp32 or bf16)."""
    n, Wg = hw // 2, hw // 2 + 2
    Mg = int(lib.wm_conv4x4s2_grid_rows(B, hw))
    dyg = torch.empty(int(lib.wm_conv4x4s2_dyg_rows(B, hw)), 2 * Cc, device=dy.device, dtype=torch.bfloat16)
    conv_dy_to_grid(dy, dyg, B, hw, Cc)
    dxg = torch.empty(Mg, 4 * Cc, device=dy.device, dtype=torch.bfloat16)
    # rows g - Wg - 1, g - Wg | g - 1, g of the bordered gradient (the buffer starts Wg + 1 rows before grid row 0)
    gemm(dyg, w_dx, dxg, Mg, 4 * Cc, 8 * Cc, 2 * Cc, 8 * Cc, 4 * Cc, 1, 1, mode=MODE_BF16, a_jump_k=4 * Cc, a_jump=(Wg - 2) * 2 * Cc)
    conv_grid_to_dx(dxg, dx, B, hw, Cc)
    dyd = dyg[Wg + 1:]                                            # the grid rows themselves: the token axis of the weight gradient
    gemm(dyd, xg, dw_fwd_order, 2 * Cc, 16 * Cc, Mg, 2 * Cc, 4 * Cc, 16 * Cc, 0, 0, b_ones=db is not None, C2=db, atomic_out=True,
         mode=MODE_BF16, scratch=scratch, b_seg_cols=4 * Cc, b_seg_rows=(0, 1, Wg, Wg + 1))


def conv_dweight(dw_fwd_order, grad, Cc):
    check(lib.wm_conv4x4s2_dweight(ptr(dw_fwd_order), ptr(grad), Cc, _stream()), "wm_conv4x4s2_dweight")


def transpose_cast_jobs(arena, arena_t, jobs, njobs, total_tiles):
    """bf16 transposes of the job list's matrices (same offsets in both arenas), one launch."""
    check(lib.wm_transpose_cast_jobs(ptr(arena), ptr(arena_t), ptr(jobs), njobs, total_tiles, _stream()), "wm_transpose_cast_jobs")


def minmax(x):
    """[min, max] of a float32 tensor as a 2-element device tensor (no host synchronisation)."""
    x = x.contiguous()
    out = torch.empty(2, device=x.device)
    scr = torch.empty(int(lib.wm_minmax_scratch_floats()), device=x.device)
    check(lib.wm_minmax(ptr(x), x.numel(), ptr(out), ptr(scr), _stream()), "wm_minmax")
    return out


def range_map(x, min_max, lo, hi, inverse=False):
    """(x - min) / (max - min) * (hi - lo) + lo, or its inverse, with device-resident statistics."""
    x = x.contiguous()
    out = torch.empty_like(x)
    check(lib.wm_range_map(ptr(x), ptr(out), ptr(min_max), float(lo), float(hi), int(inverse), x.numel(), _stream()), "wm_range_map")
    return out


def msg_expand_fwd(feat, dst, B):
    check(lib.wm_msg_expand_fwd(ptr(feat), ptr(dst), B, _stream()), "wm_msg_expand_fwd")


def msg_expand_bwd(dcat, dfeat, B):
    check(lib.wm_msg_expand_bwd(ptr(dcat), ptr(dfeat), B, _stream()), "wm_msg_expand_bwd")


ACT = {"none": 0, "relu": 1, "leaky": 2, "sigmoid": 3}


def conv_desc(B, Cin, H, W, Cout, k, stride=(1, 1), pad=(0, 0), in_nhwc=False, out_nhwc=False, act="none",
              transposed=False):
    kh, kw = k
    if transposed:
        OH, OW = H * stride[0], W * stride[1]
    else:
        OH = (H + 2 * pad[0] - kh) // stride[0] + 1
        OW = (W + 2 * pad[1] - kw) // stride[1] + 1
    return WmConv(B, Cin, H, W, Cout, kh, kw, stride[0], stride[1], pad[0], pad[1], OH, OW, int(in_nhwc),
                  int(out_nhwc), ACT[act], int(transposed))


def conv_fwd(d, x, w, b, y):
    check(lib.wm_conv_fwd(C.byref(d), ptr(x), ptr(w), ptr(b), ptr(y), _stream()), "wm_conv_fwd")


def conv_bwd(d, x, w, y, dy, dx, dw, db, dx_accumulate=False):
    check(lib.wm_conv_bwd(C.byref(d), ptr(x), ptr(w), ptr(y), ptr(dy), ptr(dx), int(dx_accumulate), ptr(dw), ptr(db),
                          _stream()), "wm_conv_bwd")


def maxpool_fwd(x, y, planes, H, W, kh, kw):
    check(lib.wm_maxpool_fwd(ptr(x), ptr(y), planes, H, W, kh, kw, _stream()), "wm_maxpool_fwd")


def maxpool_bwd(x, dy, dx, planes, H, W, kh, kw, accumulate=False):
    check(lib.wm_maxpool_bwd(ptr(x), ptr(dy), ptr(dx), planes, H, W, kh, kw, int(accumulate), _stream()),
          "wm_maxpool_bwd")


def stft256_tiles(wave, scale=1.0):
    """Data-prep STFT of U equal-length utterances [U, L] -> [U * tiles_per_utterance, 2, 128, 128] tiles."""
    if wave.dim() != 2 or wave.dtype != torch.float32 or not wave.is_contiguous():
        raise ValueError("stft256_tiles takes a contiguous float32 [U, L] waveform batch")
    U, L = wave.shape
    tpu = lib.wm_stft256_tiles_per_utterance(L)
    tiles = torch.empty(U * tpu, 2, 128, 128, device=wave.device)
    check(lib.wm_stft256_tiles(ptr(wave), L, U, ptr(tiles), float(scale), _stream()), "wm_stft256_tiles")
    return tiles


def stft255_tiles(wave, scale=1.0):
    """torch.stft(n_fft=255) of U equal-length utterances [U, L] cut into tiles (SpeechDataTest.prepare_data):
    -> ([U * tiles_per_utterance, 2, 128, 128], frames per utterance)."""
    if wave.dim() != 2 or wave.dtype != torch.float32 or not wave.is_contiguous():
        raise ValueError("stft255_tiles takes a contiguous float32 [U, L] waveform batch")
    U, L = wave.shape
    tpu = lib.wm_stft255_tiles_per_utterance(L)
    tiles = torch.empty(U * tpu, 2, 128, 128, device=wave.device)
    check(lib.wm_stft255_tiles(ptr(wave), L, U, ptr(tiles), float(scale), _stream()), "wm_stft255_tiles")
    return tiles, lib.wm_stft255_frames(L)


def istft255_stitch(tiles, n_tiles, last_len, L, out_scale=1.0):
    """torch.istft(n_fft=255, length=L) of each utterance's n_tiles tiles joined along the frame axis, the last tile
    trimmed to last_len frames (reconstruct_audio): [U * n_tiles, 2, 128, 128] -> [U, L]."""
    if tiles.dim() != 4 or tuple(tiles.shape[1:]) != (2, 128, 128) or tiles.dtype != torch.float32 or \
            not tiles.is_contiguous() or tiles.shape[0] % n_tiles:
        raise ValueError("istft255_stitch takes contiguous float32 [U * n_tiles, 2, 128, 128] tiles")
    U = tiles.shape[0] // n_tiles
    wave = torch.empty(U, L, device=tiles.device)
    check(lib.wm_istft255_stitch(ptr(tiles), n_tiles, last_len, U, ptr(wave), L, float(out_scale), _stream()),
          "wm_istft255_stitch")
    return wave


def wave_attack(wave, snr_db=None, amp=None, echo=None, jitter=None, requantize=False, seed=0):
    """Whole-utterance state-free attacks on [U, L] waveforms: AWGN (snr_db), amplitude scaling (amp), echo addition
    (echo = (delay_samples, gain)), jittering_2 (jitter = number of zeroed samples), 8-bit requantisation."""
    if wave.dim() != 2 or wave.dtype != torch.float32 or not wave.is_contiguous():
        raise ValueError("wave_attack takes a contiguous float32 [U, L] waveform batch")
    U, L = wave.shape
    out = torch.empty_like(wave)
    power = torch.empty(U, device=wave.device)
    flags = (2 if snr_db is not None else 0) | (8 if amp is not None else 0) | (16 if echo is not None else 0) | \
            (32 if jitter is not None else 0) | (64 if requantize else 0)
    delay, gain = echo if echo is not None else (0, 0.0)
    a = WmWaveAttack(flags, float(snr_db or 0.0), float(1.0 if amp is None else amp), float(gain), int(delay),
                     int(jitter or 0), 0, int(seed))
    check(lib.wm_wave_attack(C.byref(a), ptr(wave), ptr(out), L, U, ptr(power), _stream()), "wm_wave_attack")
    return out


_stft_tabs = {}


def stft_tables(device, terms=2):
    """The folded cos / sin DFT matrices (`terms` bf16 planes each, MFMA operand order) of the n_fft = 255 transforms: one
    128 KB (terms = 2) / 192 KB (terms = 3) device buffer per GPU and precision, filled once by wm_stft255_tables (outside any
    graph capture: the first, eager, call makes it)."""
    idx = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    key = (idx, int(terms))
    t = _stft_tabs.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("stft_tables: first use inside a graph capture; run one eager step (or ops.stft_tables) first")
        nbytes = int(lib.wm_stft255_table_bytes(int(terms)))
        if nbytes <= 0:
            raise ValueError("stft_tables: terms is 2 (bf16 compute mode) or 3 (fp32 / split modes)")
        t = torch.empty(nbytes // 2, device=torch.device("cuda", idx), dtype=torch.bfloat16)
        check(lib.wm_stft255_tables(ptr(t), int(terms), _stream()), "wm_stft255_tables")
        torch.cuda.current_stream().synchronize()              # other streams may use it next
        _stft_tabs[key] = t
    return t


def istft_fwd(spec, wave, B, terms=2):
    check(lib.wm_istft255_fwd(ptr(spec), ptr(wave), ptr(stft_tables(wave.device, terms)), terms, B, _stream()), "wm_istft255_fwd")


def stft_fwd(wave, spec, B, terms=2):
    check(lib.wm_stft255_fwd(ptr(wave), ptr(spec), ptr(stft_tables(spec.device, terms)), terms, B, _stream()), "wm_stft255_fwd")


def istft_bwd(dwave, dspec, B, terms=2):
    check(lib.wm_istft255_bwd(ptr(dwave), ptr(dspec), ptr(stft_tables(dspec.device, terms)), terms, B, _stream()), "wm_istft255_bwd")


def stft_bwd(dspec, dwave, B, terms=2):
    check(lib.wm_stft255_bwd(ptr(dspec), ptr(dwave), ptr(stft_tables(dwave.device, terms)), terms, B, _stream()), "wm_stft255_bwd")


def attack_fwd(desc, wave, out, sigma, B, L):
    check(lib.wm_attack_fwd(C.byref(desc), ptr(wave), ptr(out), ptr(sigma), B, L, _stream()), "wm_attack_fwd")


def attack_bwd(desc, wave, dout, sigma, dwave, B, L):
    check(lib.wm_attack_bwd(C.byref(desc), ptr(wave), ptr(dout), ptr(sigma), ptr(dwave), B, L, _stream()),
          "wm_attack_bwd")


def awgn_noise(seed, step, B, L):
    """[B, L] standard-normal stream of the in-loop AWGN stage for (seed, step tensor)."""
    z = torch.empty(B, L, device=step.device, dtype=torch.float32)
    check(lib.wm_awgn_noise(int(seed), ptr(step), ptr(z), B, L, _stream()), "wm_awgn_noise")
    return z


def loss_fwd_bwd(stft_new, x, noise, wm_pred, wm, msg, losses, d_stft_new, d_noise, d_wm_pred, d_wm, B, scratch):
    check(lib.wm_loss_fwd_bwd(ptr(stft_new), ptr(x), ptr(noise), ptr(wm_pred), ptr(wm), ptr(msg), ptr(losses),
                              ptr(d_stft_new), ptr(d_noise), ptr(d_wm_pred), ptr(d_wm), B, ptr(scratch), _stream()),
          "wm_loss_fwd_bwd")


def adamw(p, g, m, v, n, lr_t, b1, b2, eps, wd, step_t, gscale_t=None):
    check(lib.wm_adamw(ptr(p), ptr(g), ptr(m), ptr(v), n, ptr(lr_t), b1, b2, eps, wd, ptr(step_t), ptr(gscale_t),
                       _stream()), "wm_adamw")


def step_inc(step_t):
    check(lib.wm_step_inc(ptr(step_t), _stream()), "wm_step_inc")


def step_copy(src, dst):
    check(lib.wm_step_copy(ptr(src), ptr(dst), _stream()), "wm_step_copy")


def droppath_scales(scales, rates, nslots, B, seed, step_t):
    check(lib.wm_droppath_scales(ptr(scales), ptr(rates), nslots, B, seed, ptr(step_t), _stream()),
          "wm_droppath_scales")


def axpbyz(out, a, x, b, y, n):
    check(lib.wm_axpbyz(ptr(out), a, ptr(x), b, ptr(y), n, _stream()), "wm_axpbyz")


# ---- small CNN models (ModelA, HiDDeN decoder) ------------------------------------------------------------------
def batchnorm2d_fwd(x, gamma, beta, rmean, rvar, y, save_mean, save_invstd, eps, momentum, training, act="none", slope=0.0):
    B, Cc, H, W = x.shape
    check(lib.wm_batchnorm2d_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar), ptr(y), ptr(save_mean),
                                 ptr(save_invstd), B, Cc, H * W, eps, momentum, int(training), ACT[act], slope, _stream()),
          "wm_batchnorm2d_fwd")


def batchnorm2d_bwd(dy, x, y, gamma, save_mean, save_invstd, dx, dgamma, dbeta, sums, training, act="none", slope=0.0):
    B, Cc, H, W = x.shape
    check(lib.wm_batchnorm2d_bwd(ptr(dy), ptr(x), ptr(y), ptr(gamma), ptr(save_mean), ptr(save_invstd), ptr(dx), ptr(dgamma),
                                 ptr(dbeta), ptr(sums), B, Cc, H * W, int(training), ACT[act], slope, _stream()),
          "wm_batchnorm2d_bwd")


def dropout(x, y, p, seed, offset):
    check(lib.wm_dropout(ptr(x), ptr(y), x.numel(), p, seed, offset, _stream()), "wm_dropout")


def mse_fwd_bwd(a, b, loss, da=None, db=None, gscale=1.0):
    check(lib.wm_mse_fwd_bwd(ptr(a), ptr(b), a.numel(), ptr(loss), ptr(da), ptr(db), gscale, _stream()), "wm_mse_fwd_bwd")


def plane_fill(v, dst, B, L, Ctot, c0, HW, scale=1.0):
    """dst[b, c0 + l, :] = scale * v[b, l]"""
    check(lib.wm_plane_fill(ptr(v), ptr(dst), B, L, Ctot, c0, HW, float(scale), _stream()), "wm_plane_fill")


def plane_sum(src, v, B, L, Ctot, c0, HW, scale=1.0):
    """v[b, l] = scale * sum(src[b, c0 + l, :])"""
    check(lib.wm_plane_sum(ptr(src), ptr(v), B, L, Ctot, c0, HW, float(scale), _stream()), "wm_plane_sum")


def bce_logits_fwd_bwd(logits, target, loss, dlogits=None, gscale=1.0):
    check(lib.wm_bce_logits_fwd_bwd(ptr(logits), logits.numel(), float(target), ptr(loss), ptr(dlogits), float(gscale), _stream()),
          "wm_bce_logits_fwd_bwd")


def linear_small_fwd(x, w, b, y):
    check(lib.wm_linear_small_fwd(ptr(x), ptr(w), ptr(b), ptr(y), x.shape[0], w.shape[0], w.shape[1], _stream()), "wm_linear_small_fwd")


def linear_small_bwd(dy, x, w, dx, dw, db):
    check(lib.wm_linear_small_bwd(ptr(dy), ptr(x), ptr(w), ptr(dx), ptr(dw), ptr(db), x.shape[0], w.shape[0], w.shape[1], _stream()),
          "wm_linear_small_bwd")


def charbonnier_fwd_bwd(x, y, eps, loss, dx=None, dy=None, gscale=1.0):
    check(lib.wm_charbonnier_fwd_bwd(ptr(x), ptr(y), x.numel(), float(eps), ptr(loss), ptr(dx), ptr(dy), float(gscale), _stream()),
          "wm_charbonnier_fwd_bwd")


def tvloss_fwd_bwd(x, weight, loss, dx=None, gscale=1.0):
    N, Cc, H, W = x.shape
    check(lib.wm_tvloss_fwd_bwd(ptr(x), N, Cc, H, W, float(weight), ptr(loss), ptr(dx), float(gscale), _stream()), "wm_tvloss_fwd_bwd")


def tv_pow_fwd_bwd(x, beta, reg_coeff, loss, dx=None, gscale=1.0):
    N, Cc, H, W = x.shape
    check(lib.wm_tv_pow_fwd_bwd(ptr(x), N, Cc, H, W, float(beta), float(reg_coeff), ptr(loss), ptr(dx), float(gscale), _stream()),
          "wm_tv_pow_fwd_bwd")


def zero_(t):
    """t[...] = 0 with a libwm_hip kernel (a kernel node in captured steps; no ATen fill on the hot path)."""
    check(lib.wm_zero(ptr(t), t.numel(), _stream()), "wm_zero")
    return t


class ReduceJobs:
    """Host-side list of deferred partial-sum reductions (WmReduceJob x 16) that one wm_reduce_jobs launch performs.
    ``take(k)`` hands the next k entries to a producer's ``defer`` argument; ``run()`` launches and empties the list."""

    CAP = 16

    def __init__(self):
        self.arr = (WmReduceJob * self.CAP)()
        self.n = 0

    def take(self, k):
        if self.n + k > self.CAP:
            raise ValueError("ReduceJobs is full")
        p = C.cast(C.byref(self.arr, self.n * C.sizeof(WmReduceJob)), C.POINTER(WmReduceJob))
        self.n += k
        return p

    def run(self):
        if self.n:
            check(lib.wm_reduce_jobs(self.arr, self.n, _stream()), "wm_reduce_jobs")
        self.n = 0


def reduce_jobs(jobs):
    jobs.run()
