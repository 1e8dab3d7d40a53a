"""Parity of the 256 x 256 tile GEMM (csrc/gemm_big.hip) at the shapes the B = 256 bench step sends it, through the C ABI
(wm_gemm): the three epilogues, the 32-row-granular M edge, the dX form on transposed weight copies
(wm_transpose_cast_jobs) and the dispatch guard (small launches stay on the 128 x 128 kernels).

Reference: float64 products of the SAME bf16-rounded operands; 2e-4 (fp32 outputs) / 4e-3 (bf16 outputs) of max |ref|.
Replaces nn.Linear forward / input gradient of LinearProjection, WindowAttention.proj and LeFF at C >= 256
(reference model.py:460-471, 549, 700, 711).
"""
import importlib
import math
import os
import zlib

import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
TOL_F32, TOL_BF16 = 2e-4, 4e-3


@pytest.fixture(scope="module")
def wm():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return importlib.import_module("image-in-speech-watermarking_amd")


def rnd(tag, shape, scale=1.0, dtype=BF):
    g = torch.Generator(device="cuda").manual_seed(zlib.crc32(tag.encode()))
    return (torch.randn(shape, device="cuda", generator=g) * scale).to(dtype)


def rel(a, b):
    return ((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def check(a, b, tol, what):
    e = rel(a, b)
    assert e <= tol, f"{what}: max rel err {e:.3e} > {tol:.1e}"


def both(fn):
    """Run fn with the big-tile kernel enabled and disabled: the two results must agree with the reference alike."""
    outs = []
    for off in ("0", "1"):
        os.environ["WM_GEMM_BIG_OFF"] = off
        try:
            outs.append(fn())
        finally:
            os.environ.pop("WM_GEMM_BIG_OFF", None)
    return outs


# (M, N, K): >= 256 tiles of 256 x 256 each, i.e. launches the dispatcher hands to gemm_nt_big_kernel
BIG_SHAPES = [
    (65536, 256, 256, "proj C=256 trunk (one column tile: whole rows per workgroup)"),
    (65536, 1024, 256, "fc1 C=256 trunk"),
    (65536, 256, 1024, "fc2 C=256 trunk, 16 k steps"),
    (65536, 768, 256, "qkv C=256 trunk, 3 column tiles"),
    (32768 + 96, 2048, 512, "fc1 C=512, ragged M (96 = 3 x 32 rows into the last tile)"),
    (65536 + 160, 512, 2048, "fc2 C=512, 32 k steps, ragged M"),
    (40000 - 32, 1536, 512, "qkv C=512, M % 256 = 224"),
]


@pytest.mark.parametrize("M,N,K,note", BIG_SHAPES)
def test_big_tile_forward_epilogues(wm, M, N, K, note):
    ops = wm.ops
    a, w = rnd("ga" + note, (M, K)), rnd("gw" + note, (N, K), 1.0 / math.sqrt(K))
    bias = rnd("gb" + note, (N,), 1.0, torch.float32)
    ref = a.double() @ w.double().t() + bias.double()

    def epi0():
        out = torch.empty(M, N, device="cuda", dtype=BF)
        ops.linear_fwd(a, w, bias, out, 1)
        return out
    big, small = both(epi0)
    check(big, ref, TOL_BF16, f"EPI0 big {note}")
    check(small, ref, TOL_BF16, f"EPI0 128x128 {note}")
    rows_per = 32 if M % 256 else 256                              # 64-token samples (C = 512 trunk) span 32-row blocks
    nsc = (M + rows_per - 1) // rows_per
    rs = torch.rand(nsc, device="cuda") + 0.5
    res = rnd("gr" + note, (M, N), 1.0, torch.float32)

    def epi1():
        out32 = torch.empty(M, N, device="cuda")
        ops.linear_fwd(a, w, bias, out32, 1, res=res, rowscale=rs, rows_per_scale=rows_per)
        return out32
    ref1 = ref * rs.double().repeat_interleave(rows_per)[:M, None] + res.double()
    big, small = both(epi1)
    check(big, ref1, TOL_F32, f"EPI1 big {note}")
    check(small, ref1, TOL_F32, f"EPI1 128x128 {note}")
    # rows past M must not be touched: the output buffer's tail guard keeps its fill
    guard = torch.full((M + 256, N), 7.0, device="cuda")
    ops.linear_fwd(a, w, bias, guard[:M], 1, res=res, rowscale=rs, rows_per_scale=rows_per)
    assert bool((guard[M:] == 7.0).all()), "rows beyond M were written"


@pytest.mark.parametrize("M,N,K,note", BIG_SHAPES)
def test_big_tile_input_gradient_on_transposed_weights(wm, M, N, K, note):
    """dX[M,K] = epi(dY[M,N] W[N,K]) as dY x (W^T)[K,N]^T: EPI 0 (plain, DropPath row scale) and EPI 2 (stored gelu' factor)."""
    ops = wm.ops
    dy, w = rnd("hdy" + note, (M, N)), rnd("hw" + note, (N, K), 1.0 / math.sqrt(N))
    wT = w.t().contiguous()
    ref = dy.double() @ w.double()
    rows_per = 32 if M % 256 else 256
    rs = torch.rand((M + rows_per - 1) // rows_per, device="cuda") + 0.5
    sc = rs.double().repeat_interleave(rows_per)[:M, None]
    gaux = rnd("hg" + note, (M, K))
    for use_wT in (True, False):
        kw = dict(wT=wT) if use_wT else {}
        dx = torch.empty(M, K, device="cuda", dtype=BF)
        ops.linear_bwd_input(dy, w, dx, 1, **kw)
        check(dx, ref, TOL_BF16, f"dX wT={use_wT} {note}")
        ops.linear_bwd_input(dy, w, dx, 1, rowscale=rs, rows_per_scale=rows_per, **kw)
        check(dx, ref * sc, TOL_BF16, f"dX rowscale wT={use_wT} {note}")
        ops.linear_bwd_input(dy, w, dx, 1, rowscale=rs, rows_per_scale=rows_per, gaux=gaux, gaux_raw=True, **kw)
        check(dx, ref * sc * gaux.double(), TOL_BF16, f"dX EPI2 wT={use_wT} {note}")


def test_big_tile_exact_integers(wm):
    """Small integers multiply and accumulate exactly in bf16 x bf16 -> fp32: any misplaced fragment, swizzle or row shows as
    a wrong integer (asymmetric operands: a transposed or shifted tile cannot pass)."""
    ops = wm.ops
    M, N, K = 65536, 512, 320                                       # 5 k steps: odd count, both LDS buffers, tail logic
    a = ((torch.arange(M, device="cuda")[:, None] * 7 + torch.arange(K, device="cuda")[None, :] * 3) % 11 - 5).to(BF)
    w = ((torch.arange(N, device="cuda")[:, None] * 5 + torch.arange(K, device="cuda")[None, :] * 2) % 7 - 3).to(BF)
    out = torch.empty(M, N, device="cuda")
    zero = torch.zeros(M, N, device="cuda")
    ops.linear_fwd(a, w, None, out, 1, res=zero)
    ref = a.double() @ w.double().t()
    assert torch.equal(out.double(), ref)


def test_transpose_cast_jobs(wm):
    """Two matrices of one arena, transposed and rounded to bf16 in one launch."""
    ops = wm.ops
    arena = rnd("tcj", (8 + 768 * 256 + 8 + 512 * 2048,), 1.0, torch.float32)
    o1, o2 = 8, 8 + 768 * 256 + 8
    jobs = torch.tensor([[o1, 768, 256, 0], [o2, 512, 2048, 12 * 4]], dtype=torch.int32, device="cuda")
    dst = torch.zeros(arena.numel(), device="cuda", dtype=BF)
    ops.transpose_cast_jobs(arena, dst, jobs, 2, 12 * 4 + 8 * 32)
    assert torch.equal(dst[o1:o1 + 768 * 256].view(256, 768), arena[o1:o1 + 768 * 256].view(768, 256).t().to(BF))
    assert torch.equal(dst[o2:o2 + 512 * 2048].view(2048, 512), arena[o2:o2 + 512 * 2048].view(512, 2048).t().to(BF))
    assert bool((dst[:o1] == 0).all()) and bool((dst[o1 + 768 * 256:o2] == 0).all())


def test_transpose_cast_jobs_partial_tiles(wm):
    """The q | k | v weights of the C <= 128 blocks ([96, 32], [192, 64], [384, 128]) in the same launch as a wide one: edge tiles of
    a 64 x 64 grid are partial, nothing outside a matrix is written."""
    ops = wm.ops
    shapes = [(96, 32), (768, 256), (192, 64), (384, 128), (100, 36)]
    offs, jobs, off, tile0 = [], [], 8, 0
    for r, c in shapes:
        offs.append(off)
        jobs.append([off, r, c, tile0])
        tile0 += ((r + 63) // 64) * ((c + 63) // 64)
        off += r * c + 8
    arena = rnd("tcjp", (off,), 1.0, torch.float32)
    dst = torch.zeros(off, device="cuda", dtype=BF)
    ops.transpose_cast_jobs(arena, dst, torch.tensor(jobs, dtype=torch.int32, device="cuda"), len(jobs), tile0)
    mask = torch.ones(off, dtype=torch.bool, device="cuda")
    for (r, c), o in zip(shapes, offs):
        assert torch.equal(dst[o:o + r * c].view(c, r), arena[o:o + r * c].view(r, c).t().to(BF)), (r, c)
        mask[o:o + r * c] = False
    assert bool((dst[mask] == 0).all())


# (tokens, N_out, K_in): weight-gradient launches of the B = 256 step that the dispatcher hands to gemm_tn_big_kernel
TN_BIG = [
    (65536, 1024, 256, "fc1 C=256 trunk: 4 tiles x 64 token ranges"),
    (65536, 256, 1024, "fc2 C=256 trunk"),
    (65536, 768, 256, "qkv C=256 trunk: 3 tiles"),
    (65536, 256, 256, "proj C=256: one tile x 256 token ranges of 4 k steps"),
    (32768 + 64 * 5, 2048, 512, "fc1 C=512, token count not a multiple of the range length"),
    (16384, 512, 2048, "fc2 C=512, trunk token count"),
    (16384, 1536, 512, "qkv C=512 (dy row stride 3072 B)"),
]


@pytest.mark.parametrize("T,N,K,note", TN_BIG)
def test_big_tile_weight_gradient(wm, T, N, K, note):
    """dW[N,K] += dY^T X and db += column sums of dY through the scratch slices: own reduce kernel, the deferred reduction,
    and the 128 x 128 kernel on the same operands as a cross-check."""
    ops = wm.ops
    dy, x = rnd("wdy" + note, (T, N), 0.05), rnd("wx" + note, (T, K))
    ref, refb = dy.double().t() @ x.double(), dy.double().sum(0)
    scr = torch.empty(int(wm.lib.wm_gemm_scratch_floats()), device="cuda")
    tol = 3e-4

    def run():
        dw, db = torch.ones(N, K, device="cuda"), torch.zeros(N, device="cuda")
        ops.linear_bwd_weight(dy, x, dw, db, 1, scratch=scr)
        return dw - 1.0, db
    (dw_b, db_b), (dw_s, db_s) = both(run)
    check(dw_b, ref, tol, f"dW big {note}")
    check(db_b, refb, tol, f"db big {note}")
    check(dw_s, ref, tol, f"dW 128x128 {note}")
    jobs = ops.ReduceJobs()
    dw, db = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
    ops.linear_bwd_weight(dy, x, dw, db, 1, scratch=scr, defer=jobs.take(1))
    ops.reduce_jobs(jobs)
    check(dw, ref, tol, f"dW deferred {note}")
    check(db, refb, tol, f"db deferred {note}")


def test_big_tile_weight_gradient_exact_integers(wm):
    ops = wm.ops
    T, N, K = 16384, 512, 256
    dy = ((torch.arange(T, device="cuda")[:, None] * 3 + torch.arange(N, device="cuda")[None, :] * 5) % 7 - 3).to(BF)
    x = ((torch.arange(T, device="cuda")[:, None] * 5 + torch.arange(K, device="cuda")[None, :] * 7) % 5 - 2).to(BF)
    scr = torch.empty(int(wm.lib.wm_gemm_scratch_floats()), device="cuda")
    dw, db = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
    ops.linear_bwd_weight(dy, x, dw, db, 1, scratch=scr)
    assert torch.equal(dw.double(), dy.double().t() @ x.double())     # |sums| < 2^24: exact in fp32 in any order
    assert torch.equal(db.double(), dy.double().sum(0))
