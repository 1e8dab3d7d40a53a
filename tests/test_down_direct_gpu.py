"""Downsample (Conv2d 4 x 4, stride 2, padding 1, C -> 2C; reference uformerWM/model.py:759-775) without the im2col image:
the 2 x 2-block grid layout passes (csrc/conv_grid.hip) + wm_gemm with grid addressing (WmGemm::a_grid_n / a_jump_k / b_seg_cols),
forward / dX / dW at the four encoder stage shapes of the model and small ragged ones, against torch's conv2d in float64 on the
SAME bf16-rounded operands.  Bounds: fp32 outputs of bf16 products 1e-3 of the tensor's scale (measured ~1e-6), the input gradient
(stored once as bf16 before the un-blocking pass) 4e-3."""
import importlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wm():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return importlib.import_module("image-in-speech-watermarking_amd")


def bfr(t):
    return t.to(torch.bfloat16).to(t.dtype)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


# (B, hw, C): the four Downsample layers of a trunk (hw 128 / 64 / 32 / 16 with C 32 / 64 / 128 / 256) and small cases whose grid
# rows are not multiples of the GEMM tile (B (hw/2 + 2)^2 = 3 * 36, 5 * 100)
# (16, 32, 128) / (48, 16, 256): enough grid rows (>= 4096, a multiple of 64) for the weight gradient to take the 256 x 256 kernel
SHAPES = [(3, 8, 32), (5, 16, 64), (2, 128, 32), (2, 64, 64), (4, 32, 128), (8, 16, 256), (16, 32, 128), (48, 16, 256)]


@pytest.mark.parametrize("B,hw,C", SHAPES)
def test_direct_downsample_fwd_dx_dw(wm, B, hw, C):
    ops = wm.ops
    assert ops.down_direct_ok(hw, C)
    g = torch.Generator().manual_seed(1000 * hw + C + B)
    n = hw // 2
    x = bfr(torch.randn(B, hw, hw, C, generator=g))
    w = bfr(torch.randn(2 * C, C, 4, 4, generator=g) / (4.0 * C ** 0.5))
    bias = 0.3 * torch.randn(2 * C, generator=g)
    dy = bfr(torch.randn(B, n, n, 2 * C, generator=g))
    xd, wd = x.double().permute(0, 3, 1, 2).requires_grad_(), w.double().requires_grad_()
    bd = bias.double().requires_grad_()
    yr = F.conv2d(xd, wd, bd, stride=2, padding=1)                       # [B, 2C, n, n]
    yr.backward(dy.double().permute(0, 3, 1, 2))
    # ---- kernels
    xc, wc = x.cuda().contiguous(), w.cuda().contiguous()
    w_fwd, w_dx = ops.conv_weights(wc, C)
    y = torch.empty(B * n * n, 2 * C, device="cuda")
    xg = ops.down_direct_fwd(xc.view(B * hw * hw, C), w_fwd, bias.cuda(), y, B, hw, C)
    e = rel(y.view(B, n, n, 2 * C), yr.detach().permute(0, 2, 3, 1))
    assert e <= 1e-3, f"forward {e:.3e}"
    scr = torch.empty(int(wm.lib.wm_gemm_scratch_floats()), device="cuda")
    for dyd in (dy.cuda().view(B * n * n, 2 * C).contiguous(), dy.cuda().view(B * n * n, 2 * C).to(torch.bfloat16)):
        dx = torch.empty(B * hw * hw, C, device="cuda")
        dwf, db = torch.zeros(2 * C, 16 * C, device="cuda"), torch.zeros(2 * C, device="cuda")
        ops.down_direct_bwd(dyd, xg, w_dx, dx, dwf, db, B, hw, C, scratch=scr)
        grad = torch.ones(2 * C, C, 4, 4, device="cuda")                 # the gradient arena accumulates
        ops.conv_dweight(dwf, grad, C)
        e = rel(dx.view(B, hw, hw, C), xd.grad.permute(0, 2, 3, 1))
        assert e <= 4e-3, f"dX {e:.3e}"
        e = rel(grad - 1.0, wd.grad)
        assert e <= 1e-3, f"dW {e:.3e}"
        e = rel(db, bd.grad)
        assert e <= 1e-3, f"db {e:.3e}"
    big = (B, hw, C) in ((16, 32, 128), (48, 16, 256))
    assert wm.lib.wm_last_gemm_family() == (4 if big else 2)              # weight gradient: gemm_tn_pp_kernel / gemm_tn_kernel, both with grid addressing


def test_grid_addressing_is_refused_where_it_does_not_exist(wm):
    ops = wm.ops
    a = torch.zeros(4 * 36, 128, device="cuda", dtype=torch.bfloat16)
    b = torch.zeros(64, 512, device="cuda", dtype=torch.bfloat16)
    c = torch.zeros(4 * 16, 64, device="cuda")
    with pytest.raises(wm._lib.WmKernelError):                            # fp32 mode
        ops.gemm(a.float(), b.float(), c, 64, 64, 512, 128, 512, 64, 1, 1, mode=ops.MODE_FP32, a_grid_n=4, a_jump_k=256, a_jump=512)
    with pytest.raises(wm._lib.WmKernelError):                            # jump inside a k tile
        ops.gemm(a, b, c, 64, 64, 512, 128, 512, 64, 1, 1, mode=ops.MODE_BF16, a_grid_n=4, a_jump_k=64, a_jump=512)
