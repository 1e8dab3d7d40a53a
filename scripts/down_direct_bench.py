#!/usr/bin/env python3
"""Downsample (conv 4 x 4, stride 2) at the four stage shapes: im2col + GEMMs + col2im against the grid form (csrc/conv_grid.hip),
forward / backward, graph replays.   python scripts/down_direct_bench.py [B]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wm = importlib.import_module("image-in-speech-watermarking_amd")
ops = wm.ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bf = torch.bfloat16


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


tot = [0.0, 0.0, 0.0, 0.0]
for hw, C in ((128, 32), (64, 64), (32, 128), (16, 256)):
    n = hw // 2
    Mo = B * n * n
    x = torch.randn(B * hw * hw, C, device="cuda")
    w = torch.randn(2 * C, C, 4, 4, device="cuda") * 0.02
    bias = torch.zeros(2 * C, device="cuda")
    dy = torch.randn(Mo, 2 * C, device="cuda")
    scr = torch.empty(int(wm.lib.wm_gemm_scratch_floats()), device="cuda")
    # ---- im2col form
    cols, w2, y = torch.empty(Mo, 16 * C, device="cuda", dtype=bf), torch.empty(2 * C, 16 * C, device="cuda", dtype=bf), torch.empty(Mo, 2 * C, device="cuda")
    dcols, dx, dw2, db = torch.empty_like(cols), torch.empty_like(x), torch.zeros(2 * C, 16 * C, device="cuda"), torch.zeros(2 * C, device="cuda")

    def old_f():
        ops.im2col(x, cols, B, hw, C)
        ops.strided3(w, w2, (2 * C, 16, C), (16 * C, 1, 16), (16 * C, C, 1))
        ops.linear_fwd(cols, w2, bias, y, 1)

    def old_b():
        ops.linear_bwd_input(dy, w2, dcols, 1)
        ops.col2im(dcols, dx, B, hw, C)
        ops.linear_bwd_weight(dy, cols, dw2, db, 1, scratch=scr)
        ops.strided3(dw2, w, (2 * C, 16, C), (16 * C, C, 1), (16 * C, 1, 16), atomic=True)
    # ---- grid form
    hold = {}

    def new_f():
        w_fwd, w_dx = ops.conv_weights(w, C)
        hold["w_dx"] = w_dx
        hold["xg"] = ops.down_direct_fwd(x, w_fwd, bias, y, B, hw, C)

    def new_b():
        ops.down_direct_bwd(dy, hold["xg"], hold["w_dx"], dx, dw2, db, B, hw, C, scratch=scr)
        ops.conv_dweight(dw2, w, C)
    r = [timeit(old_f), timeit(old_b), timeit(new_f), timeit(new_b)]
    for i in range(4):
        tot[i] += r[i]
    print(f"hw {hw:3d} C {C:3d} | im2col form fwd {r[0]:7.1f} bwd {r[1]:7.1f} | grid form fwd {r[2]:7.1f} bwd {r[3]:7.1f}   (us)", flush=True)
    del x, cols, dcols, dx, dy
    torch.cuda.empty_cache()
print(f"sum over the four layers of a trunk, us: im2col form {tot[0] + tot[1]:.0f} (fwd {tot[0]:.0f}, bwd {tot[1]:.0f}), grid form {tot[2] + tot[3]:.0f} (fwd {tot[2]:.0f}, bwd {tot[3]:.0f})")
