#!/usr/bin/env python3
"""Fused LeFF kernels (csrc/leff_fused.hip) against the launches they replace, at the step's five (hw, C <= 128) shapes, as graph
replays:  forward  leff_fwd            vs  fc1 + depthwise conv (a2, g2) + fc2 (+ residual)
          backward leff_bwd + GELU-dW2 vs  fc2 dX (x gelu') + depthwise backward + fc1 dX + plain dW2
  python scripts/leff_fused_bench.py [B]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wm = importlib.import_module("image-in-speech-watermarking_amd")
ops = wm.ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev, bf = "cuda", torch.bfloat16
SHAPES = ((128, 32, 2), (64, 64, 4), (32, 128, 16), (64, 128, 2), (128, 64, 1))     # (hw, C, blocks per step)


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        for _ in range(n):
            fn()
    gph.replay()
    torch.cuda.synchronize()
    s.record()
    gph.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


tot = {"fwd_fused": 0.0, "fwd_old": 0.0, "bwd_fused": 0.0, "bwd_old": 0.0}
print(f"B = {B}; us per launch; step-weighted totals in ms at the end")
for hw, C, nblk in SHAPES:
    M, Ch = B * hw * hw, 4 * C
    torch.manual_seed(0)
    xn = torch.randn(M, C, device=dev).to(bf)
    x1, x2 = torch.randn(M, C, device=dev), torch.empty(M, C, device=dev)
    w1, b1 = (torch.randn(Ch, C, device=dev) * 0.1).to(bf), torch.zeros(Ch, device=dev)
    w2, b2 = (torch.randn(C, Ch, device=dev) * 0.05).to(bf), torch.zeros(C, device=dev)
    wd, bd = torch.randn(Ch, 1, 3, 3, device=dev) * 0.2, torch.randn(Ch, device=dev) * 0.1
    h1, h2, a2, g2, dh2, dh1 = (torch.empty(M, Ch, device=dev, dtype=bf) for _ in range(6))
    dyh, dxn = torch.randn(M, C, device=dev).to(bf), torch.empty(M, C, device=dev, dtype=bf)
    gwd, gbd = torch.zeros(Ch, 9, device=dev), torch.zeros(Ch, device=dev)
    gw2, gb2 = torch.zeros(C, Ch, device=dev), torch.zeros(C, device=dev)
    scr = torch.empty(int(ops.lib.wm_gemm_scratch_floats()), device=dev)
    scr2 = torch.empty(int(ops.lib.wm_dwconv_bwd_scratch_floats(B, hw, Ch)), device=dev)
    r = {}
    r["leff_fwd"] = timeit(lambda: ops.leff_fwd(xn, x1, w1, b1, wd, bd, w2, b2, x2, B, hw, C, h1=h1, h2=h2))
    r["leff_fwd_nosave"] = timeit(lambda: ops.leff_fwd(xn, x1, w1, b1, wd, bd, w2, b2, x2, B, hw, C))
    r["fc1"] = timeit(lambda: ops.linear_fwd(xn, w1, b1, h1, 1))
    r["dw_fwd"] = timeit(lambda: ops.dwconv_fwd(h1, wd, bd, None, B, hw, Ch, a2=a2, g2=g2))
    r["fc2"] = timeit(lambda: ops.linear_fwd(a2, w2, b2, x2, 1, res=x1))
    ops.leff_fwd(xn, x1, w1, b1, wd, bd, w2, b2, x2, B, hw, C, h1=h1, h2=h2)
    r["leff_bwd"] = timeit(lambda: ops.leff_bwd(dyh, h1, h2, w1, wd, w2, dh1, dxn, gwd, gbd, B, hw, C, scratch=scr2))
    r["dW2_gelu"] = timeit(lambda: ops.linear_bwd_weight(dyh, h2, gw2, gb2, 1, x_op=1, scratch=scr))
    r["fc2_dX"] = timeit(lambda: ops.linear_bwd_input(dyh, w2, dh2, 1, gaux=g2, gaux_raw=True))
    r["dw_bwd"] = timeit(lambda: ops.dwconv_bwd(h1, dh2, wd, dh1, gwd, gbd, B, hw, Ch, scratch=scr2))
    r["fc1_dX"] = timeit(lambda: ops.linear_bwd_input(dh1, w1, dxn, 1))
    r["dW2"] = timeit(lambda: ops.linear_bwd_weight(dyh, a2, gw2, gb2, 1, scratch=scr))
    ff, fo = r["leff_fwd"], r["fc1"] + r["dw_fwd"] + r["fc2"]
    bfu, bo = r["leff_bwd"] + r["dW2_gelu"], r["fc2_dX"] + r["dw_bwd"] + r["fc1_dX"] + r["dW2"]
    tot["fwd_fused"] += nblk * ff; tot["fwd_old"] += nblk * fo; tot["bwd_fused"] += nblk * bfu; tot["bwd_old"] += nblk * bo
    print(f"hw {hw:3d} C {C:3d} x{nblk:2d} | fwd fused {ff:7.1f} (nosave {r['leff_fwd_nosave']:7.1f}) old {fo:7.1f} = {r['fc1']:.0f}+{r['dw_fwd']:.0f}+{r['fc2']:.0f}"
          f" | bwd fused {bfu:7.1f} = {r['leff_bwd']:.0f}+{r['dW2_gelu']:.0f} old {bo:7.1f} = {r['fc2_dX']:.0f}+{r['dw_bwd']:.0f}+{r['fc1_dX']:.0f}+{r['dW2']:.0f}",
          flush=True)
    del xn, x1, x2, h1, h2, a2, g2, dh2, dh1, dyh, dxn
    torch.cuda.empty_cache()
print({k: round(v / 1e3, 2) for k, v in tot.items()})
