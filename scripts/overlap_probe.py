#!/usr/bin/env python3
"""Does running the weight-gradient GEMMs beside the input-gradient chain pay?  Times pairs of kernels of the backward pass
back to back on one stream and as two parallel branches of one captured graph.
  python scripts/overlap_probe.py [B]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wm = importlib.import_module("image-in-speech-watermarking_amd")
ops = wm.ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 192
dev = "cuda"
bf = torch.bfloat16


def run(graph_fn, n=10):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            graph_fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


side = torch.cuda.Stream()
for hw, C in ((128, 32), (64, 64), (32, 128), (16, 256), (16, 512), (32, 256)):
    M, Ch = B * hw * hw, 4 * C
    h1 = torch.randn(M, Ch, device=dev).to(bf)
    dh2, dh1 = torch.randn(M, Ch, device=dev).to(bf), torch.empty(M, Ch, device=dev, dtype=bf)
    w, dw, db = torch.randn(Ch, 1, 3, 3, device=dev) * 0.2, torch.zeros(Ch, 1, 3, 3, device=dev), torch.zeros(Ch, device=dev)
    a2, dyh = torch.randn(M, Ch, device=dev).to(bf), torch.randn(M, C, device=dev).to(bf)
    gw2, gb2 = torch.zeros(C, Ch, device=dev), torch.zeros(C, device=dev)
    scr = torch.empty(int(ops.lib.wm_gemm_scratch_floats()), device=dev)
    scr2 = torch.empty(int(ops.lib.wm_dwconv_bwd_scratch_floats(B, hw, Ch)), device=dev)
    x, dres, dxn = torch.randn(M, C, device=dev), torch.randn(M, C, device=dev), torch.randn(M, C, device=dev).to(bf)
    mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
    gam, dg, dbt = torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dx, dxh = torch.empty(M, C, device=dev), torch.empty(M, C, device=dev, dtype=bf)
    scr3 = torch.empty(int(ops.lib.wm_layernorm_bwd_scratch_floats(C)), device=dev)

    def dwg():
        ops.linear_bwd_weight(dyh, a2, gw2, gb2, ops.MODE_BF16, scratch=scr)

    def conv():
        ops.dwconv_bwd(h1, dh2, w, dh1, dw, db, B, hw, Ch, scratch=scr2)

    def ln():
        ops.layernorm_bwd(dxn, x, gam, mean, rstd, dres, dx, dg, dbt, None, hw, 0, scratch=scr3, dx_bf16=dxh)

    for name, other in (("dwconv_bwd", conv), ("ln_bwd", ln)):
        def serial():
            dwg()
            other()

        def forked():
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                dwg()
            other()
            cur.wait_stream(side)

        ta, tb, ts, tf = run(dwg), run(other), run(serial), run(forked)
        print(f"hw={hw:3d} C={C:3d}  dW {ta:7.1f} us  {name} {tb:7.1f} us  serial {ts:7.1f}  parallel {tf:7.1f}  ({tf / ts:.2f})", flush=True)
    del h1, dh2, dh1, a2, dyh, x, dres, dxn, dx, dxh
    torch.cuda.empty_cache()
