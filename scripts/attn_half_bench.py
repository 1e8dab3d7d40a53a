#!/usr/bin/env python3
"""The attention half of a LeWin block (norm1 -> qkv -> window attention -> proj -> + residual, reference model.py:982-1015) as the
step launches it today, per kernel, at the (hw, C) shapes of the blocks, as graph replays.
  python scripts/attn_half_bench.py [B]"""
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
SHAPES = ((128, 32, 2), (64, 64, 4), (32, 128, 16), (64, 128, 2), (128, 64, 1), (16, 256, 16), (32, 256, 8), (16, 512, 8), (8, 512, 4))


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


tot, ftot = {}, {}
print(f"B = {B}; us per launch")
for hw, C, nblk in SHAPES:
    M, heads = B * hw * hw, C // 32
    torch.manual_seed(0)
    x, x1 = torch.randn(M, C, device=dev), torch.empty(M, C, device=dev)
    g, b_ = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    xn1, xn2 = torch.empty(M, C, device=dev, dtype=bf), torch.empty(M, C, device=dev, dtype=bf)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    wq, bq = (torch.randn(3 * C, C, device=dev) * 0.1).to(bf), torch.zeros(3 * C, device=dev)
    wp, bp = (torch.randn(C, C, device=dev) * 0.1).to(bf), torch.zeros(C, device=dev)
    table = torch.randn(225, heads, device=dev) * 0.5
    qkv, o = torch.empty(M, 3 * C, device=dev, dtype=bf), torch.empty(M, C, device=dev, dtype=bf)
    lse = torch.empty(M, heads, device=dev)
    dx1, dx1h = torch.randn(M, C, device=dev), torch.randn(M, C, device=dev).to(bf)
    do, dqkv, dxn1 = torch.empty_like(o), torch.empty_like(qkv), torch.empty_like(xn1)
    dx, dxh = torch.empty_like(x), torch.empty_like(xn1)
    gwq, gbq, gwp, gbp = torch.zeros(3 * C, C, device=dev), torch.zeros(3 * C, device=dev), torch.zeros(C, C, device=dev), torch.zeros(C, device=dev)
    gg, gb, gt = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(225, heads, device=dev)
    scr = torch.empty(int(ops.lib.wm_gemm_scratch_floats()), device=dev)
    scl = torch.empty(int(ops.lib.wm_layernorm_bwd_scratch_floats(C)), device=dev)
    sca = torch.empty(int(ops.lib.wm_winattn_bwd_scratch_floats()), device=dev)
    r = {}
    r["ln1"] = timeit(lambda: ops.layernorm_fwd(x, g, b_, None, xn1, mean, rstd, hw, 0))
    r["qkv"] = timeit(lambda: ops.linear_fwd(xn1, wq, bq, qkv, 1))
    r["attn"] = timeit(lambda: ops.winattn_fwd(qkv, table, o, B, hw, C, 4 if hw > 8 else 0, 1, lse=lse))
    lnf = ops.ln_fusable(M, C, C, 1)
    if lnf:
        r["proj+ln2"] = timeit(lambda: ops.linear_fwd(o, wp, bp, x1, 1, res=x, ln=(g, b_, None, xn2, mean, rstd, hw, 0)))
    else:
        r["proj+ln2"] = timeit(lambda: ops.linear_fwd(o, wp, bp, x1, 1, res=x)) + timeit(lambda: ops.layernorm_fwd(x1, g, b_, None, xn2, mean, rstd, hw, 0))
    r["proj_dX"] = timeit(lambda: ops.linear_bwd_input(dx1h, wp, do, 1))
    r["proj_dW"] = timeit(lambda: ops.linear_bwd_weight(dx1h, o, gwp, gbp, 1, scratch=scr))
    r["attn_bwd"] = timeit(lambda: ops.winattn_bwd(qkv, table, do, dqkv, gt, B, hw, C, 4 if hw > 8 else 0, 1, out=o, lse=lse, scratch=sca))
    r["qkv_dX"] = timeit(lambda: ops.linear_bwd_input(dqkv, wq, dxn1, 1))
    r["qkv_dW"] = timeit(lambda: ops.linear_bwd_weight(dqkv, xn1, gwq, gbq, 1, scratch=scr))
    r["ln1_bwd"] = timeit(lambda: ops.layernorm_bwd(dxn1, x, g, mean, rstd, dx1, dx, gg, gb, None, hw, 0, scratch=scl, dx_bf16=dxh))
    dxn2 = torch.randn(M, C, device=dev).to(bf)
    r["ln2_bwd"] = timeit(lambda: ops.layernorm_bwd(dxn2, x1, g, mean, rstd, dx1, dx, gg, gb, None, hw, 0, scratch=scl, dx_bf16=dxh))
    if C <= 128:
        m2, r2 = torch.empty(M, device=dev), torch.empty(M, device=dev)
        wqT = wq.t().contiguous()
        sab = torch.empty(int(ops.lib.wm_attn_half_bwd_scratch_floats(B, hw, C, 0)), device=dev)
        r["fused_bwd"] = timeit(lambda: ops.attn_half_bwd(dxn2, dx1, x1, mean, rstd, g, x, mean, rstd, g, qkv, o, lse, table, wp, wqT, dx1h, dqkv, dx,
                                                          gg, gb, gg, gb, gt, B, hw, C, 4 if hw > 8 else 0, sab, dxh=dxh))
        sh = 4 if hw > 8 else 0
        r["fused_fwd"] = timeit(lambda: ops.attn_half_fwd(x, g, b_, None, wq, bq, table, wp, bp, g, b_, x1, xn2, mean, rstd, m2, r2, lse, B, hw, C, sh))
        r["fused_fwd_save"] = timeit(lambda: ops.attn_half_fwd(x, g, b_, None, wq, bq, table, wp, bp, g, b_, x1, xn2, mean, rstd, m2, r2, lse, B, hw, C, sh,
                                                                xn1=xn1, qkv=qkv, o=o))
    fwd = r["ln1"] + r["qkv"] + r["attn"] + r["proj+ln2"]
    bwd = r["ln2_bwd"] + r["proj_dX"] + r["proj_dW"] + r["attn_bwd"] + r["qkv_dX"] + r["qkv_dW"] + r["ln1_bwd"]
    if "fused_bwd" in r:
        fb = r["fused_bwd"] + r["proj_dW"] + r["qkv_dW"]
        ftot[(hw, C)] = (nblk * r["fused_fwd_save"] / 1e3, nblk * fb / 1e3)
    tot[(hw, C)] = (nblk * fwd / 1e3, nblk * bwd / 1e3)
    print(f"hw {hw:3d} C {C:3d} x{nblk:2d} | fwd {fwd:7.1f} = " + "+".join(f"{r[k]:.0f}" for k in ("ln1", "qkv", "attn", "proj+ln2")) +
          (f" [fused {r['fused_fwd']:.0f}, saving xn1|qkv|o {r['fused_fwd_save']:.0f}]" if "fused_fwd" in r else "") + f" | bwd {bwd:7.1f} = " + "+".join(f"{r[k]:.0f}" for k in ("ln2_bwd", "proj_dX", "proj_dW", "attn_bwd", "qkv_dX", "qkv_dW", "ln1_bwd")) +
          (f" [fused {r['fused_bwd']:.0f} + the two dW {r['proj_dW'] + r['qkv_dW']:.0f}]" if "fused_bwd" in r else ""), flush=True)
    # ideal fused traffic: forward x in, x1 + xn2 out (10 C); backward x, dx1 in, dx + dxh out (+ dW partials): ~14 C
    del x, x1, xn1, xn2, qkv, o, dx1, dx1h, do, dqkv, dxn1, dx, dxh
    torch.cuda.empty_cache()
small = sum(v[0] for k, v in tot.items() if k[1] <= 128), sum(v[1] for k, v in tot.items() if k[1] <= 128)
big = sum(v[0] for k, v in tot.items() if k[1] > 128), sum(v[1] for k, v in tot.items() if k[1] > 128)
print(f"per step, ms: C <= 128 fwd {small[0]:.2f} bwd {small[1]:.2f} | C >= 256 fwd {big[0]:.2f} bwd {big[1]:.2f}")
print(f"fused (C <= 128), ms: fwd {sum(v[0] for v in ftot.values()):.2f} bwd {sum(v[1] for v in ftot.values()):.2f}")
