#!/usr/bin/env python3
"""Experiment: does the cost of the fused LeFF forward's saved-tensor stores depend on WHERE the buffers lie?
Times leff_fwd with no / h1 / h2 / both saves (a) alone, (b) with 4 more 1 GB tensors allocated first (the layout of
scripts/leff_fused_bench.py), (c) with the output buffers shifted by odd multiples of 4 KB.  python scripts/leff_store_exp.py"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wm = importlib.import_module("image-in-speech-watermarking_amd"); ops = wm.ops
B = 256; dev, bf = "cuda", torch.bfloat16


def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize(); s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def run(tag, hw, C, extra, shift):
    M, Ch = B * hw * hw, 4 * C
    torch.manual_seed(0)
    xn = torch.randn(M, C, device=dev).to(bf); x1 = torch.randn(M, C, device=dev); x2 = torch.empty(M, C, device=dev)
    w1, b1 = (torch.randn(Ch, C, device=dev) * 0.1).to(bf), torch.zeros(Ch, device=dev)
    w2, b2 = (torch.randn(C, Ch, device=dev) * 0.05).to(bf), torch.zeros(C, device=dev)
    wd, bd = torch.randn(Ch, 9, device=dev) * 0.2, torch.randn(Ch, device=dev) * 0.1
    keep = [torch.empty(M, Ch, device=dev, dtype=bf) for _ in range(extra)]
    def buf(k):
        raw = torch.empty(M * Ch + 65536, device=dev, dtype=bf)
        off = (2048 * (2 * k + 1)) * shift                       # elements: odd multiples of 4 KB
        return raw[off:off + M * Ch].view(M, Ch), raw
    (h1, r1), (h2, r2) = buf(0), buf(1)
    r = {}
    for name, kw in (("none", {}), ("h1", dict(h1=h1)), ("h2", dict(h2=h2)), ("both", dict(h1=h1, h2=h2))):
        r[name] = timeit(lambda: ops.leff_fwd(xn, x1, w1, b1, wd, bd, w2, b2, x2, B, hw, C, **kw))
    print(tag, hw, C, {k: round(v, 1) for k, v in r.items()}, "h1 % 1MiB", h1.data_ptr() % (1 << 20), "h2 % 1MiB", h2.data_ptr() % (1 << 20), flush=True)
    del keep


for hw, C in ((128, 32), (64, 64)):
    run("alone      ", hw, C, 0, 0)
    run("4 x 1 GB    ", hw, C, 4, 0)
    run("shifted     ", hw, C, 0, 1)
    run("4 x 1 GB + sh", hw, C, 4, 1)
    torch.cuda.empty_cache()
