#!/usr/bin/env python3
"""Time of the in-loop attack kernels by stage (flags subsets) at a bench-like batch."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wm = importlib.import_module("image-in-speech-watermarking_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 96
A = wm.AttackConfig
step = torch.zeros(1, dtype=torch.int64, device="cuda")
x = 0.1 * torch.randn(B, 8002, device="cuda")
g = torch.randn(B, 8002, device="cuda")


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n):
            fn()
    gr.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); gr.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for name, cfg in (("amp only", A(amp=0.8)), ("resample", A(resample=True)), ("awgn", A(snr_db=20.0)), ("lowpass mfma", A(lowpass=True)),
                  ("lowpass scalar", A(lowpass=True, mfma_fir=False)), ("full", A.full_stack())):
    out, sg = wm.attacks.attack_forward(cfg, x, step)
    dw = torch.empty_like(x)
    f = timeit(lambda: wm.attacks.attack_forward(cfg, x, step, out=out, sigma=sg))
    b = timeit(lambda: wm.attacks.attack_backward(cfg, x, g, sg, step, dwave=dw))
    print(f"{name:16s} fwd {f:7.2f} us   bwd {b:7.2f} us")
