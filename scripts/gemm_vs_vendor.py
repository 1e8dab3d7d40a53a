#!/usr/bin/env python3
"""Per-shape timing of the forward / dX / dW GEMMs of the step at B = 96: libwm_hip kernels against the vendor GEMM
torch.matmul dispatches to (hipBLASLt / rocBLAS), plain product only (no epilogue) — a yardstick, not a product path."""
import importlib, os, sys, math
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wm = importlib.import_module("image-in-speech-watermarking_amd")
ops = wm.ops
BF = torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3):
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


shapes = [(98304, 384, 128, "t2 qkv"), (98304, 512, 128, "t2 fc1"), (98304, 128, 512, "t2 fc2"), (98304, 128, 128, "t2 proj"),
          (24576, 768, 256, "t3 qkv"), (24576, 1024, 256, "t3 fc1"), (24576, 256, 1024, "t3 fc2"), (24576, 256, 256, "t3 proj"),
          (24576, 1536, 512, "d0 qkv"), (24576, 2048, 512, "d0 fc1"), (24576, 512, 2048, "d0 fc2"), (24576, 512, 512, "d0 proj"),
          (98304, 768, 256, "d1 qkv"), (98304, 1024, 256, "d1 fc1"), (98304, 256, 1024, "d1 fc2"), (98304, 256, 256, "d1 proj"),
          (393216, 384, 128, "d2 qkv"), (1572864, 96, 32, "t0 qkv"), (1572864, 128, 32, "t0 fc1")]
scr = torch.empty(int(wm.lib.wm_gemm_scratch_floats()), device="cuda")
print(f"{'shape':28s} {'fwd ours':>9s} {'vendor':>8s} | {'dX ours':>8s} {'vendor':>8s} | {'dW ours':>8s} {'vendor':>8s}   (us)")
tot = [0.0] * 6
for M, N, K, name in shapes:
    x = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(BF)
    bias = torch.randn(N, device="cuda"); y = torch.empty(M, N, device="cuda", dtype=BF)
    dy = torch.randn(M, N, device="cuda").to(BF); dx = torch.empty(M, K, device="cuda", dtype=BF)
    dw = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda")
    wt = w.t().contiguous()
    t = [timeit(lambda: ops.linear_fwd(x, w, bias, y, 1)), timeit(lambda: torch.mm(x, wt, out=y)),
         timeit(lambda: ops.linear_bwd_input(dy, w, dx, 1)), timeit(lambda: torch.mm(dy, w, out=dx)),
         timeit(lambda: ops.linear_bwd_weight(dy, x, dw, db, 1, scratch=scr)), timeit(lambda: torch.mm(dy.t(), x))]
    for i in range(6):
        tot[i] += t[i]
    print(f"{name:8s} {M:8d}x{N:4d}x{K:4d} {t[0]:9.1f} {t[1]:8.1f} | {t[2]:8.1f} {t[3]:8.1f} | {t[4]:8.1f} {t[5]:8.1f}")
print("sum".ljust(28), " ".join(f"{v:9.1f}" for v in tot))
