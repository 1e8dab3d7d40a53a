#!/usr/bin/env python3
"""Per-shape timing of the forward / dX / dW GEMMs of the step at batch B (environment, default 96): libwm_hip kernels
against the vendor GEMM torch.matmul dispatches to (hipBLASLt / rocBLAS), plain product only (no epilogue) — a yardstick,
not a product path.  Operand sets are rotated so that a launch does not find its inputs in the 256 MB Infinity Cache."""
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


B = int(os.environ.get("B", "96"))
shapes = [(98304, 384, 128, "t2 qkv"), (98304, 512, 128, "t2 fc1"), (98304, 128, 512, "t2 fc2"), (98304, 128, 128, "t2 proj"),
          (24576, 768, 256, "t3 qkv"), (24576, 1024, 256, "t3 fc1"), (24576, 256, 1024, "t3 fc2"), (24576, 256, 256, "t3 proj"),
          (24576, 1536, 512, "d0 qkv"), (24576, 2048, 512, "d0 fc1"), (24576, 512, 2048, "d0 fc2"), (24576, 512, 512, "d0 proj"),
          (98304, 768, 256, "d1 qkv"), (98304, 1024, 256, "d1 fc1"), (98304, 256, 1024, "d1 fc2"), (98304, 256, 256, "d1 proj"),
          (393216, 384, 128, "d2 qkv"), (1572864, 96, 32, "t0 qkv"), (1572864, 128, 32, "t0 fc1")]
shapes = [(M * B // 96, N, K, name) for (M, N, K, name) in shapes]
if os.environ.get("ONLY"):
    shapes = [s for s in shapes if s[3].split()[0] in os.environ["ONLY"].split(",")]
scr = torch.empty(int(wm.lib.wm_gemm_scratch_floats()), device="cuda")
print(f"{'shape':28s} {'fwd ours':>9s} {'vendor':>8s} | {'dX ours':>8s} {'vendor':>8s} | {'dW ours':>8s} {'vendor':>8s}   (us)")
tot = [0.0] * 7
for M, N, K, name in shapes:
    nrot = max(2, int(600e6 // (2 * M * (N + K))) + 1)
    xs = [torch.randn(M, K, device="cuda").to(BF) for _ in range(nrot)]
    ys = [torch.empty(M, N, device="cuda", dtype=BF) for _ in range(nrot)]
    dys = [torch.randn(M, N, device="cuda").to(BF) for _ in range(nrot)]
    dxs = [torch.empty(M, K, device="cuda", dtype=BF) for _ in range(nrot)]
    w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(BF)
    bias = torch.randn(N, device="cuda")
    dw = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda")
    wt = w.t().contiguous()
    ctr = [0]

    def rot():
        ctr[0] = (ctr[0] + 1) % nrot
        return ctr[0]
    t = [timeit(lambda: (lambda i: ops.linear_fwd(xs[i], w, bias, ys[i], 1))(rot())),
         timeit(lambda: (lambda i: torch.mm(xs[i], wt, out=ys[i]))(rot())),
         timeit(lambda: (lambda i: ops.linear_bwd_input(dys[i], w, dxs[i], 1))(rot())),
         timeit(lambda: (lambda i: torch.mm(dys[i], w, out=dxs[i]))(rot())),
         timeit(lambda: (lambda i: ops.linear_bwd_weight(dys[i], xs[i], dw, db, 1, scratch=scr))(rot())),
         timeit(lambda: (lambda i: torch.mm(dys[i].t(), xs[i]))(rot())),
         timeit(lambda: (lambda i: ops.linear_bwd_input(dys[i], w, dxs[i], 1, wT=wt))(rot()))]
    del xs, ys, dys, dxs
    for i in range(7):
        tot[i] += t[i]
    tf = 2.0 * M * N * K / 1e6
    print(f"{name:8s} {M:8d}x{N:4d}x{K:4d} {t[0]:9.1f} {t[1]:8.1f} | {t[2]:8.1f} {t[3]:8.1f} | {t[4]:8.1f} {t[5]:8.1f} | dX on W^T {t[6]:7.1f}   "
          f"TF/s ours {tf / t[0]:5.0f} {tf / min(t[2], t[6]):5.0f} {tf / t[4]:5.0f} vendor {tf / t[1]:5.0f} {tf / t[3]:5.0f}", flush=True)
print("sum".ljust(28), " ".join(f"{v:9.1f}" for v in tot))
