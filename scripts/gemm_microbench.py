"""Standalone timing of representative GEMM launches of the train step (B=24), for kernel work under rocprofv3."""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wm = importlib.import_module("image-in-speech-watermarking_amd")
ops = wm.ops
bf = torch.bfloat16
dev = "cuda"
iters = int(os.environ.get("ITERS", "20"))
shapes = [("fwd", 24576, 256, 1024), ("fwd", 24576, 1024, 256), ("fwd", 6144, 512, 2048), ("fwd", 98304, 128, 128),
          ("dX", 24576, 1024, 256), ("dX", 6144, 2048, 512), ("dW", 512, 2048, 6144), ("dW", 512, 128, 24576)]
only = os.environ.get("ONLY")
for kind, M, N, K in shapes:
    if only and only != kind:
        continue
    if kind == "fwd":
        x = torch.randn(M, K, device=dev).to(bf); w = torch.randn(N, K, device=dev).to(bf); b = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=bf)
        f = lambda: ops.linear_fwd(x, w, b, out, 1)
    elif kind == "dX":
        dy = torch.randn(M, K, device=dev).to(bf); w = torch.randn(K, N, device=dev).to(bf)
        out = torch.empty(M, N, device=dev, dtype=bf)
        f = lambda: ops.linear_bwd_input(dy, w, out, 1)
    else:
        dy = torch.randn(K, M, device=dev).to(bf); x = torch.randn(K, N, device=dev).to(bf)
        dw = torch.zeros(M, N, device=dev); db = torch.zeros(M, device=dev)
        f = lambda: ops.linear_bwd_weight(dy, x, dw, db, 1)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / iters
    print(f"{kind:4s} M={M:6d} N={N:5d} K={K:6d}  {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s", flush=True)
