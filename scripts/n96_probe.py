"""Why is the N = 96 forward GEMM (qkv at C = 32) slow?  Same rows, K = 32: N and the output row stride varied."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wm = importlib.import_module("image-in-speech-watermarking_amd")
ops = wm.ops
bf = torch.bfloat16
M, K = 1572864, 32
for N, ldc in ((96, 96), (96, 128), (128, 128), (64, 64), (64, 96), (32, 32), (32, 96)):
    bufs = []
    for _ in range(3):
        x = torch.randn(M, K, device="cuda").to(bf); w = torch.randn(N, K, device="cuda").to(bf); b = torch.randn(N, device="cuda")
        out = torch.empty(M, ldc, device="cuda", dtype=bf)
        bufs.append((x, w, b, out))
    f = lambda i: ops.gemm(bufs[i % 3][0], bufs[i % 3][1], bufs[i % 3][3], M, N, K, K, K, ldc, bias=bufs[i % 3][2], mode=1)
    for i in range(3):
        f(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(12):
        f(i)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / 12
    by = M * (K * 2 + N * 2)
    print(f"N={N:4d} ldc={ldc:4d}  {us:7.1f} us  {by / us / 1e3:6.0f} GB/s (useful bytes)", flush=True)
