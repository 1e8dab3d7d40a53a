"""Forward-GEMM sweep over the LeWin stages at B = 96 (cold operands: rotating buffers larger than the Infinity Cache).
Prints time, algorithmic GB/s and TFLOP/s per shape so that the slow corners of gemm_nt_kernel stand out."""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wm = importlib.import_module("image-in-speech-watermarking_amd")
ops = wm.ops
bf = torch.bfloat16
dev = "cuda"
B = int(os.environ.get("B", "96"))
iters = 12
rows = []
for hw, C in ((128, 32), (64, 64), (32, 128), (16, 256), (8, 512), (128, 64), (64, 128), (32, 256), (16, 512)):
    M = B * hw * hw
    for name, N, K, res in (("qkv", 3 * C, C, 0), ("proj", C, C, 1), ("fc1", 4 * C, C, 0), ("fc2", C, 4 * C, 1)):
        if os.environ.get("ONLY") and name not in os.environ["ONLY"].split(","):
            continue
        if os.environ.get("MAXC") and C > int(os.environ["MAXC"]):
            continue
        by = M * K * 2 + N * K * 2 + M * N * (4 + 4 if res else 2)
        nrot = max(2, int(700e6 // by) + 1)
        bufs = []
        for _ in range(nrot):
            x = torch.randn(M, K, device=dev).to(bf); w = torch.randn(N, K, device=dev).to(bf); b = torch.randn(N, device=dev)
            r = torch.randn(M, N, device=dev) if res else None
            out = torch.empty(M, N, device=dev, dtype=torch.float32 if res else bf)
            bufs.append((x, w, b, r, out))
        f = lambda i: ops.linear_fwd(bufs[i % nrot][0], bufs[i % nrot][1], bufs[i % nrot][2], bufs[i % nrot][4], 1, res=bufs[i % nrot][3])
        for i in range(2):
            f(i)
        torch.cuda.synchronize()
        torch.cuda._sleep(int(1e8))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(iters):
            f(i)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / iters
        print(f"hw={hw:3d} C={C:3d} {name:4s} M={M:7d} N={N:4d} K={K:4d}  {us:7.1f} us  {by / us / 1e3:6.0f} GB/s  {2.0 * M * N * K / us / 1e6:6.1f} TF/s", flush=True)
        del bufs
