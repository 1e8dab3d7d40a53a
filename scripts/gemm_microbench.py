"""Standalone timing of representative GEMM launches of the train step (B=24), hot (same buffers: operands served by the
Infinity Cache) and cold (rotating buffers larger than the 256 MiB cache: operands come from HBM, as inside the step)."""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wm = importlib.import_module("image-in-speech-watermarking_amd")
ops = wm.ops
bf = torch.bfloat16
dev = "cuda"
iters = int(os.environ.get("ITERS", "24"))
if os.environ.get("BIG"):
    shapes = [("fwd", 98304, 1024, 256, {}), ("fwd", 98304, 256, 1024, {"res": 1}), ("dX", 98304, 1024, 256, {"gaux": 1, "a32": 1}),
              ("dX", 24576, 2048, 512, {"gaux": 1, "a32": 1})]
else:
  shapes = [("fwd", 24576, 256, 1024, {}), ("fwd", 24576, 256, 1024, {"res": 1}), ("fwd", 24576, 1024, 256, {}),
          ("fwd", 98304, 128, 128, {}), ("fwd", 393216, 96, 32, {}),
          ("dX", 24576, 1024, 256, {}), ("dX", 24576, 1024, 256, {"gaux": 1}), ("dX", 24576, 256, 1024, {"a32": 1}),
          ("dW", 512, 2048, 6144, {}), ("dW", 512, 128, 24576, {}), ("dW", 256, 1024, 24576, {"a32": 1})]
for kind, M, N, K, opt in shapes:
    for cold in (0, 1):
        nrot = 1
        if cold:
            per = max(M * K, M * N) * 2 + M * N * 4
            nrot = max(2, int(600e6 // per) + 1)
        bufs = []
        for _ in range(nrot):
            if kind == "fwd":
                x = torch.randn(M, K, device=dev).to(bf); w = torch.randn(N, K, device=dev).to(bf); b = torch.randn(N, device=dev)
                res = torch.randn(M, N, device=dev) if opt.get("res") else None
                out = torch.empty(M, N, device=dev, dtype=torch.float32 if res is not None else bf)
                bufs.append((x, w, b, res, out))
            elif kind == "dX":
                dy = torch.randn(M, K, device=dev); dy = dy if opt.get("a32") else dy.to(bf)
                w = torch.randn(K, N, device=dev).to(bf)
                aux = torch.randn(M, N, device=dev).to(bf) if opt.get("gaux") else None
                out = torch.empty(M, N, device=dev, dtype=bf)
                bufs.append((dy, w, aux, out))
            else:
                dy = torch.randn(K, M, device=dev); dy = dy if opt.get("a32") else dy.to(bf)
                x = torch.randn(K, N, device=dev).to(bf)
                bufs.append((dy, x, torch.zeros(M, N, device=dev), torch.zeros(M, device=dev)))

        def f(i):
            t = bufs[i % nrot]
            if kind == "fwd":
                ops.linear_fwd(t[0], t[1], t[2], t[4], 1, res=t[3])
            elif kind == "dX":
                ops.linear_bwd_input(t[0], t[1], t[3], 1, gaux=t[2])
            else:
                ops.linear_bwd_weight(t[0], t[1], t[2], t[3], 1)
        for i in range(3):
            f(i)
        torch.cuda.synchronize()
        torch.cuda._sleep(int(2e8))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(iters):
            f(i)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / iters
        print(f"{kind:4s} M={M:6d} N={N:5d} K={K:6d} {str(opt):14s} {'cold' if cold else 'hot ':4s} {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s", flush=True)
        del bufs
