"""Backward-GEMM sweep over the LeWin stages at B = 96 (cold operands): dX (gemm_nt_kernel, transposed weight reads) and dW
(gemm_tn_kernel + slice reduction) of the four Linear layers, with the operand types the bf16 step uses."""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wm = importlib.import_module("image-in-speech-watermarking_amd")
ops = wm.ops
bf = torch.bfloat16
dev = "cuda"
B = int(os.environ.get("B", "96"))
iters = 10
scratch = torch.empty(int(ops.lib.wm_gemm_scratch_floats()), device=dev)
for hw, C in ((128, 32), (64, 64), (32, 128), (16, 256), (8, 512), (128, 64), (64, 128), (32, 256), (16, 512)):
    if os.environ.get("MAXC") and C > int(os.environ["MAXC"]):
        continue
    if os.environ.get("MINC") and C < int(os.environ["MINC"]):
        continue
    M = B * hw * hw
    # layer: (name, N_out, K_in, gaux on dX)
    for name, N, K, gaux in (("qkv", 3 * C, C, 0), ("proj", C, C, 0), ("fc1", 4 * C, C, 0), ("fc2", C, 4 * C, 1)):
        for kind in os.environ.get("KINDS", "dX,dW").split(","):
            by = M * N * 2 + (M * K * 2 * (2 if gaux and kind == "dX" else 1)) + N * K * (2 if kind == "dX" else 4)
            nrot = max(2, int(700e6 // by) + 1)
            bufs = []
            for _ in range(nrot):
                dy = torch.randn(M, N, device=dev).to(bf)
                if kind == "dX":
                    w = torch.randn(N, K, device=dev).to(bf)
                    aux = torch.randn(M, K, device=dev).to(bf) if gaux else None
                    bufs.append((dy, w, aux, torch.empty(M, K, device=dev, dtype=bf)))
                else:
                    bufs.append((dy, torch.randn(M, K, device=dev).to(bf), torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)))
            if kind == "dX":
                f = lambda i: ops.linear_bwd_input(bufs[i % nrot][0], bufs[i % nrot][1], bufs[i % nrot][3], 1, gaux=bufs[i % nrot][2], gaux_raw=True)
            else:
                f = lambda i: ops.linear_bwd_weight(bufs[i % nrot][0], bufs[i % nrot][1], bufs[i % nrot][2], bufs[i % nrot][3], 1, scratch=scratch)
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
            print(f"hw={hw:3d} C={C:3d} {name:4s} {kind} M={M:7d} N={N:4d} K={K:4d}  {us:7.1f} us  {by / us / 1e3:6.0f} GB/s  {2.0 * M * N * K / us / 1e6:6.1f} TF/s", flush=True)
            del bufs
