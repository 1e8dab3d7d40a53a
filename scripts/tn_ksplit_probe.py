"""Weight-gradient GEMM (gemm_tn_kernel) time against the number of K slices, for the shapes that run far below the rest."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wm = importlib.import_module("image-in-speech-watermarking_amd")
ops = wm.ops
bf = torch.bfloat16
scratch = torch.empty(int(ops.lib.wm_gemm_scratch_floats()), device="cuda")
for M, N, K in ((24576, 1536, 512), (24576, 2048, 512), (24576, 512, 2048), (98304, 768, 256), (24576, 768, 256), (24576, 256, 256)):
    bufs = [(torch.randn(M, N, device="cuda").to(bf), torch.randn(M, K, device="cuda").to(bf), torch.zeros(N, K, device="cuda"),
             torch.zeros(N, device="cuda")) for _ in range(4)]
    for ks in (0, 4, 6, 8, 10, 12, 16, 24, 32):
        def f(i):
            dy, x, dw, db = bufs[i % 4]
            ops.gemm(dy, x, dw, N, K, M, N, K, K, 0, 0, b_ones=True, C2=db, atomic_out=True, mode=1, scratch=scratch, ksplit=ks)
        for i in range(2):
            f(i)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(8):
            f(i)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / 8
        print(f"tokens={M:6d} out={N:4d} in={K:4d} ksplit={ks:2d}  {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TF/s", flush=True)
