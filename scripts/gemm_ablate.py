#!/usr/bin/env python3
"""Where the 256 x 256 tile GEMM spends its time: the forward launch of a few step shapes with parts of the kernel switched off
(WM_GEMM_DBG bits: 1 no operand loads, 2 no MFMAs, 4 no epilogue).  Results are garbage by design; timing only.
The switches are template instances that exist only in an ablation build of the library:
    make -C image-in-speech-watermarking_amd/csrc clean && make -C image-in-speech-watermarking_amd/csrc ABLATE=1
(the product build carries none of them; rebuild without ABLATE afterwards)."""
import importlib, os, sys, math
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wm = importlib.import_module("image-in-speech-watermarking_amd")
ops = wm.ops
BF = torch.bfloat16


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(2e8))
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


shapes = [(65536, 2048, 512, "d0 fc1 (bf16 out)", 0), (65536, 512, 2048, "d0 fc2 (fp32 out + res)", 1), (262144, 1024, 256, "d1 fc1", 0),
          (262144, 256, 1024, "d1 fc2 (fp32 out + res)", 1), (65536, 768, 256, "t3 qkv", 0)]
print(f"{'shape':40s} " + " ".join(f"{n:>10s}" for n in ("full", "no-load", "no-mfma", "no-epi", "ld only", "mfma only", "epi only")) + "   (us)")
for M, N, K, name, epi in shapes:
    nrot = max(2, int(600e6 // (2 * M * (N + K))) + 1)
    xs = [torch.randn(M, K, device="cuda").to(BF) for _ in range(nrot)]
    ys = [torch.empty(M, N, device="cuda", dtype=BF if epi == 0 else torch.float32) for _ in range(nrot)]
    res = torch.randn(M, N, device="cuda") if epi else None
    w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(BF)
    bias = torch.randn(N, device="cuda")
    ctr = [0]

    def run():
        ctr[0] = (ctr[0] + 1) % nrot
        ops.linear_fwd(xs[ctr[0]], w, bias, ys[ctr[0]], 1, res=res)
    t = []
    for dbg in (0, 1, 2, 4, 6, 5, 3):
        os.environ["WM_GEMM_DBG"] = str(dbg)
        t.append(timeit(run))
    os.environ.pop("WM_GEMM_DBG")
    tf = 2.0 * M * N * K / 1e6
    print(f"{name:24s} {M:7d}x{N:4d}x{K:4d} " + " ".join(f"{v:10.1f}" for v in t) + f"   {tf / t[0]:5.0f} TF/s full, {tf / t[5]:5.0f} TF/s MFMA + LDS reads only", flush=True)
