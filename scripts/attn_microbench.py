"""Standalone timing of the window-attention / depthwise-conv / LayerNorm kernels at the train step's stage shapes (B=24)."""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wm = importlib.import_module("image-in-speech-watermarking_amd")
ops = wm.ops
bf = torch.bfloat16
dev = "cuda"
iters = int(os.environ.get("ITERS", "20"))
B = int(os.environ.get("B", "24"))
which = os.environ.get("WHICH", "attn,dwconv,ln").split(",")


def timeit(f, nrot):
    for i in range(3):
        f(i)
    torch.cuda.synchronize()
    torch.cuda._sleep(int(2e8))
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        f(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


for hw, C in [(128, 32), (64, 64), (32, 128), (16, 256), (16, 512)]:
    M = B * hw * hw
    nrot = max(2, int(600e6 // (M * C * 2 * 8)) + 1)
    if "attn" in which:
        bufs = []
        for _ in range(nrot):
            qkv = torch.randn(M, 3 * C, device=dev).to(bf)
            bufs.append((qkv, torch.empty(M, C, device=dev, dtype=bf), torch.empty(M, C // 32, device=dev),
                         torch.randn(M, C, device=dev).to(bf), torch.empty_like(qkv)))
        table = 0.1 * torch.randn(225, C // 32, device=dev)
        dt = torch.zeros_like(table)
        scr = torch.empty(1024 * 232, device=dev) if os.environ.get("SCRATCH", "1") == "1" else None
        for shift in (0, 4):
            tf = timeit(lambda i: ops.winattn_fwd(bufs[i % nrot][0], table, bufs[i % nrot][1], B, hw, C, shift, 1, lse=bufs[i % nrot][2]), nrot)
            tb = timeit(lambda i: ops.winattn_bwd(bufs[i % nrot][0], table, bufs[i % nrot][3], bufs[i % nrot][4], dt, B, hw, C, shift, 1,
                                                  out=bufs[i % nrot][1], lse=bufs[i % nrot][2], scratch=scr), nrot)
            byf, byb = M * C * 2 * 4, M * C * 2 * 9
            print(f"attn hw={hw:3d} C={C:3d} shift={shift} fwd {tf:7.1f} us {byf/tf/1e3:7.1f} GB/s | bwd {tb:7.1f} us {byb/tb/1e3:7.1f} GB/s", flush=True)
        del bufs
    if "dwconv" in which:
        Ch = 4 * C
        nr = max(2, int(600e6 // (M * Ch * 2 * 4)) + 1)
        bufs = []
        for _ in range(nr):
            h1 = torch.randn(B, hw, hw, Ch, device=dev).to(bf)
            bufs.append((h1, torch.empty_like(h1), torch.empty_like(h1), torch.randn(B, hw, hw, Ch, device=dev).to(bf), torch.empty_like(h1)))
        w = torch.randn(Ch, 1, 3, 3, device=dev); bb = torch.randn(Ch, device=dev)
        dw = torch.zeros_like(w); db = torch.zeros_like(bb)
        tf = timeit(lambda i: ops.dwconv_fwd(bufs[i % nr][0], w, bb, None, B, hw, Ch, a2=bufs[i % nr][1], g2=bufs[i % nr][2]), nr)
        scr = torch.empty(B * (hw // 8) * Ch * 10, device=dev) if os.environ.get("SCRATCH", "1") == "1" else None
        tb = timeit(lambda i: ops.dwconv_bwd(bufs[i % nr][0], bufs[i % nr][3], w, bufs[i % nr][4], dw, db, B, hw, Ch, scratch=scr), nr)
        byf, byb = M * Ch * 2 * 3, M * Ch * 2 * 3
        print(f"dwconv hw={hw:3d} Ch={Ch:4d} fwd {tf:7.1f} us {byf/tf/1e3:7.1f} GB/s | bwd {tb:7.1f} us {byb/tb/1e3:7.1f} GB/s", flush=True)
        del bufs
    if "ln" in which:
        nr = max(2, int(600e6 // (M * C * 14)) + 1)
        bufs = []
        for _ in range(nr):
            bufs.append((torch.randn(M, C, device=dev), torch.empty(M, C, device=dev, dtype=bf), torch.empty(M, device=dev), torch.empty(M, device=dev),
                         torch.randn(M, C, device=dev).to(bf), torch.empty(M, C, device=dev)))
        gam = torch.randn(C, device=dev); bet = torch.randn(C, device=dev)
        dg = torch.zeros(C, device=dev); dbt = torch.zeros(C, device=dev)
        scr = torch.empty(1024 * C, device=dev) if os.environ.get("SCRATCH", "1") == "1" else None
        tf = timeit(lambda i: ops.layernorm_fwd(bufs[i % nr][0], gam, bet, None, bufs[i % nr][1], bufs[i % nr][2], bufs[i % nr][3], hw, 0), nr)
        tb = timeit(lambda i: ops.layernorm_bwd(bufs[i % nr][4], bufs[i % nr][0], gam, bufs[i % nr][2], bufs[i % nr][3], None, bufs[i % nr][5], dg, dbt, None, hw, 0, scratch=scr), nr)
        byf, byb = M * C * 6, M * C * 10
        print(f"ln hw={hw:3d} C={C:3d} fwd {tf:7.1f} us {byf/tf/1e3:7.1f} GB/s | bwd {tb:7.1f} us {byb/tb/1e3:7.1f} GB/s", flush=True)
        del bufs
