"""Times the Downsample data-movement kernels (im2col / col2im of the 4 x 4 stride-2 convolution) at the step's shapes.\n  python scripts/im2col_probe.py"""
import importlib, sys, torch
sys.path.insert(0, '.')
wm = importlib.import_module("image-in-speech-watermarking_amd"); ops = wm.ops
B = 192
def timeit(fn, n=10):
    for _ in range(2): fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for hw, C in ((128, 32), (64, 64), (32, 128), (16, 256)):
    x = torch.randn(B * hw * hw, C, device="cuda")
    cols = torch.empty(B * hw * hw // 4, 16 * C, device="cuda", dtype=torch.bfloat16)
    us = timeit(lambda: ops.im2col(x, cols, B, hw, C))
    dx = torch.empty_like(x)
    us2 = timeit(lambda: ops.col2im(cols, dx, B, hw, C))
    print(f"hw={hw} C={C}: im2col {us:.1f} us  {(x.numel() * 4 + cols.numel() * 2) / us / 1e6:.2f} TB/s   col2im {us2:.1f} us  {(x.numel() * 4 + cols.numel() * 2) / us2 / 1e6:.2f} TB/s", flush=True)
