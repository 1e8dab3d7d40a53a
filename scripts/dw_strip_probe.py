import importlib, os, sys, torch
sys.path.insert(0, ".")
wm = importlib.import_module("image-in-speech-watermarking_amd")
ops = wm.ops
bf = torch.bfloat16
B = 96
for hw, Ch in ((128, 128), (128, 256), (64, 256), (64, 512), (32, 512)):
    h1 = torch.randn(B * hw * hw, Ch, device="cuda").to(bf)
    w = torch.randn(Ch, 9, device="cuda") * 0.2; b = torch.randn(Ch, device="cuda") * 0.1
    a2 = torch.empty_like(h1); g2 = torch.empty_like(h1)
    f = lambda: ops.dwconv_fwd(h1, w, b, None, B, hw, Ch, a2=a2, g2=g2)
    for _ in range(2): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 100
    print(f"hw={hw} Ch={Ch} {us:8.1f} us  {h1.numel() * 6 / us / 1e3:6.0f} GB/s  checksum {a2.float().sum().item():.4f} {g2.float().sum().item():.4f}", flush=True)
