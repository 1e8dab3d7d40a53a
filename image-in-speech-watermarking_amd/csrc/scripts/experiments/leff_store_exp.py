import importlib, os, sys, torch
sys.path.insert(0, "/root/repo")
wm = importlib.import_module("image-in-speech-watermarking_amd"); ops = wm.ops
B = 256; dev, bf = "cuda", torch.bfloat16
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize(); s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for hw, C in ((128, 32), (32, 128)):
    M, Ch = B * hw * hw, 4 * C
    xn = torch.randn(M, C, device=dev).to(bf); x1 = torch.randn(M, C, device=dev); x2 = torch.empty(M, C, device=dev)
    w1, b1 = (torch.randn(Ch, C, device=dev) * 0.1).to(bf), torch.zeros(Ch, device=dev)
    w2, b2 = (torch.randn(C, Ch, device=dev) * 0.05).to(bf), torch.zeros(C, device=dev)
    wd, bd = torch.randn(Ch, 9, device=dev) * 0.2, torch.randn(Ch, device=dev) * 0.1
    h1, h2 = torch.empty(M, Ch, device=dev, dtype=bf), torch.empty(M, Ch, device=dev, dtype=bf)
    r = {}
    for name, kw in (("none", {}), ("h1", dict(h1=h1)), ("h2", dict(h2=h2)), ("both", dict(h1=h1, h2=h2))):
        r[name] = timeit(lambda: ops.leff_fwd(xn, x1, w1, b1, wd, bd, w2, b2, x2, B, hw, C, **kw))
    print(os.environ.get("WM_LEFF_PLAIN_STORES", "nt"), hw, C, {k: round(v, 1) for k, v in r.items()}, flush=True)
