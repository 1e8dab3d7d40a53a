"""Soak at the bench batch on one MI355X: N captured train steps of B = 256 (bf16, full attack stack) on a fixed batch; prints the
loss terms every 50 steps, fails on a non-finite value or on a total loss that does not fall.  (The trained-BER leg covers 1600 steps
at B = 32; this one keeps every kernel at its bench occupancy — persistent grids, 256 x 256 tiles, two workgroups per CU — for minutes.)
  python scripts/soak_bench_batch.py [steps] [batch]"""
import importlib, sys, time, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
wm = importlib.import_module("image-in-speech-watermarking_amd")
from synth import load_synth
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
net = wm.UformerAudio(img_size=128, embed_dim=32, win_size=8, token_projection="linear", token_mlp="leff", depths=[1, 2, 8, 8, 2, 8, 8, 2, 1],
                      modulator=True, dd_in=2, in_chans=2, audio_scale="0", compute_mode="bf16")
load_synth(net)
net = net.cuda().train()
opt = wm.FusedAdamW(net, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)
step = wm.TrainStep(net, opt, B, attack=wm.AttackConfig.full_stack(), extract_from="attacked")
tiles, n = wm.dataprep.stft_tiles(bench.synth_waves(B, 7).cuda())
x = tiles.view(B, n, 2, 128, 128)[:, 0].contiguous()
msg = torch.randint(0, 2, (B, 1, 32, 32), generator=torch.Generator().manual_seed(1)).float().cuda()
hist, t0 = [], time.time()
for i in range(steps):
    step.step(x, msg)
    if i % 50 == 0 or i == steps - 1:
        l = step.losses.cpu().tolist()
        assert all(v == v and abs(v) < 1e6 for v in l[:6]), (i, l)
        hist.append(round(l[4], 4))
        print(i, [round(v, 5) for v in l[:6]], f"{time.time() - t0:.0f} s", flush=True)
assert hist[-1] < hist[0], hist
print("ok: total loss", hist[0], "->", hist[-1])
