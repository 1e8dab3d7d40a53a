"""Training soak on one MI355X: 120 captured train steps (B = 8, full attack stack) in bf16 and fp32 mode from the same weights;
prints the total loss every 20 steps and fails on a non-finite value.  Round 1: bf16 9.5151 -> 8.8405, fp32 9.5151 -> 8.8392."""
import importlib, sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
wm = importlib.import_module("image-in-speech-watermarking_amd")
from synth import load_synth
import bench
for mode in ("bf16", "fp32"):
    net = wm.UformerAudio(img_size=128, embed_dim=32, win_size=8, token_projection="linear", token_mlp="leff", depths=[1,2,8,8,2,8,8,2,1], modulator=True, dd_in=2, in_chans=2, audio_scale="0", compute_mode=mode)
    load_synth(net); net = net.cuda().train()
    opt = wm.FusedAdamW(net, lr=2e-4, betas=(0.9,0.999), eps=1e-8, weight_decay=0.02)
    B = 8
    step = wm.TrainStep(net, opt, B, attack=wm.AttackConfig.full_stack(), extract_from="attacked")
    tiles, n = wm.dataprep.stft_tiles(bench.synth_waves(B, 7).cuda())
    x = tiles.view(B, n, 2, 128, 128)[:, 0].contiguous()
    msg = torch.randint(0, 2, (B, 1, 32, 32), generator=torch.Generator().manual_seed(1)).float().cuda()
    hist = []
    for i in range(120):
        step.step(x, msg)
        if i % 20 == 0 or i == 119:
            l = step.losses.cpu().tolist()
            hist.append(round(l[4], 4))
            assert all(v == v and abs(v) < 1e6 for v in l[:6]), l
    print(mode, "total loss every 20 steps:", hist)
