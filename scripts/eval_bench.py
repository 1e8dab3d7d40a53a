"""Throughput of the whole-utterance evaluation path (rows N1 / N2) on one MI355X: `python scripts/eval_bench.py`.

Times `evaluate.reconstruct_audio` (encode -> stitch -> attack -> re-analyse -> decode) on synthetic 10 s utterances and
the two transform kernels on their own against their HBM traffic (tiles written / read once, waveform once)."""
import importlib
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
wm = importlib.import_module("image-in-speech-watermarking_amd")


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    dev = torch.device("cuda:0")
    L, U = 160000, 16
    g = torch.Generator().manual_seed(5)
    waves = (0.1 * torch.randn(U, L, generator=g)).to(dev)
    tiles, frames = wm.ops.stft255_tiles(waves)
    n = tiles.shape[0] // U
    t_an = timed(lambda: wm.ops.stft255_tiles(waves))
    t_sy = timed(lambda: wm.ops.istft255_stitch(tiles, n, frames % 128, L))
    bytes_an = waves.numel() * 4 + tiles.numel() * 4
    out = {"utterance_s": L / 16000, "utterances": U, "tiles_per_utterance": n,
           "stft255_tiles_ms": t_an, "stft255_tiles_GBps": bytes_an / t_an / 1e6,
           "istft255_stitch_ms": t_sy, "istft255_stitch_GBps": bytes_an / t_sy / 1e6}
    net = wm.UformerAudio(img_size=128, embed_dim=32, win_size=8, token_projection="linear", token_mlp="leff",
                          depths=[1, 2, 8, 8, 2, 8, 8, 2, 1], modulator=True, dd_in=2, in_chans=2, audio_scale="0")
    net = net.to(dev).eval()
    net.compute_mode = "bf16"
    msg = (torch.rand(1, 1, 32, 32, device=dev) > 0.5).float()
    data = [wm.evaluate.prepare_utterance(waves[i:i + 1]) for i in range(U)]
    for attack in ("closed_loop", "awgn-20"):
        wm.evaluate.reconstruct_audio(data[0], msg, net, attack=attack)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for d in data:
            wm.evaluate.reconstruct_audio(d, msg, net, attack=attack)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[f"reconstruct_audio[{attack}]_audio_s_per_s"] = U * L / 16000 / dt
        out[f"reconstruct_audio[{attack}]_ms_per_utterance"] = 1e3 * dt / U
    for attack in ("closed_loop", "awgn-20"):
        wm.evaluate.reconstruct_audio_batch(data[:4], msg, net, attack=attack)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(0, U, 4):
            wm.evaluate.reconstruct_audio_batch(data[i:i + 4], msg, net, attack=attack)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[f"reconstruct_audio_batch4[{attack}]_audio_s_per_s"] = U * L / 16000 / dt
    print(json.dumps(out))


if __name__ == "__main__":
    main()
