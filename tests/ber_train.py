"""Trained-state decoder BER (second half of BASELINE's metric: "decoder BER parity").

An untrained extractor answers ~0.45 for every payload pixel, so BER sits at 0.5 on both sides and "equal BER" says nothing.
This module trains the HIP TrainStep (bf16, BASELINE config 3: full attack stack in the loop, extractor on the attacked
spectrogram; step = audio_uformer_stft.py:421-540) on a synthetic stream until the payload is actually recovered, then
hands the trained weights to the CPU oracle and compares decoded bits, BER (hidden/test_model.py:60-64) and tensors on
held-out (tile, payload) pairs.  Used by bench.py (``ber_trained`` block) and tests/test_trained_ber_gpu.py.

Payloads: the reference trains on MNIST digits padded to 32 x 32 (audio_uformer_stft.py:210-220) — low-entropy images, not
available offline.  The stand-in keeps that property: an 8 x 8 random bitmap shown as 4 x 4 pixel blocks (64 bits per
payload instead of a digit class + style), fresh bitmaps every step, held-out bitmaps for the evaluation.
Weights are produced in-run and never stored.
"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

FULL_STACK = dict(resample=True, snr_db=20.0, lowpass=True, amp=0.8)


def block_payloads(n, gen, device="cpu"):
    """[n,1,32,32] float payloads: 8 x 8 random bits, each shown as a 4 x 4 block."""
    bits = torch.randint(0, 2, (n, 1, 8, 8), generator=gen).float()
    return bits.repeat_interleave(4, 2).repeat_interleave(4, 3).contiguous().to(device)


def binary_payloads(n, gen, device="cpu"):
    """[n,1,32,32] float payloads of the configs' own kind: BinaryWM, 32 x 32 independent random bits = 1024 bits per payload
    (uformerWM/audio_uformer_stft.py:222-244: ``torch.randint(0, 2, (32, 32))``), fresh ones for every sample."""
    return torch.randint(0, 2, (n, 1, 32, 32), generator=gen).float().to(device)


PAYLOADS = {"block": block_payloads, "binary": binary_payloads}


def tile_pool(wm, n, dev, tag="ber", scale=1.0):
    """n synthetic speech-like STFT tiles through the device data-prep path; `scale` as the reference's
    audio_scale factor (audio_test.py:44-55 scales the tiles into a unit-ish range before training)."""
    from synth import chirp_wave
    waves = torch.stack([chirp_wave(16384 - 128, f"{tag}_{i}") for i in range(n)]).to(dev)
    tiles, n_per = wm.dataprep.stft_tiles(waves, scale)
    return tiles.view(n, n_per, 2, 128, 128)[:, 0].contiguous()


def ber_of(wm_out, bits):
    return float((wm_out.float().round().clamp(0, 1) - bits).abs().mean())


def evaluate(wm, net, x, msg):
    """BER of the extractor output without attack and after the full stack (eval mode; net.attack / extract_from restored)."""
    cfg = wm.AttackConfig(**FULL_STACK)
    was, att0, ext0 = net.training, net.attack, net.extract_from
    net.eval()
    out = {}
    try:
        for name, att, ext in (("no_attack", None, "y"), ("attacked", cfg, "attacked")):
            net.attack, net.extract_from = att, ext
            with torch.no_grad():
                o = net(x, msg)
            out[name] = ber_of(o[3], msg)
            out[name + "_mse"] = float(((o[3].float() - msg) ** 2).mean())
    finally:
        net.attack, net.extract_from = att0, ext0
        net.train(was)
    return out


def train(wm, steps, batch=32, mode="bf16", lr=2e-4, eval_every=50, pool=256, held_out=16, seed=0, log=None, scale=0.05,
          payload="block", eval_at=None):
    """Fresh UformerAudio (its own default initialisation, torch seed `seed`) trained for `steps` captured steps of config 3.
    payload: 'block' (64-bit stand-in for the MNIST digits) or 'binary' (BinaryWM: 1024 random bits, the configs' payload);
    eval_at: explicit list of evaluation steps (else every `eval_every`).  Returns (net, curve, held-out x, held-out payloads)."""
    make = PAYLOADS[payload]
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(seed)
    net = wm.UformerAudio(img_size=128, embed_dim=32, win_size=8, token_projection="linear", token_mlp="leff",
                          depths=[1, 2, 8, 8, 2, 8, 8, 2, 1], modulator=True, dd_in=2, in_chans=2, audio_scale="0",
                          compute_mode=mode)
    net = net.to(dev).train()
    opt = wm.FusedAdamW(net, lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)
    step = wm.TrainStep(net, opt, batch, attack=wm.AttackConfig(**FULL_STACK), extract_from="attacked")
    gen = torch.Generator().manual_seed(seed + 1)
    tiles = tile_pool(wm, pool + held_out, dev, scale=scale)
    xh, mh = tiles[pool:].contiguous(), make(held_out, torch.Generator().manual_seed(seed + 2), dev)
    curve = []
    for it in range(steps + 1):
        if (it in eval_at if eval_at is not None else it % eval_every == 0) or it == steps:
            e = evaluate(wm, net, xh, mh)
            e["step"] = it
            e["loss_terms"] = [round(v, 5) for v in step.losses.cpu().tolist()[:5]] if it else None
            curve.append(e)
            if log:
                log(e)
        if it == steps:
            break
        idx = torch.randint(0, pool, (batch,), generator=gen)
        step.step(tiles[idx.to(dev)], make(batch, gen, dev))
    torch.cuda.synchronize()
    step.release()
    return net, curve, xh, mh


DECISION_BAND = {"fp32": 2e-4, "split": 8e-4}               # twice the `wm` tolerance of the mode (tests/test_trained_ber_gpu.py)


def oracle_parity(wm, net, x, msg, tiles=8):
    """The CPU oracle (fp32) on the TRAINED weights against the HIP path in the trained mode and in fp32 mode: decoded bits,
    BER without / with the attack stack (the kernels' AWGN realisation replayed through the oracle), max relative deviations."""
    from oracle import uformer_wm as O
    from oracle import attacks as OA
    x, msg = x[:tiles].contiguous(), msg[:tiles].contiguous()
    xc, mc = x.cpu(), msg.cpu()
    bits = mc.round().clamp(0, 1)
    cfg = wm.AttackConfig(**FULL_STACK)
    was, mode0, att0, ext0 = net.training, net.compute_mode, net.attack, net.extract_from
    net.eval()
    net._ensure_runtime(x.device)
    z = wm.ops.awgn_noise(wm.attacks.rank_seed(cfg.seed), net._step_t, tiles, 8002).double().cpu().numpy()
    hip = {}
    try:
        for mode in dict.fromkeys((mode0, "fp32", "split")):
            net.compute_mode = mode
            for name, att, ext in (("no_attack", None, "y"), ("attacked", cfg, "attacked")):
                net.attack, net.extract_from = att, ext
                with torch.no_grad():
                    hip[mode, name] = [o.float().cpu() for o in net(x, msg)]
    finally:
        net.compute_mode, net.attack, net.extract_from = mode0, att0, ext0
        net.train(was)
    ref = O.UformerAudio()
    ref.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()})
    ref.eval()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cpu = {}
    with torch.no_grad():
        cpu["no_attack"] = ref(xc, mc)
        attack = lambda w: torch.from_numpy(OA.attack_chain(w.numpy(), z=z, **FULL_STACK)).float()
        cpu["attacked"] = ref(xc, mc, attack=attack, extract_from="attacked")

    def rel(p, q):
        return float((p.double() - q.double()).abs().max() / q.double().abs().max().clamp_min(1e-30))
    out = {"tiles": tiles, "formula": "mean(|round(clip(wm,0,1)) - bits|)  (hidden/test_model.py:60-64)",
           "awgn": "HIP Philox stream replayed through the CPU oracle (wm_awgn_noise)"}
    for name in ("no_attack", "attacked"):
        c = cpu[name]
        margin = float((c[3] - 0.5).abs().min())
        e = {"cpu_fp32": ber_of(c[3], bits), "cpu_min_margin_from_threshold": margin}
        for mode in dict.fromkeys((mode0, "fp32", "split")):
            h = hip[mode, name]
            e[f"hip_{mode}"] = ber_of(h[3], bits)
            eq = h[3].round().clamp(0, 1) == c[3].round().clamp(0, 1)
            e[f"hip_{mode}_bits_equal_cpu"] = float(eq.float().mean())
            if mode in DECISION_BAND:
                # a bit is DECIDED where the oracle's own output is further from the 0.5 threshold than twice the deviation the mode is
                # allowed on `wm` (fp32: 1e-4, split: 4e-4 of max |wm|): inside that band a flip is rounding across the threshold, not a
                # wrong result (a training run whose held-out outputs come within 1e-4 of 0.5 is an ordinary event: 8 192 outputs,
                # an eighth of them undecodable without the attack)
                band = DECISION_BAND[mode] * float(c[3].abs().max())
                decided = (c[3] - 0.5).abs() > band
                e[f"hip_{mode}_bits_equal_cpu_decided"] = float(eq[decided].float().mean()) if bool(decided.any()) else 1.0
                e[f"hip_{mode}_undecided_bits"] = int((~decided).sum())
            e[f"hip_{mode}_max_rel_dev"] = {"stft_new": rel(h[0], c[0]), "noise": rel(h[1], c[1]), "wm_pred": rel(h[2], c[2]),
                                            "wm": rel(h[3], c[3])}
        out[name] = e
    return out


def main():
    import argparse
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--eval-every", type=int, default=50)
    ap.add_argument("--mode", default="bf16")
    ap.add_argument("--scale", type=float, default=0.05)
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--payload", default="block", choices=sorted(PAYLOADS))
    a = ap.parse_args()
    wm = importlib.import_module("image-in-speech-watermarking_amd")
    net, curve, xh, mh = train(wm, a.steps, a.batch, a.mode, a.lr, a.eval_every, scale=a.scale, payload=a.payload,
                               log=lambda e: print(json.dumps(e), file=sys.stderr, flush=True))
    res = {"steps": a.steps, "batch": a.batch, "lr": a.lr, "mode": a.mode, "tile_scale": a.scale, "payload": a.payload, "curve": curve}
    if not a.no_oracle:
        res["oracle_parity"] = oracle_parity(wm, net, xh, mh, 8)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
