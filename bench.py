#!/usr/bin/env python3
"""Benchmark of the watermark train step on MI355X (contract: see the task brief / DESIGN.md "Measurement").

  python bench.py --gpus N --steps K --warmup W [--batch B] [--attack full|none] [--mode bf16|fp32]

Metric (BASELINE.json): audio-sec/s per train step (enc+attack+dec); decoder BER parity.  One step = zero_grad ->
forward -> attacks -> 4-term loss -> backward -> (RCCL gradient all-reduce) -> AdamW on B synthetic (STFT tile, payload)
pairs per GPU; a tile is 128 hops x 128 samples = 1.024 s of 16 kHz audio.  The default workload is
BASELINE config 3 (full audio_attack stack in the loop, extractor on the attacked spectrogram).  Prints ONE JSON line.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (one process per GPU, before this
process touches a GPU) and relays rank 0's line; under `python -m torch.distributed.run --nproc-per-node N` the ranks are
already there and RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment.
"""
import argparse
import importlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
TILE_SECONDS = 1.024
FULL_STACK = dict(resample=True, snr_db=20.0, lowpass=True, amp=0.8)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256,
                    help="tiles per GPU.  The per-device batch is free (SURVEY 8: the reference script hard-codes 24, its CLI default "
                         "is 4).  256 tiles hold about 143 GB of the 288 GB at the end of the forward (0.56 GB per tile): at B=24 a launch puts ~1.5 workgroups on each CU and the step is "
                         "latency-bound; the rate rises 0.87 -> 1.32 -> 1.44 -> 1.48 -> 1.51 k audio-s/s over B = 24 / 96 / 192 / 256 / 384. "
                         "The line also reports the B=24 rate (reference_batch) measured in the same run.")
    ap.add_argument("--no-reference-batch", action="store_true", help="skip the extra B=24 measurement")
    ap.add_argument("--attack", default="full", choices=["full", "none"],
                    help="full: BASELINE config 3 (resample + AWGN 20 dB + low-pass + x0.8 between iSTFT and STFT, extractor reads the "
                         "attacked spectrogram); none: config 2 (Identity attack)")
    ap.add_argument("--no-identity-leg", action="store_true", help="skip the extra config-2 (Identity attack) measurement")
    ap.add_argument("--clip-seconds", type=int, default=1, choices=[1, 10],
                    help="10: config 5 (10 s clips = 10 tiles each, 64x64 grayscale payload as four quadrants; --batch must be a multiple of 10)")
    ap.add_argument("--mode", default="bf16", choices=["bf16", "fp32", "split"],
                    help="bf16: the benched mode (BASELINE configs say bf16); fp32 / split: the two modes that meet north_star's 1e-4 / "
                         "bit-exact bound (fp32 MFMA; fp32 storage with every GEMM as three bf16 MFMA products of hi / lo halves)")
    ap.add_argument("--no-parity-mode-leg", action="store_true", help="skip the extra measurement of the tolerance-meeting 'split' mode")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--grad-exchange", default="fp32", choices=["fp32", "bf16"],
                    help="N > 1: how the gradient buckets cross xGMI: fp32 all-reduce, or bf16 all-gather with fp32 accumulation on arrival")
    ap.add_argument("--segmented", action="store_true",
                    help="capture the step as the multi-rank chain of graphs (cut at the gradient buckets) on one rank too")
    ap.add_argument("--no-secondary-legs", action="store_true", help="skip the config-5 (10 s clips) and fp32-mode rates")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU legs (cpu_baseline and the BER / deviation parity block)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--ber-train-steps", type=int, default=2000,
                    help="trained-state BER leg: train a fresh model for this many captured steps (B = 32, config 3) on the synthetic "
                         "stream, then compare the decoded payload of the HIP path and of the CPU oracle on the trained weights "
                         "(0: skip; skipped with --no-cpu-baseline)")
    ap.add_argument("--detail", action="store_true", help="per-shape GEMM timing table on stderr")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="run the N-rank code path with every rank on GPU 0 over gloo (a functional rehearsal on a one-GPU "
                         "box: RCCL refuses two ranks per device; the numbers mean nothing)")
    ap.add_argument("--rehearse-rccl", action="store_true",
                    help="--gpus 1 only: run the N-rank call sequence (chain of graphs cut at the gradient buckets, every bucket's "
                         "collective really issued on the side stream) on a ONE-rank RCCL process group: the sum over one rank is the "
                         "identity, so the losses equal the plain run's; what it shows on a one-GPU box is that RCCL takes the calls "
                         "between the graph replays and how long the host needs to issue one step (allreduce.host_issue)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------- launcher
def visible_gpus():
    """GPUs this process may use, counted WITHOUT starting the HIP runtime (the parent of the rank processes must never touch
    a GPU): the visibility variables if set, else the KFD topology (nodes with SIMDs are GPUs)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(",") if t.strip() != ""])
    n, root = 0, "/sys/class/kfd/kfd/topology/nodes"
    if not os.path.isdir(root):
        return 0                                                 # no AMD GPU driver on this machine
    try:
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except OSError:
        return None                                              # unknown: let the ranks report a shortage themselves
    return n


def launch_ranks(a):
    """Parent of a plain `python bench.py --gpus N` call: starts N rank processes (fresh interpreters; this process never
    initialises a GPU), relays rank 0's JSON line and every rank's stderr, and returns the first non-zero exit code — when a
    rank dies the others are terminated instead of waiting in a collective until the distributed timeout."""
    if any(k.startswith("ROCPROF") or k.startswith("ROCPROFILER") for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        raise SystemExit("bench.py --gpus N under rocprofv3: the profiler's preload has already initialised the GPU in this "
                         "process, starting rank programs from it is the forbidden hop; profile one rank directly "
                         "(rocprofv3 ... -- python3 bench.py --gpus 1 ...)")
    have = visible_gpus()
    if not a.rehearse_one_gpu and have is not None and have < a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {have} GPU(s) visible "
                         "(use --rehearse-one-gpu for a functional rehearsal of the N-rank path on one GPU)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))   # stderr: inherited, every rank
    import threading
    out = []
    reader = threading.Thread(target=lambda: out.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = abs(bad[0][1]) or 1
            print(f"bench.py: rank {bad[0][0]} exited with code {bad[0][1]}; stopping the other ranks", file=sys.stderr)
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.2)
    reader.join(timeout=10)
    # rank 0 prints ONE JSON line; libraries may add their own chatter to its stdout (gloo announces its peers in the one-GPU
    # rehearsal): relay the JSON line alone, the rest goes to stderr
    for line in (out[0] if out else b"").decode().splitlines():
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    return failed


# ---------------------------------------------------------------------------------------------------- workload
def synth_waves(batch, seed):
    """one chirp+noise waveform of 16256 samples (= exactly 128 data-prep frames) per tile."""
    import torch
    from synth import chirp_wave
    return torch.stack([chirp_wave(16384 - 128, f"bench_{seed}_{i}") for i in range(batch)])


def make_batch(wm, a, rank, dev):
    """Waveforms -> device data-prep STFT (row A1) -> [B,2,128,128] tiles; payload bits (audio_uformer_stft.py:231) or, for
    config 5, 64x64 grayscale images cut into quadrants (row A14)."""
    import torch
    g = torch.Generator().manual_seed(42 + rank)
    if a.clip_seconds == 10:
        if a.batch % 10:
            raise SystemExit("--clip-seconds 10 needs --batch to be a multiple of 10 tiles")
        from synth import chirp_wave
        clips = a.batch // 10
        waves = torch.stack([chirp_wave(10 * 16384 - 128, f"bench10_{rank}_{i}") for i in range(clips)]).to(dev)
        tiles, n_per = wm.dataprep.stft_tiles(waves)
        # a clip of 128 * k frames gets one extra all-zero tile from the reference's padding rule (audio_test.py:476-477)
        x = tiles.view(clips, n_per, 2, 128, 128)[:, :10].reshape(a.batch, 2, 128, 128).contiguous()
        img = torch.rand(clips, 1, 64, 64, generator=g).to(dev)                  # grayscale payload in [0, 1]
        return x, wm.dataprep.payload_for_tiles(img, 10).contiguous()
    msg = torch.randint(0, 2, (a.batch, 1, 32, 32), generator=g).float().to(dev)
    tiles, n_per = wm.dataprep.stft_tiles(synth_waves(a.batch, rank).to(dev))
    return tiles.view(a.batch, n_per, 2, 128, 128)[:, 0].contiguous(), msg


def workload_name(a, attacked):
    head = ("uformerWM Uformer-STFT, 10 s 16 kHz clips (10 tiles each), 64x64 grayscale payload as 4 quadrants, "
            if a.clip_seconds == 10 else "uformerWM Uformer-STFT, 1 s 16 kHz clips, 32x32 binary image payload, ")
    att = ("full audio_attack stack in the loop (resample 16k-8k-16k [Kaiser polyphase: librosa kernel parity unpinned] + AWGN 20 dB "
           "+ Butterworth-8 low-pass + amp x0.8), extractor on the attacked spectrogram" if attacked else "Identity attack")
    return head + att + f", {a.mode}"


# ---------------------------------------------------------------------------------------------------- CPU legs
def usable_cores():
    """(physical cores this process can actually run on, what bounds them): affinity mask, cgroup v2 / v1 CPU quota, SMT."""
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    phys, why = max(1, min(allowed, (os.cpu_count() or 2) // 2)), "affinity mask / SMT pairs"
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()) if q > 0 else None
        except (OSError, ValueError):
            pass
    if quota is not None and quota < phys:
        phys, why = max(1, int(quota)), f"cgroup CPU quota {quota:g}"
    return phys, why


def cpu_baseline(attacked, batch=4, steps=8, threads=None, budget_s=None):
    """The oracle (CPU restatement, parity-pinned to the reference) timed on this box's host cores: same step definition
    (zero_grad -> fwd -> attacks -> loss -> bwd -> AdamW), fp32, train mode.  Bounded sample (8 timed steps of 4 tiles: 12-14 s of
    CPU work at the measured 1.5 s per step) so the run stays in minutes.
    threads: None = the 16 threads of the 1-GPU box's CPU share (the primary leg); a number = that many (the all-cores leg)."""
    import torch
    from oracle import uformer_wm as O
    from oracle import attacks_torch as AT
    from synth import load_synth
    torch.manual_seed(0)
    torch.set_num_threads(min(16, os.cpu_count() or 1) if threads is None else int(threads))
    net = load_synth(O.UformerAudio()).train()
    opt = torch.optim.AdamW(net.parameters(), lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)
    waves = synth_waves(batch, 0)
    st = torch.stft(waves, n_fft=256, hop_length=128, win_length=256, window=torch.ones(256), return_complex=True)
    x = torch.view_as_real(st)[:, :-1].permute(0, 3, 1, 2)                       # audio_test.py:465-469
    x = torch.nn.functional.pad(x, (0, 128 - x.shape[3])).contiguous()
    msg = torch.randint(0, 2, (batch, 1, 32, 32), generator=torch.Generator().manual_seed(42)).float()
    gen = torch.Generator().manual_seed(1)
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        opt.zero_grad()
        att = None
        if attacked:
            z = torch.randn(batch, 8002, generator=gen)
            att = lambda w: AT.attack_chain(w, z=z, **FULL_STACK)
        out = net(x, msg, keeps=O.draw_keeps(net, batch, gen), attack=att, extract_from="attacked" if attacked else "y")
        loss, _ = O.train_losses(out, x, msg)
        loss.backward()
        opt.step()
        dt = time.perf_counter() - t0
        if it:
            times.append(dt)
        elif budget_s is not None and dt > budget_s:             # bounded sample: the warm-up step alone blew the budget
            times.append(dt)
            steps = 0
            break
    med = statistics.median(times)
    return {"value": batch * TILE_SECONDS / med, "unit": "audio-sec/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": (f"{steps} timed steps (1 warm-up) of B={batch} tiles, " if steps else
                       f"the single first step (no warm-up: it exceeded the {budget_s:.0f} s budget) of B={batch} tiles, ") +
                                      f"{'full attack stack' if attacked else 'Identity attack'}, fp32, train mode, AdamW; "
                                      f"median {med:.3f} s/step", "host_cpus": os.cpu_count()}


def cpu_baseline_parity(wm, net, x, msg, tiles=4):
    """Second part of the cpu_baseline leg — the oracle as the CHECKER, not as anything measured or shipped.
    Decoder BER parity (second half of BASELINE's metric): BER = mean(|round(clip(wm, 0, 1)) - bits|)
    (hidden/test_model.py:60-64) of the extractor output without attack and after the full attack stack, for the HIP path
    in the benched mode and in fp32 parity mode, against the fp32 CPU oracle on identical weights and inputs; the AWGN
    realisation of the HIP kernels (wm_awgn_noise) is replayed through the oracle's attack chain.  Eval mode, `tiles`
    tiles of the bench batch; max relative deviation (max |a - b| / max |b|) of the output tensors and loss terms."""
    import numpy as np
    import torch
    from oracle import uformer_wm as O
    from oracle import attacks as OA
    from synth import load_synth
    x, msg = x[:tiles].contiguous(), msg[:tiles].contiguous()
    xc, mc = x.cpu(), msg.cpu()
    bits = mc.round().clamp(0, 1)
    cfg = wm.AttackConfig(**FULL_STACK)
    was_training, mode0, att0, ext0 = net.training, net.compute_mode, net.attack, net.extract_from
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
        net.train(was_training)
    ref = load_synth(O.UformerAudio()).eval()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cpu = {}
    with torch.no_grad():
        cpu["no_attack"] = ref(xc, mc)
        attack = lambda w: torch.from_numpy(OA.attack_chain(w.numpy(), z=z, **FULL_STACK)).float()
        cpu["attacked"] = ref(xc, mc, attack=attack, extract_from="attacked")

    def ber(wm_out):
        return float((wm_out.round().clamp(0, 1) - bits).abs().mean())

    def rel(p, q):
        return float((p.double() - q.double()).abs().max() / q.double().abs().max().clamp_min(1e-30))
    out = {"tiles": tiles, "formula": "mean(|round(clip(wm,0,1)) - bits|)  (hidden/test_model.py:60-64)",
           "weights": "synthetic name-keyed init (tests/synth.py), identical on both sides; untrained, so BER sits near 0.5 "
                      "and what is compared is GPU vs CPU", "awgn": "HIP Philox stream replayed through the CPU oracle (wm_awgn_noise)"}
    for name in ("no_attack", "attacked"):
        c = cpu[name]
        e = {"cpu_fp32": ber(c[3])}
        for mode in dict.fromkeys((mode0, "fp32", "split")):
            h = hip[mode, name]
            agree = float((h[3].round().clamp(0, 1) == c[3].round().clamp(0, 1)).float().mean())
            lh, lc = O.train_losses(h, xc, mc)[1], O.train_losses(c, xc, mc)[1]
            e[f"hip_{mode}"] = ber(h[3])
            e[f"hip_{mode}_bits_equal_cpu"] = agree
            e[f"hip_{mode}_max_rel_dev"] = {"stft_new": rel(h[0], c[0]), "noise": rel(h[1], c[1]), "wm_pred": rel(h[2], c[2]),
                                            "wm": rel(h[3], c[3]),
                                            "loss_terms": [abs(float(p) - float(q)) / max(abs(float(q)), 1e-30) for p, q in zip(lh, lc)]}
        out[name] = e
    return out


# ---------------------------------------------------------------------------------------------------- main
def main():
    a = parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a))
    if a.clip_seconds == 10 and a.batch % 10:                    # (every rank checks its own arguments, before it builds anything)
        raise SystemExit("--clip-seconds 10 needs --batch to be a multiple of 10 tiles")
    # ONE JSON line on stdout and nothing else: libraries write to file descriptor 1 behind Python's back (RCCL prints its version
    # block there when the first communicator is built — seen in the one-rank rehearsal on the box), and under torch.distributed.run
    # rank 0's descriptor IS the job's stdout.  Everything this process writes to fd 1 goes to stderr; the line goes to the saved fd.
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {a.gpus} bench.py --gpus {a.gpus} ...)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the watermark hot path has no CPU fallback")
    if a.rehearse_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if a.rehearse_rccl:
        if world != 1:
            raise SystemExit("bench.py --rehearse-rccl is the ONE-rank rehearsal (--gpus 1)")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        a.no_reference_batch = a.no_identity_leg = a.no_parity_mode_leg = a.no_secondary_legs = True
        a.no_cpu_baseline = a.no_kernel_timing = True
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.rehearse_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    wm = importlib.import_module("image-in-speech-watermarking_amd")
    from synth import load_synth

    net = wm.UformerAudio(img_size=128, embed_dim=32, win_size=8, token_projection="linear", token_mlp="leff",
                          depths=[1, 2, 8, 8, 2, 8, 8, 2, 1], modulator=True, dd_in=2, in_chans=2, audio_scale="0",
                          compute_mode=a.mode)
    load_synth(net)                                              # identical random-init weights on every rank
    net = net.to(dev).train()
    opt = wm.FusedAdamW(net, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)
    attacked = a.attack == "full"
    attack = wm.AttackConfig(**FULL_STACK) if attacked else None
    extract_from = "attacked" if attacked else "y"
    x, msg = make_batch(wm, a, rank, dev)

    ber = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.clip_seconds == 1:
        ber = cpu_baseline_parity(wm, net, x, msg)               # before any optimiser step: weights = the synthetic init

    step = wm.TrainStep(net, opt, a.batch, attack=attack, extract_from=extract_from, use_graph=not a.no_graph,
                        segmented=a.segmented, grad_exchange=a.grad_exchange, loopback=a.rehearse_rccl)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(st, xs, ms, steps, warm):
        for _ in range(warm):
            st.step(xs, ms)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            st.step(xs, ms)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt

    dt = timed(step, x, msg, a.steps, max(a.warmup, 2))           # >= 2: first call warms up + captures the graph
    losses = step.losses.cpu().tolist()
    used_graph, segmented = step.use_graph, step.segmented
    n_graphs = len(step._segments[0]) if step._segments else 1
    allreduce = None
    if world > 1 or a.rehearse_rccl:
        # per-bucket gradient all-reduce time on the side stream (5 more steps; events on the communication stream) and the time
        # the host needs to issue one step (its graph replays and the collectives between them): the chain only keeps the GPU
        # busy while that stays well below the step itself (config 5: 26 ms per step)
        step.sync.timing = []
        barrier()
        step.host_timing = []
        for _ in range(5):
            step.step(x, msg)
        host = list(step.host_timing)
        step.host_timing = None
        barrier()
        per = {}
        for lo, hi, s, e in step.sync.timing:
            per.setdefault((lo, hi), []).append(s.elapsed_time(e))
        step.sync.timing = None
        # bus bandwidth of a ring all-reduce: 2 (N - 1) / N x bytes / time; one xGMI link carries ~153 GB/s per direction pair
        # (MI355X_MICROARCH / SURVEY 5), so a ring is bound by that per link
        wire = 2 if a.grad_exchange == "bf16" else 4              # bytes per gradient element on the links
        allreduce = {"exchange": a.grad_exchange + (" (all-gather of the ranks' bf16-rounded buckets, fp32 sum on arrival)" if wire == 2 else " all-reduce"),
                     "buckets_in_launch_order": [
            {"MB": round((hi - lo) * 4 / 1e6, 1), "ms": round(statistics.median(t), 3),
             "bus_GBps": round((1 if wire == 2 else 2) * (world - 1) / world * (hi - lo) * wire / (statistics.median(t) * 1e-3) / 1e9, 1)}
            for (lo, hi), t in per.items()],
            "xgmi_link_GBps": 153, "bucket_target_MB": round(step.sync.bucket_elems * 4 / 2 ** 20, 1),
            "exposed_tail_MB": round(sum((hi - lo) for (lo, hi) in list(per)[-2:]) * 4 / 1e6, 1)}
        if host:
            # the first of the five starts on an idle queue; later ones include any wait for queue space behind the running step
            allreduce["host_issue"] = {"ms_per_step_first": round(host[0] * 1e3, 3), "ms_per_step_median": round(statistics.median(host) * 1e3, 3),
                                       "graphs": n_graphs, "collectives": len(per),
                                       "frac_of_step_first": round(host[0] / (dt / a.steps), 4)}
        if a.rehearse_rccl:
            allreduce["rehearsal"] = "ONE rank on RCCL (loopback): call sequence and host issue time only, the bus figures mean nothing"
    # the captured step holds 0.56 GB per tile; the legs below build their own steps at the same batch: give the memory back first
    step.release()
    if a.rehearse_rccl:
        # the same number of steps as ONE captured graph on a fresh copy of the model (same init, same Philox keys and step
        # counter): the chain with the collectives between its graphs must leave the same loss terms (one-rank SUM = identity)
        net1 = wm.UformerAudio(img_size=128, embed_dim=32, win_size=8, token_projection="linear", token_mlp="leff",
                               depths=[1, 2, 8, 8, 2, 8, 8, 2, 1], modulator=True, dd_in=2, in_chans=2, audio_scale="0",
                               compute_mode=a.mode)
        load_synth(net1)
        net1 = net1.to(dev).train()
        opt1 = wm.FusedAdamW(net1, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)
        step1 = wm.TrainStep(net1, opt1, a.batch, attack=attack, extract_from=extract_from, use_graph=not a.no_graph)
        timed(step1, x, msg, a.steps, max(a.warmup, 2))
        allreduce["single_graph_loss_terms"] = step1.losses.cpu().tolist()[:5]
        step1.release()
        del step1, opt1, net1
    ref_batch = identity = None
    if a.batch != 24 and not a.no_reference_batch and a.clip_seconds == 1:
        # the same step at the batch the reference script hard-codes (audio_uformer_stft.py:50), timed the same way
        step24 = wm.TrainStep(net, opt, 24, attack=attack, extract_from=extract_from, use_graph=not a.no_graph)
        dt24 = timed(step24, x[:24], msg[:24], 10, 3)
        ref_batch = {"tiles_per_gpu": 24, "value": world * 24 * TILE_SECONDS * 10 / dt24, "ms_per_step": dt24 / 10 * 1e3, "steps": 10}
        step24.release()
        del step24
    if attacked and not a.no_identity_leg and a.clip_seconds == 1:
        # BASELINE config 2 (Identity attack) at the same batch, as a secondary field
        step_id = wm.TrainStep(net, opt, a.batch, attack=None, extract_from="y", use_graph=not a.no_graph)
        dti = timed(step_id, x, msg, 10, 3)
        identity = {"workload": workload_name(a, False), "value": world * a.batch * TILE_SECONDS * 10 / dti,
                    "ms_per_step": dti / 10 * 1e3, "steps": 10}
        step_id.release()
        del step_id

    parity_leg = None
    if a.mode == "bf16" and not a.no_parity_mode_leg and a.clip_seconds == 1 and world == 1:
        # the mode that meets north_star's tolerance (1e-4, payload bits exact: `ber` block, tests/test_model_gpu.py PARITY_MODES),
        # timed the same way at a batch its fp32 storage fits comfortably
        pb = min(a.batch, 96)
        net.compute_mode = "split"
        try:
            step_p = wm.TrainStep(net, opt, pb, attack=attack, extract_from=extract_from, use_graph=not a.no_graph)
            dtp = timed(step_p, x[:pb], msg[:pb], 8, 3)
            parity_leg = {"mode": "split (fp32 storage, GEMMs as three bf16 MFMA products of hi / lo operand halves, fp32 accumulate)",
                          "tiles_per_gpu": pb, "value": world * pb * TILE_SECONDS * 8 / dtp, "ms_per_step": dtp / 8 * 1e3, "steps": 8}
            step_p.release()
            del step_p
        finally:
            net.compute_mode = a.mode
    secondary = None
    if a.mode == "bf16" and not a.no_secondary_legs and a.clip_seconds == 1 and world == 1 and attacked:
        # the rates DESIGN.md quoted in prose, measured in this run — BASELINE config 5's per-GPU workload
        # (2 clips of 10 s = 20 tiles, 64 x 64 grayscale payload as quadrants) and the exact fp32 mode at B = 12
        import copy
        secondary = {}
        a5 = copy.copy(a)
        a5.clip_seconds, a5.batch = 10, 20
        x5, m5 = make_batch(wm, a5, rank, dev)
        st5 = wm.TrainStep(net, opt, 20, attack=attack, extract_from=extract_from, use_graph=not a.no_graph)
        dt5 = timed(st5, x5, m5, 10, 3)
        secondary["config5"] = {"workload": workload_name(a5, attacked), "tiles_per_gpu": 20,
                                "value": 20 * TILE_SECONDS * 10 / dt5, "ms_per_step": dt5 / 10 * 1e3, "steps": 10}
        st5.release()
        del st5, x5, m5
        net.compute_mode = "fp32"
        try:
            fb = min(12, a.batch)
            stf = wm.TrainStep(net, opt, fb, attack=attack, extract_from=extract_from, use_graph=not a.no_graph)
            dtf = timed(stf, x[:fb], msg[:fb], 5, 2)
            secondary["fp32_mode"] = {"mode": "fp32 (fp32 storage, fp32-input MFMA / VALU arithmetic throughout)", "tiles_per_gpu": fb,
                                      "value": fb * TILE_SECONDS * 5 / dtf, "ms_per_step": dtf / 5 * 1e3, "steps": 5}
            stf.release()
            del stf
        finally:
            net.compute_mode = a.mode
    result = {
        "metric": "audio-sec/s per train step (enc+attack+dec)", "value": world * a.batch * TILE_SECONDS * a.steps / dt,
        "unit": "audio-sec/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.mode, "data": "synthetic",
        "config": {"workload": workload_name(a, attacked),
                   "baseline_config": 5 if a.clip_seconds == 10 else (3 if attacked else 2),
                   "tiles_per_gpu": a.batch, "global_tiles": world * a.batch, "audio_sec_per_tile": TILE_SECONDS,
                   "parallelism": f"dp{world}", "graph": used_graph,
                   "graph_segments": (n_graphs if segmented else 1) if used_graph else 0, "params_M": 68.71},
        "loss_terms": losses[:5],
    }
    if ref_batch is not None:
        result["reference_batch"] = ref_batch
    if parity_leg is not None:
        result["parity_mode"] = parity_leg
    if identity is not None:
        result["identity_attack"] = identity
    if secondary:
        result.update(secondary)
    if ber is not None:
        result["ber"] = ber
    if allreduce is not None:
        result["allreduce"] = allreduce
    if rank == 0 and world == 1 and not a.no_kernel_timing:
        kernel_rooflines(a, wm, net, opt, attack, extract_from, x, msg, result)
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.ber_train_steps > 0 and a.clip_seconds == 1:
        result["ber_trained"] = trained_ber(wm, a)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(attacked)
        result["speedup_vs_cpu_baseline"] = result["value"] / result["cpu_baseline"]["value"]
        # "all physical cores".  Second leg at the reference script's batch (24) on every physical core this
        # process may USE: the affinity mask and the cgroup CPU quota both bound it (a 1-GPU box shows the host's 256 CPUs
        # but grants a share of 16; 128 threads on that share measured 52.7 s/step, 18 x slower per tile than 16 threads)
        phys, why = usable_cores()
        if phys > result["cpu_baseline"]["cores"]:
            # budget of the warm-up step: 3 x what the primary leg's rate predicts for 24 tiles
            leg = cpu_baseline(attacked, batch=24, steps=2, threads=phys, budget_s=3.0 * 24 * TILE_SECONDS / result["cpu_baseline"]["value"])
            if leg["value"] < result["cpu_baseline"]["value"]:
                leg["note"] = "slower than the primary leg: this process is confined to fewer CPUs than it can see"
            result["cpu_baseline_all_cores"] = leg
        else:
            result["cpu_baseline_all_cores"] = {"skipped": f"{why}: the primary leg already runs on all {phys} usable cores"}
    if rank == 0:
        sys.stdout.flush()
        os.write(line_fd, (json.dumps(result) + "\n").encode())
    os.close(line_fd)
    if world > 1 or a.rehearse_rccl:
        dist.destroy_process_group()


def trained_ber(wm, a):
    """Matched decoder BER on weights that actually decode (the `ber` block above is the untrained synthetic init, BER 0.5 on
    both sides): tests/ber_train.py trains a fresh model with the HIP TrainStep in the benched mode, the fp32 CPU oracle gets
    the same weights and both decode held-out (tile, payload) pairs.  Weights live in this process only."""
    import ber_train
    t0 = time.perf_counter()
    net, curve, xh, mh = ber_train.train(wm, a.ber_train_steps, batch=32, mode=a.mode, lr=2e-4, eval_every=max(a.ber_train_steps // 10, 1))
    t1 = time.perf_counter()
    out = {"trained": f"{a.ber_train_steps} captured steps of B = 32, config 3 (full attack stack, extractor on the attacked spectrogram), "
                      f"{a.mode}, AdamW lr 2e-4, synthetic chirp tiles scaled 0.05, payloads = 8 x 8 random bits as 4 x 4 blocks",
           # the configs' own payload (BinaryWM: 32 x 32 independent random bits, audio_uformer_stft.py:222-244) does not converge in a
           # benchmark's budget: 14 000 such steps (profiles/r04_ber_binary_payload_14000steps.json, 8 min) leave chance at step
           # ~5 000 and level off at BER 0.36 attacked / 0.37 clean — the payload autoencoder squeezes the image through 4 x 8 x 8 = 256
           # values (model.py ConvAutoencoder), which MNIST digits survive and 1 024 independent bits do not.  The block payload (64
           # bits per image) is the leg that reaches a decodable state
           "binary_payload_1024_bits": "see profiles/r04_ber_binary_payload_14000steps.json: BER 0.50 -> 0.36 (attacked) after 14 000 steps, plateau",
           "train_seconds": round(t1 - t0, 1),
           "ber_curve_held_out": [{k: (round(v, 5) if isinstance(v, float) else v) for k, v in e.items() if k != "loss_terms"} for e in curve],
           "oracle_parity": ber_train.oracle_parity(wm, net, xh, mh, tiles=8)}
    out["oracle_seconds"] = round(time.perf_counter() - t1, 1)
    return out


def kernel_rooflines(a, wm, net, opt, attack, extract_from, x, msg, result):
    """Per-kernel-family timing of an un-captured step with HIP events on the launch stream -> roofline objects."""
    import torch
    prof = importlib.import_module("image-in-speech-watermarking_amd.profiling")
    eager = wm.TrainStep(net, opt, a.batch, attack=attack, extract_from=extract_from, use_graph=False)
    eager.step(x, msg)
    torch.cuda.synchronize()

    def timed_step(detail=False):
        # Park the GPU behind a spin kernel while the host enqueues the whole step: otherwise every event pair
        # also measures the host's launch latency (eager launches are host-bound for kernels under ~50 us).
        torch.cuda._sleep(int(0.6e9))
        with prof.KernelTimer(detail=detail) as kt_:
            eager.step(x, msg)
        return kt_.summary()
    rows = timed_step()
    total = sum(r["total_s"] for r in rows)
    result["roofline"] = prof.roofline_of(rows[0])
    result["roofline"]["share_of_step_kernel_time"] = rows[0]["total_s"] / total
    # HBM traffic per launch and MFMA-busy counters come from rocprofv3 PMC passes of the same workload
    # (scripts/gpu_profile.sh; bench.py cannot run the profiler on itself).  A summary is used only when it was taken on the
    # kernel sources of THIS build (source id) at this batch / mode / attack: otherwise null.
    pmc = prof.find_pmc_summary(os.path.join(ROOT, "profiles"), a.batch, a.mode, a.attack)
    result["roofline"]["traffic"] = prof.pmc_traffic(rows[0]["kernel"], pmc)
    result["roofline"]["traffic_source"] = os.path.relpath(pmc, ROOT) if pmc else None
    # whole-step HBM traffic per tile from the same PMC summary (the number the fusion work is judged by)
    result["hbm_traffic"] = prof.pmc_step_traffic(pmc, a.batch)
    result["roofline_by_kernel"] = [
        {k: (round(v, 4) if isinstance(v, float) else v) for k, v in prof.roofline_of(r).items()
         if k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "launches", "avg_launch_us")}
        for r in rows[:8]]
    stft_rows = [r for r in rows if r["kernel"] in prof.STFT_PATH]
    if stft_rows:
        # these six launches last 10-50 us: re-timed as graph replays (an event pair around one such launch reads high)
        per = prof.time_stft_path(wm, a.batch, attack)
        by = sum(r["bytes"] for r in stft_rows)
        ts = sum(per[r["kernel"]] * r["launches"] for r in stft_rows)
        tr = [prof.pmc_traffic(r["kernel"], pmc) for r in stft_rows]
        result["roofline_stft_path"] = {"bound": "hbm", "achieved": by / ts / 1e9, "peak": prof.HBM_PEAK_GBS,
                                        "unit": "GB/s", "frac": by / ts / 1e9 / prof.HBM_PEAK_GBS,
                                        "traffic": sum(t * r["launches"] for t, r in zip(tr, stft_rows)) if all(t is not None for t in tr) else None,
                                        "alg_bytes_per_step": by, "time_us_per_step": ts * 1e6,
                                        # the fused ideal moves 652 320 B per tile and step (attacks in registers);
                                        # `frac` above books the attack kernels' own waveform round trips as algorithmic
                                        "frac_vs_fused_ideal": 652320.0 * a.batch / ts / 1e9 / prof.HBM_PEAK_GBS,
                                        "kernels_us": {r["kernel"]: round(per[r["kernel"]] * 1e6, 1) for r in stft_rows},
                                        "kernels_us_eager_event_pairs": {r["kernel"]: round(r["avg_us"], 1) for r in stft_rows}}
    mf = [r for r in rows if r["kernel"].startswith(("gemm_", "winattn", "attn_half"))]
    fl, ts = sum(r["flops"] for r in mf), sum(r["total_s"] for r in mf)
    peak = prof.MFMA_PEAK_TFLOPS[1 if a.mode == "bf16" else 0]
    result["mfma_path"] = {"achieved_TFLOPs": fl / ts / 1e12, "time_ms_per_step": ts * 1e3, "alg_TFLOP_per_step": fl / 1e12,
                           "frac_of_dense_peak_in_kernels": fl / ts / 1e12 / peak,
                           # 161.3 GFLOP per tile and step over the whole (graph-replayed) step time
                           "step_TFLOPs": 161.3e9 * a.batch / (result["ms_per_step"] * 1e-3) / 1e12,
                           "step_frac_of_dense_peak": 161.3e9 * a.batch / (result["ms_per_step"] * 1e-3) / 1e12 / peak,
                           "util": prof.pmc_mfma_util(pmc)}
    if a.detail:
        for r in timed_step(detail=True)[:60]:
            fl = r["flops"] / max(r["total_s"], 1e-12) / 1e12
            bw = r["bytes"] / max(r["total_s"], 1e-12) / 1e9
            print(f"{r['kernel']:60s} n={r['launches']:4d} total={r['total_s']*1e3:8.3f} ms avg={r['avg_us']:9.1f} us "
                  f"{fl:8.1f} TF/s {bw:8.1f} GB/s", file=sys.stderr)
    result["kernel_breakdown"] = [{"kernel": r["kernel"], "launches": r["launches"], "ms": round(r["total_s"] * 1e3, 3),
                                   "avg_us": round(r["avg_us"], 1)} for r in rows[:12]]


if __name__ == "__main__":
    main()
