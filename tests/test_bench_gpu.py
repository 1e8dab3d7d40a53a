"""bench.py as the driver calls it: a plain `python bench.py --gpus N` starts the N ranks itself and prints rank 0's line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUICK = ["--steps", "2", "--warmup", "1", "--batch", "2", "--no-cpu-baseline", "--no-kernel-timing", "--no-reference-batch",
         "--no-identity-leg"]


def test_gpus_flag_without_devices_fails_loudly():
    """No GPU (or fewer than N): the launcher refuses before spawning anything; never a silent N = 1 run."""
    import torch
    if torch.cuda.device_count() >= 4:
        pytest.skip("a multi-GPU node would simply run it")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"] + QUICK, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "--gpus 4" in r.stderr and not r.stdout.strip()
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"] + QUICK, capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


@pytest.mark.gpu
def test_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2 --rehearse-one-gpu`: two rank processes share GPU 0 over gloo (RCCL refuses two ranks per
    device), the chain of graphs with the ~25 MB bucket all-reduces between them runs, the line says n_gpus 2."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-one-gpu"] + QUICK,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["global_tiles"] == 4 and line["config"]["graph_segments"] >= 10
    assert line["config"]["baseline_config"] == 3 and "full audio_attack stack" in line["config"]["workload"]
    ar = line["allreduce"]
    assert len(ar["buckets_in_launch_order"]) >= 10 and abs(sum(b["MB"] for b in ar["buckets_in_launch_order"]) - 274.9) < 1.0
    assert ar["exposed_tail_MB"] < 20 and all(b["ms"] > 0 for b in ar["buckets_in_launch_order"])
    assert "roofline" not in line and "cpu_baseline" not in line


@pytest.mark.gpu
def test_two_rank_rehearsal_of_config5_clips():
    """`--gpus 2 --clip-seconds 10` (config 5's shape: 10-tile clips, 64 x 64 payload as quadrants) over the N-rank path, with the
    bf16 gradient exchange (device path: wm_cast_bf16 -> all-gather -> wm_sum_bf16_rows; the other rehearsal uses the fp32 all-reduce)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-one-gpu", "--clip-seconds", "10",
                        "--steps", "2", "--warmup", "1", "--batch", "10", "--no-cpu-baseline", "--no-kernel-timing",
                        "--grad-exchange", "bf16"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["baseline_config"] == 5 and line["config"]["global_tiles"] == 20
    assert "10 s 16 kHz clips" in line["config"]["workload"] and line["value"] > 0
    assert line["allreduce"]["exchange"].startswith("bf16")


@pytest.mark.gpu
def test_one_rank_rccl_rehearsal_of_the_graph_chain():
    """`--rehearse-rccl`: the N-rank call sequence on a ONE-rank RCCL group at config 5's per-GPU shape (2 clips x 10 tiles, the
    26 ms step): every bucket's collective is really issued on the side stream between the graph replays (the gloo rehearsals
    above never touch RCCL).  The sum over one rank is the identity, so the loss terms equal the plain single-graph run's; the
    line reports how long the host needs to issue one step's replays and collectives."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rehearse-rccl", "--clip-seconds", "10", "--batch", "20",
                        "--steps", "6", "--warmup", "2"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    reh = json.loads(r.stdout.strip().splitlines()[-1])
    assert reh["n_gpus"] == 1 and reh["config"]["graph_segments"] >= 10 and reh["config"]["baseline_config"] == 5
    ar = reh["allreduce"]
    assert len(ar["buckets_in_launch_order"]) >= 10 and abs(sum(b["MB"] for b in ar["buckets_in_launch_order"]) - 274.9) < 1.0
    assert all(b["ms"] > 0 for b in ar["buckets_in_launch_order"]) and "ONE rank" in ar["rehearsal"]
    hi = ar["host_issue"]
    assert hi["graphs"] == reh["config"]["graph_segments"] and hi["collectives"] == len(ar["buckets_in_launch_order"])
    assert 0 < hi["ms_per_step_first"] < reh["ms_per_step"], hi          # the host stays ahead of the device at the 26 ms step
    for p_, q_ in zip(reh["loss_terms"], ar["single_graph_loss_terms"]):  # same steps, same streams: only the order of atomics differs
        assert abs(p_ - q_) <= 1e-2 * abs(q_) + 5e-4, (reh["loss_terms"], ar["single_graph_loss_terms"])   # (measured: 9e-5 on the 1e-2 term)
    assert r.stdout.strip().count("\n") == 0, r.stdout[:400]              # the line and nothing else (RCCL's version block goes to stderr)


def test_launcher_stops_when_a_rank_fails():
    """A rank that dies at start-up (here: --batch 7 with --clip-seconds 10 is rejected by every rank, before it builds anything —
    so the case needs no GPU) ends the launcher with that rank's message on stderr instead of leaving the others in a collective
    until the distributed timeout."""
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-one-gpu", "--clip-seconds", "10",
                        "--batch", "7", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timing"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "multiple of 10" in r.stderr and "stopping the other ranks" in r.stderr and time.time() - t0 < 120
    assert not r.stdout.strip()


@pytest.mark.gpu
def test_default_line_carries_config3_roofline_and_ber():
    # (--ber-train-steps 200: the key and its plumbing; the trained state itself is tests/test_trained_ber_gpu.py's 2000 steps)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "4",
                        "--no-reference-batch", "--no-identity-leg", "--ber-train-steps", "200"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["config"]["baseline_config"] == 3 and line["dtype"] == "bf16"
    for k in ("roofline", "cpu_baseline", "ber", "ber_trained", "mfma_path", "roofline_stft_path", "parity_mode", "config5", "fp32_mode"):
        assert k in line, k
    assert line["config5"]["tiles_per_gpu"] == 20 and "10 s 16 kHz clips" in line["config5"]["workload"] and line["config5"]["value"] > 0
    assert line["fp32_mode"]["value"] > 0 and line["parity_mode"]["value"] > 0
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    ber = line["ber"]
    assert ber["no_attack"]["hip_fp32_bits_equal_cpu"] == 1.0 and ber["attacked"]["hip_fp32_bits_equal_cpu"] == 1.0
    assert ber["no_attack"]["hip_bf16_bits_equal_cpu"] >= 0.99 and ber["attacked"]["hip_bf16_bits_equal_cpu"] >= 0.99
    for leg in ("no_attack", "attacked"):
        d = ber[leg]["hip_fp32_max_rel_dev"]
        assert max(d["stft_new"], d["noise"], d["wm"]) <= 2e-4, (leg, d)
