"""Rehearsal of the multi-rank step on a one-GPU box: a 1-rank RCCL process group, the step captured as the chain of
graphs, and the ~25 MB bucket all-reduces really issued on the side stream between the replays (GradSync is told it has
two ranks; a 1-rank SUM leaves the gradients unchanged, so the losses must follow the single-graph run).

    python scripts/segmented_rccl_check.py
"""
import importlib
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
wm = importlib.import_module("image-in-speech-watermarking_amd")
from synth import load_synth, synth_bits, synth_tile  # noqa: E402


def make(B, **kw):
    m = wm.UformerAudio(modulator=True, drop_path_rate=0.1, compute_mode="bf16")
    load_synth(m)
    m = m.cuda().train()
    o = wm.FusedAdamW(m, lr=2e-4)
    return wm.TrainStep(m, o, B, **kw)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    B = 24
    x, msg = synth_tile(B, "rc_x").cuda(), synth_bits("rc_msg", (B, 1, 32, 32)).cuda()
    ref = make(B, use_graph=True)
    seg = make(B, use_graph=True, segmented=True)
    seg.sync.world = 2                                           # issue the collectives (SUM over one rank = identity)
    seg.sync._comm = torch.cuda.Stream()
    calls = []
    orig = dist.all_reduce
    dist.all_reduce = lambda t, **kw: (calls.append(t.numel()), orig(t, **kw))[1]
    for i in range(6):
        ref.step(x, msg)
        seg.step(x, msg)
        a, b = ref.losses.cpu(), seg.losses.cpu()
        assert torch.allclose(a[:5], b[:5], rtol=5e-3, atol=1e-5), (i, a, b)
    per_step = sum(len(sends) for sends in seg._segments[1])
    assert per_step >= 10 and len(calls) == per_step * 5, (per_step, len(calls))   # first call = warm-up + capture, then 5 replays
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        seg.step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(10):
        ref.step()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"segmented+rccl {100 * (t1 - t0):.2f} ms/step, single graph {100 * (t2 - t1):.2f} ms/step, "
          f"{per_step} all-reduces per step (MB: {[round(c * 4 / 1e6, 1) for c in calls[:per_step]]}), "
          f"{len(seg._segments[0])} graphs, losses {seg.losses.cpu()[:5].tolist()}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
