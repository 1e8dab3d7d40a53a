"""Two data-parallel ranks on ONE GPU (gloo moves the CUDA gradient buckets; RCCL refuses two ranks on one device):
the multi-rank step — warm-up, chain of four graphs, collectives between the replays, AdamW with the 1/world scale —
run for real.  Checks that the replicas stay bit-identical, that the gradients were averaged (a run with the other
rank's data alone ends elsewhere) and reports the step time.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 \
        scripts/two_rank_check.py
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


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B = 8
    m = wm.UformerAudio(modulator=True, drop_path_rate=0.1, compute_mode="bf16")
    load_synth(m)
    m = m.cuda().train()
    o = wm.FusedAdamW(m, lr=2e-4)
    step = wm.TrainStep(m, o, B)
    assert step.world == world and step.segmented and step.use_graph
    x, msg = synth_tile(B, f"tr_x{rank}").cuda(), synth_bits(f"tr_msg{rank}", (B, 1, 32, 32)).cuda()
    for _ in range(6):
        step.step(x, msg)
    torch.cuda.synchronize()
    assert step.use_graph and step._segments is not None and len(step._segments[0]) >= 10, "segmented capture not in use"
    n = m.optimized_numel
    mine = m._flat[:n].clone()
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    assert all(torch.equal(both[0], b) for b in both), "replicas diverged"
    assert torch.isfinite(step.losses).all() and int(o.step_t.item()) == 6
    t0 = time.perf_counter()
    for _ in range(10):
        step.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    if rank == 0:
        # the same six steps on rank 0's data alone must end somewhere else: the other rank's gradients were mixed in
        m1 = wm.UformerAudio(modulator=True, drop_path_rate=0.1, compute_mode="bf16")
        load_synth(m1)
        m1 = m1.cuda().train()
        o1 = wm.FusedAdamW(m1, lr=2e-4)
        solo = wm.TrainStep(m1, o1, B, process_group=None, segmented=True)
        solo.sync.world, solo.world = 1, 1
        solo.opt.grad_scale_t.fill_(1.0)
        for _ in range(6):
            solo.step(x, msg)
        torch.cuda.synchronize()
        d = (m1._flat[:n] - both[0]).abs().max().item()
        print(f"two ranks in sync after 6 steps ({len(step._segments[0])} graphs per step); |solo - dp| max {d:.3e}; "
              f"{1e3 * dt:.1f} ms/step with gloo on a shared GPU; losses {step.losses[:5].tolist()}")
        assert d > 1e-6
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
