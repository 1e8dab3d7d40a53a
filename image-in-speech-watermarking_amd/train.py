"""Fused optimizer and the whole train step (reference uformerWM/audio_uformer_stft.py:421-540).

``TrainStep`` = zero_grad -> forward -> attacks -> 4-term loss -> backward -> (gradient all-reduce) -> AdamW,
issued as one stream of HIP kernel launches with nothing returning to the host; on one GPU the sequence is
captured once into a hipGraph and replayed.  Data parallel: one process per GPU, the flat gradient arena is
all-reduced over RCCL in ~25 MB buckets of finished arena runs (GradSync) on a side stream, each launched as
soon as backward has produced it so the xGMI transfer overlaps the remaining backward; the captured step is
then a chain of graphs cut at the bucket launches.
"""
import time

import torch
import torch.distributed as dist

from . import ops


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (lr 2e-4, betas 0.9/0.999, eps 1e-8, wd 0.02 at audio_uformer_stft.py:252-255)
    applied to the model's flat parameter arena by ONE kernel.  lr / step / grad-scale live in device memory so a
    captured graph follows the LR schedule.

    It IS a ``torch.optim.Optimizer`` (one param group holding the model's Parameters, which are views of the arena), so the
    reference trainer's scheduler construction works unchanged (audio_uformer_stft.py:347-356): ``CosineAnnealingLR(opt, ...)``,
    the reference's ``GradualWarmupScheduler(opt, ...)`` (or ``train.GradualWarmupScheduler``, the same rule) and ``StepLR``
    write ``param_groups[0]['lr']``; ``step()`` / ``TrainStep.step()`` copy it to the device scalar the kernel reads
    (``sync_lr``), also between replays of a captured step."""

    def __init__(self, net, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02):
        self.net = net
        super().__init__(list(net.parameters()), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        dev = net._flat.device
        n = net.optimized_numel
        self.exp_avg = torch.zeros(n, device=dev)
        self.exp_avg_sq = torch.zeros(n, device=dev)
        self.step_t = torch.zeros(1, dtype=torch.int64, device=dev)
        self.lr_t = torch.full((1,), float(lr), device=dev)
        self.grad_scale_t = torch.ones(1, device=dev)
        self._lr_host = float(lr)

    def add_param_group(self, param_group):
        if getattr(self, "param_groups", None):
            raise ValueError("FusedAdamW updates the model's whole parameter arena with one set of hyper-parameters: one param group")
        super().add_param_group(param_group)

    def sync_lr(self):
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_host:
            self.lr_t.fill_(lr)
            self._lr_host = lr

    def zero_grad(self, set_to_none=False):
        """Zeroes the gradient arena in place (the Parameters' ``.grad`` stay views of it, whatever ``set_to_none`` says)."""
        self.net._flat_grad.zero_()

    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.sync_lr()
        self.launch()
        return loss

    def launch(self):
        """Graph-capturable part of step()."""
        g = self.param_groups[0]
        n = self.net.optimized_numel
        ops.step_inc(self.step_t)
        ops.adamw(self.net._flat, self.net._flat_grad, self.exp_avg, self.exp_avg_sq, n, self.lr_t, g["betas"][0],
                  g["betas"][1], g["eps"], g["weight_decay"], self.step_t, self.grad_scale_t)

    def state_dict(self):
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "step": self.step_t, "lr": self._lr_host}

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"]); self.step_t.copy_(sd["step"])
        self.param_groups[0]["lr"] = sd["lr"]
        self.sync_lr()


class GradualWarmupScheduler(torch.optim.lr_scheduler.LRScheduler):
    """Linear warm-up in front of another scheduler, the rule of the reference's ``warmup_scheduler/scheduler.py:5-63`` as the
    trainer uses it (audio_uformer_stft.py:347-356: ``multiplier=1, total_epoch=warmup_epochs, after_scheduler=cosine``):
    epoch e <= total_epoch: lr = base_lr * e / total_epoch (multiplier 1) or base_lr * ((multiplier - 1) e / total_epoch + 1);
    afterwards the wrapped scheduler runs from base_lr * multiplier, stepped once per call.  (The ReduceLROnPlateau branch of the
    reference class is not on the trainer's path and is left out.)"""

    def __init__(self, optimizer, multiplier, total_epoch, after_scheduler=None):
        if multiplier < 1.0:
            raise ValueError("multiplier should be greater than or equal to 1.")
        self.multiplier, self.total_epoch, self.after_scheduler = multiplier, total_epoch, after_scheduler
        self.finished = False
        super().__init__(optimizer)

    def get_lr(self):
        if self.last_epoch > self.total_epoch:
            if self.after_scheduler is not None:
                if not self.finished:
                    self.after_scheduler.base_lrs = [b * self.multiplier for b in self.base_lrs]
                    self.finished = True
                # the hand-over epoch asks the wrapped scheduler's get_lr() directly, as the reference class does: with a chained
                # CosineAnnealingLR that is its recursive form evaluated at ITS epoch 0 (tests/test_host_cpu.py holds the whole
                # sequence to the reference's class under the installed torch)
                return self.after_scheduler.get_lr()
            return [b * self.multiplier for b in self.base_lrs]
        if self.multiplier == 1.0:
            return [b * (float(self.last_epoch) / self.total_epoch) for b in self.base_lrs]
        return [b * ((self.multiplier - 1.0) * self.last_epoch / self.total_epoch + 1.0) for b in self.base_lrs]

    def step(self, epoch=None):
        if self.finished and self.after_scheduler is not None:
            self.after_scheduler.step(None if epoch is None else epoch - self.total_epoch)
            self._last_lr = self.after_scheduler.get_last_lr()
            return
        super().step(epoch)


class GradSync:
    """Data-parallel gradient exchange over the flat gradient arena (reference: none — the reference is single-device,
    SURVEY 2 / 8e).  Backward reports every finished arena run (a LeWin block, a down / up-sampling convolution: `ready`);
    adjacent runs are merged and a merged run is all-reduced (SUM; the 1/world mean is folded into AdamW's grad scale) as soon
    as it reaches ``bucket_bytes`` (~25 MB: SURVEY 8e), on a side stream over RCCL / xGMI, so that the transfers overlap the
    remaining backward and only the small remainder of the last segment is exposed in front of AdamW.  ``segment_done`` flushes
    what is left of a segment (its convolutions, a partial bucket).  With gloo / CPU tensors (tests) the collectives are issued
    synchronously.  The merge logic is deterministic in the call sequence, so a planning pass (``plan=True``: nothing is sent,
    `on_launch` is called where a collective would start) tells TrainStep where to cut its captured graphs."""

    def __init__(self, net, process_group=None, bucket_bytes=25 << 20, exchange="fp32"):
        """exchange: 'fp32' (all-reduce of the fp32 buckets, 275 MB per step and rank) or 'bf16': every rank rounds its bucket to
        bf16, the ranks' buckets are all-gathered and summed in fp32 on arrival, in rank order on every rank — a quarter of the
        bytes per xGMI link pair ((N - 1) / N x 2 B instead of 2 (N - 1) / N x 4 B per element), replicas stay bit-identical,
        and a gradient carries bf16 rounding once (2^-9 relative per rank's contribution, as the bf16 activations already do)."""
        if exchange not in ("fp32", "bf16"):
            raise ValueError("exchange is 'fp32' or 'bf16'")
        self.exchange = exchange
        self._xbuf = {}
        self.net, self.pg = net, process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.cuda = net._flat.is_cuda
        self._comm = torch.cuda.Stream(device=net._flat.device) if (self.cuda and self.world > 1) else None
        self.bucket_elems = max(int(bucket_bytes) // 4, 1)
        self.timing = None                                      # bench.py: list of (lo, hi, start event, stop event)
        self.force = False                                      # tests / rehearsals: run the bucket logic on one rank too
        self.loopback = False                                   # ... and issue its collectives on a one-rank group (enable_loopback)
        self.reset()

    def enable_loopback(self):
        """One-rank rehearsal of the N-rank call sequence: the bucket logic runs (``force``) and every collective is really
        issued, on the side stream, on the one-rank process group the caller has initialised (RCCL accepts a single rank; the
        sum over one rank is the identity, so the step's results do not change).  What it exercises on a one-GPU box is the
        RCCL launches between the graph replays, the stream hand-overs and the bf16 exchange's device path."""
        if not dist.is_initialized() or self.world != 1:
            raise RuntimeError("enable_loopback needs an initialised process group of exactly one rank")
        self.force = self.loopback = True
        if self.cuda and self._comm is None:
            self._comm = torch.cuda.Stream(device=self.net._flat.device)

    # -- bookkeeping ---------------------------------------------------------------------------------
    def reset(self):
        self._pending = []                                      # sorted, disjoint [lo, hi) runs that are final but not sent
        self._sent = []                                         # runs already handed to a collective in this step
        self._seg = 0

    def _active(self):
        return self.world > 1 or self.force

    def ready(self, lo, hi, plan=False, on_launch=None):
        if not self._active() or hi <= lo:
            return
        runs = sorted(self._pending + [(lo, hi)])
        merged = [runs[0]]
        for a, b in runs[1:]:
            if a <= merged[-1][1]:
                merged[-1] = (merged[-1][0], max(merged[-1][1], b))
            else:
                merged.append((a, b))
        self._pending = []
        for a, b in merged:
            if b - a >= self.bucket_elems:
                self._launch(a, b, plan, on_launch)
            else:
                self._pending.append((a, b))

    def segment_done(self, idx, plan=False, on_launch=None):
        """Everything of segment idx that has not been sent yet goes out now (in as few contiguous pieces as possible)."""
        if not self._active():
            return
        lo, hi = self.net.segment_bounds()[idx]
        sent = sorted((a, b) for a, b in self._sent if a < hi and b > lo)
        self._pending = [(a, b) for a, b in self._pending if not (a < hi and b > lo)]
        cur = lo
        for a, b in sent + [(hi, hi)]:
            if a > cur:
                self._launch(cur, a, plan, on_launch)
            cur = max(cur, b)

    def _launch(self, lo, hi, plan, on_launch):
        self._sent.append((lo, hi))
        if on_launch is not None:
            on_launch(lo, hi)
        if not plan:
            self.all_reduce(lo, hi)

    # -- the collective ------------------------------------------------------------------------------
    def _exchange(self, bucket):
        """SUM of `bucket` over the ranks, in place (on the current stream / synchronously on CPU tensors)."""
        if self.exchange == "fp32":
            dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.pg)
            return
        n = bucket.numel()
        npad = (n + 7) // 8 * 8                                  # (arena runs are multiples of 8 elements; a tail stays safe)
        key = (npad, bucket.device)
        if key not in self._xbuf:                                # per bucket size: the send copy and the gathered copies
            self._xbuf[key] = (torch.zeros(npad, dtype=torch.bfloat16, device=bucket.device),
                               torch.empty(self.world, npad, dtype=torch.bfloat16, device=bucket.device))
        send, gathered = self._xbuf[key]
        if bucket.is_cuda:
            ops.cast_bf16(bucket, send, n)
            dist.all_gather_into_tensor(gathered.view(-1), send, group=self.pg)    # (flat output: gloo accepts no [world, n] view)
            if npad == n:
                ops.sum_bf16_rows(gathered, bucket, self.world, n)
            else:
                tmp = torch.empty(npad, device=bucket.device)
                ops.sum_bf16_rows(gathered, tmp, self.world, npad)
                bucket.copy_(tmp[:n])
        else:                                                    # gloo / CPU tensors: the test path of the bucket logic
            send[:n].copy_(bucket)
            parts = [torch.empty_like(send) for _ in range(self.world)]
            dist.all_gather(parts, send, group=self.pg)
            bucket.copy_(sum(p.float() for p in parts)[:n])

    def all_reduce(self, lo, hi):
        bucket = self.net._flat_grad[lo:hi]
        if self.world == 1 and not self.loopback:
            return
        if self._comm is None:
            self._exchange(bucket)
            return
        self._comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._comm):
            if self.timing is not None:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
            self._exchange(bucket)
            if self.timing is not None:
                e.record()
                self.timing.append((lo, hi, s, e))

    # the three-segment interface of round 1 / 2 (segment i complete): now a flush
    def after_segment(self, idx):
        self.segment_done(idx)

    def finish(self):
        if self._comm is not None:
            torch.cuda.current_stream().wait_stream(self._comm)
        self.reset()


class TrainStep:
    """One optimisation step on a batch of (STFT tile, payload) pairs.

    step(x [B,2,128,128], message [B,1,32,32]) -> None; losses are left in ``self.losses`` (device tensor:
    l1, l2, l3, l4, total, ||noise||) so nothing forces a host sync.
    """

    def __init__(self, net, opt, batch, attack=None, extract_from="y", use_graph=True, process_group=None,
                 segmented=False, grad_exchange="fp32", loopback=False):
        self.net, self.opt, self.B = net, opt, int(batch)
        self.attack, self.extract_from = attack, extract_from
        dev = net._flat.device
        net._ensure_runtime(dev)
        self.x = torch.zeros(batch, 2, 128, 128, device=dev)
        self.msg = torch.zeros(batch, 1, 32, 32, device=dev)
        self.losses = torch.zeros(8, device=dev)
        self._scratch = torch.zeros(8, device=dev)
        self.sync = GradSync(net, process_group, exchange=grad_exchange)
        self.world = self.sync.world
        if self.world > 1:
            opt.grad_scale_t.fill_(1.0 / self.world)          # all-reduce(sum) then scale == mean
        self.use_graph = bool(use_graph)
        # with more than one rank the step is captured as a chain of graphs cut at the gradient-bucket boundaries: the
        # RCCL all-reduces stay outside the captures and are issued between the replays (``segmented`` forces the same
        # structure on one rank, for tests)
        self.segmented = bool(segmented) or self.world > 1
        if loopback:                                            # one-rank rehearsal: the chain of graphs with real collectives between them
            self.sync.enable_loopback()
            self.segmented = True
        self.host_timing = None                                 # bench.py: seconds the host spent issuing one step's replays + collectives
        self._graph = None
        self._segments = None
        self.outputs = None

    # -- the launch sequence (no host syncs, no python-side data dependence) ------------------------
    def _launch(self, after_segment=None, before_optimizer=None, grad_ready=None):
        net, B = self.net, self.B
        if after_segment is None and self.sync._active():
            after_segment, before_optimizer, grad_ready = self.sync.segment_done, self.sync.finish, self.sync.ready
            self.sync.reset()
        ops.zero_(net._flat_grad)
        ops.step_inc(net._step_t)                               # advances the DropPath / AWGN Philox streams
        outs, S = net._forward_impl(self.x, self.msg, train=net.training, keep_scales=None, attack=self.attack,
                                    extract_from=self.extract_from)
        d = [torch.empty_like(o) for o in outs]
        ops.loss_fwd_bwd(outs[0], self.x, outs[1], outs[2], outs[3], self.msg, self.losses, d[0], d[1], d[2], d[3], B,
                         self._scratch)
        net._backward_impl(S, d[0], d[1], d[2], d[3], after_segment=after_segment, grad_ready=grad_ready)
        if before_optimizer is not None:
            before_optimizer()
        self.opt.launch()
        self.outputs = outs
        self._tape = S if getattr(self, 'keep_tape', False) else None

    def step(self, x=None, message=None):
        if x is not None:
            self.x.copy_(x, non_blocking=True)
            self.msg.copy_(message, non_blocking=True)
        self.opt.sync_lr()
        if not self.use_graph:
            self._launch()
            return
        if self.segmented:
            if self._segments is None:
                self._warm_up()
                try:
                    self._capture_segments()
                except RuntimeError as e:                       # same HIP launches, just not captured: say so and go on
                    import sys
                    print(f"[TrainStep] segmented graph capture failed ({e}); running the step un-captured", file=sys.stderr)
                    torch.cuda.synchronize()
                    self.use_graph = False
                return                                          # as below: the warm-up pass was this call's step
            self._replay_segments()
            return
        if self._graph is None:
            self._warm_up()
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._launch()
            return                                              # the capture pass itself does not execute: replay next
        self._graph.replay()

    def release(self):
        """Drop the captured graph(s) and their memory pool (0.56 GB per tile) and hand the allocator's cache back; the next
        ``step`` captures again.  For callers that keep several TrainStep objects of bench-sized batches around."""
        import gc
        self._graph = None
        self._segments = None
        self.outputs = None
        self._tape = None
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()

    def _warm_up(self):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                              # outside capture (lazy inits, hipFuncSetAttribute)
            self._launch()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        # the capture below allocates the whole step again from the graph's private pool: hand the warm-up pass's cached blocks
        # (≈0.5 GB per tile at the end of the forward) back first instead of leaving it to the allocator's out-of-memory retry
        torch.cuda.empty_cache()

    def _capture_segments(self):
        """Capture the launch sequence as graphs cut wherever the gradient exchange starts collectives (a ~25 MB bucket is
        complete, or a segment ends).  Nothing executes during the pass and no collective is recorded: GradSync runs its bucket
        logic in planning mode and reports the launches; ``_replay_segments`` issues them after the graph they follow.  All
        graphs allocate from one pool, replayed in capture order, so tensors that live across a cut keep their addresses."""
        graphs, launches = [torch.cuda.CUDAGraph()], []
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        sync = self.sync
        force0, sync.force = sync.force, True                      # plan the cuts on one rank too (``segmented`` rehearsals)

        def planned(call, *args):
            sends = []
            call(*args, plan=True, on_launch=lambda lo, hi: sends.append((lo, hi)))
            if sends:                                              # the collectives of this point follow the graph that ends here
                graphs[-1].capture_end()
                launches.append(sends)
                graphs.append(torch.cuda.CUDAGraph())
                graphs[-1].capture_begin(pool=graphs[0].pool(), capture_error_mode="thread_local")

        sync.reset()
        try:
            with torch.cuda.stream(side):
                graphs[0].capture_begin(capture_error_mode="thread_local")
                try:
                    self._launch(after_segment=lambda idx: planned(sync.segment_done, idx), before_optimizer=lambda: None,
                                 grad_ready=lambda lo, hi: planned(sync.ready, lo, hi))
                except BaseException:
                    try:
                        graphs[-1].capture_end()                    # leave capture mode before reporting
                    except RuntimeError:
                        pass
                    raise
                graphs[-1].capture_end()
        finally:
            sync.force = force0
            sync.reset()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        launches.append([])                                        # the last graph (AdamW) is followed by nothing
        self._segments = (graphs, launches)

    def _replay_segments(self):
        graphs, launches = self._segments
        t0 = time.perf_counter() if self.host_timing is not None else 0.0
        for i, (g, sends) in enumerate(zip(graphs, launches)):
            if i == len(graphs) - 1:
                self.sync.finish()                              # the last graph (AdamW) waits for the collectives
            g.replay()
            for lo, hi in sends:
                self.sync.all_reduce(lo, hi)
        if self.host_timing is not None:
            self.host_timing.append(time.perf_counter() - t0)
