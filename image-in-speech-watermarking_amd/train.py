"""Fused optimizer and the whole train step (reference uformerWM/audio_uformer_stft.py:421-540).

``TrainStep`` = zero_grad -> forward -> attacks -> 4-term loss -> backward -> (gradient all-reduce) -> AdamW,
issued as one stream of HIP kernel launches with nothing returning to the host; on one GPU the sequence is
captured once into a hipGraph and replayed.  Data parallel: one process per GPU, the flat gradient arena is
all-reduced over RCCL in three contiguous slices (extractor / decoder / encoder) on a side stream, each
launched as soon as its part of backward has finished so the xGMI transfer overlaps the remaining backward.
"""
import torch
import torch.distributed as dist

from . import ops


class FusedAdamW:
    """torch.optim.AdamW semantics (lr 2e-4, betas 0.9/0.999, eps 1e-8, wd 0.02 at audio_uformer_stft.py:252-255)
    applied to the model's flat parameter arena by ONE kernel.  lr / step / grad-scale live in device memory so a
    captured graph follows the LR schedule."""

    def __init__(self, net, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02):
        self.net = net
        self.defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.param_groups = [dict(self.defaults, params=list(net.parameters()))]   # LR schedulers poke 'lr' here
        dev = net._flat.device
        n = net.optimized_numel
        self.exp_avg = torch.zeros(n, device=dev)
        self.exp_avg_sq = torch.zeros(n, device=dev)
        self.step_t = torch.zeros(1, dtype=torch.int64, device=dev)
        self.lr_t = torch.full((1,), float(lr), device=dev)
        self.grad_scale_t = torch.ones(1, device=dev)
        self._lr_host = float(lr)

    def sync_lr(self):
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_host:
            self.lr_t.fill_(lr)
            self._lr_host = lr

    def zero_grad(self, set_to_none=False):
        self.net._flat_grad.zero_()

    def step(self):
        self.sync_lr()
        self.launch()

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


class GradSync:
    """Data-parallel gradient exchange over the flat arena: three contiguous slices, each all-reduced (SUM; the
    1/world mean is folded into AdamW's grad scale) as soon as backward has finished it.  On a GPU the collective
    runs on a side stream (RCCL over xGMI) so it overlaps the remaining backward; with gloo/CPU tensors (tests)
    it is issued synchronously.  Reference: none — the reference is single-device."""

    def __init__(self, net, process_group=None):
        self.net, self.pg = net, process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.cuda = net._flat.is_cuda
        self._comm = torch.cuda.Stream(device=net._flat.device) if (self.cuda and self.world > 1) else None
        self.timing = None                                      # bench.py: list of (bucket, start event, stop event)

    def after_segment(self, idx):
        if self.world == 1:
            return
        bucket = self.net.grad_buckets()[idx]
        if self._comm is None:
            dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.pg)
            return
        self._comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._comm):
            if self.timing is not None:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
            dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.pg)
            if self.timing is not None:
                e.record()
                self.timing.append((idx, s, e))

    def finish(self):
        if self._comm is not None:
            torch.cuda.current_stream().wait_stream(self._comm)


class TrainStep:
    """One optimisation step on a batch of (STFT tile, payload) pairs.

    step(x [B,2,128,128], message [B,1,32,32]) -> None; losses are left in ``self.losses`` (device tensor:
    l1, l2, l3, l4, total, ||noise||) so nothing forces a host sync.
    """

    def __init__(self, net, opt, batch, attack=None, extract_from="y", use_graph=True, process_group=None,
                 segmented=False):
        self.net, self.opt, self.B = net, opt, int(batch)
        self.attack, self.extract_from = attack, extract_from
        dev = net._flat.device
        net._ensure_runtime(dev)
        self.x = torch.zeros(batch, 2, 128, 128, device=dev)
        self.msg = torch.zeros(batch, 1, 32, 32, device=dev)
        self.losses = torch.zeros(8, device=dev)
        self._scratch = torch.zeros(8, device=dev)
        self.sync = GradSync(net, process_group)
        self.world = self.sync.world
        if self.world > 1:
            opt.grad_scale_t.fill_(1.0 / self.world)          # all-reduce(sum) then scale == mean
        self.use_graph = bool(use_graph)
        # with more than one rank the step is captured as a chain of graphs cut at the gradient-bucket boundaries: the
        # RCCL all-reduces stay outside the captures and are issued between the replays (``segmented`` forces the same
        # structure on one rank, for tests)
        self.segmented = bool(segmented) or self.world > 1
        self._graph = None
        self._segments = None
        self.outputs = None

    # -- the launch sequence (no host syncs, no python-side data dependence) ------------------------
    def _launch(self, after_segment=None, before_optimizer=None):
        net, B = self.net, self.B
        if after_segment is None and self.world > 1:
            after_segment, before_optimizer = self.sync.after_segment, self.sync.finish
        ops.zero_(net._flat_grad)
        ops.step_inc(net._step_t)                               # advances the DropPath / AWGN Philox streams
        outs, S = net._forward_impl(self.x, self.msg, train=net.training, keep_scales=None, attack=self.attack,
                                    extract_from=self.extract_from)
        d = [torch.empty_like(o) for o in outs]
        ops.loss_fwd_bwd(outs[0], self.x, outs[1], outs[2], outs[3], self.msg, self.losses, d[0], d[1], d[2], d[3], B,
                         self._scratch)
        net._backward_impl(S, d[0], d[1], d[2], d[3], after_segment=after_segment)
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
        """Capture the launch sequence as graphs cut where backward has finished a gradient bucket.  Nothing executes
        during the pass and no collective is recorded: ``_replay_segments`` issues the all-reduce of bucket i after
        graph i.  All graphs allocate from one pool, replayed in capture order, so tensors that live across a cut keep
        their addresses."""
        graphs, cuts = [torch.cuda.CUDAGraph()], []
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())

        def cut(idx):
            graphs[-1].capture_end()
            cuts.append(idx)
            graphs.append(torch.cuda.CUDAGraph())
            graphs[-1].capture_begin(pool=graphs[0].pool(), capture_error_mode="thread_local")

        with torch.cuda.stream(side):
            graphs[0].capture_begin(capture_error_mode="thread_local")
            try:
                self._launch(after_segment=cut, before_optimizer=lambda: None)
            except BaseException:
                try:
                    graphs[-1].capture_end()                    # leave capture mode before reporting
                except RuntimeError:
                    pass
                raise
            graphs[-1].capture_end()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._segments = (graphs, cuts)

    def _replay_segments(self):
        graphs, cuts = self._segments
        for i, g in enumerate(graphs):
            g.replay()
            if i < len(cuts):
                self.sync.after_segment(cuts[i])
                if i == len(cuts) - 1:
                    self.sync.finish()                          # the last graph (tail of backward + AdamW) waits for the collectives
