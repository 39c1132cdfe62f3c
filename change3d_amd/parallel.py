"""Data-parallel training for the Change3D hot path: one process per GPU, `torch.distributed`
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" on CPU for tests).

The reference is single-GPU (no distributed code at all), so this layer is new.
Semantics: every rank runs the whole model on its own B samples with
PER-RANK BatchNorm statistics (no SyncBN — the reference's B=16 single-GPU statistics are
per-replica too) and a rank-local BCE+Dice loss; the one exchange per step is a
sum-all-reduce of the gradients divided by the world size.

All gradients live in ONE flat f32 buffer (`ParamArena.flat_grad`, ~7 MB for BCD), laid out so
that everything finished when the LAST encoder stage's backward returns (decoder, fc[3],
res4 = 75 % of the payload) is a contiguous tail.  That tail is all-reduced on a side HIP
stream while res3/res2/stem backward (the ~70 % of backward traffic that remains) still runs;
the head of the buffer follows at the end.  The payload is latency-bound on xGMI (7 MB ->
~0.9 MB per peer link), so two large collectives beat many small buckets.
"""
import time

import torch
import torch.distributed as dist

from . import ops
from .model.utils import ParamArena, cc_named_params, hot_path_named_params

_EARLY = ("encoder.x3d.blocks.3.", "encoder.fc.3.", "decoder")


def ordered_hot_params(trainer):
    """Hot-path parameters ordered [late-final ... | early-final tail]."""
    named = hot_path_named_params(trainer)
    late = [(n, p) for n, p in named if not n.startswith(_EARLY)]
    early = [(n, p) for n, p in named if n.startswith(_EARLY)]
    return late + early, len(late)


class GradSync:
    """Flat-buffer gradient all-reduce with one overlapped bucket."""

    def __init__(self, arena, split_index, world_size=None, group=None):
        self.arena = arena
        self.group = group
        self.world = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.split = arena.offsets[split_index] if split_index < len(arena.offsets) else arena.numel
        self.use_stream = arena.flat_grad.is_cuda
        self.comm_stream = torch.cuda.Stream(device=arena.flat_grad.device) if self.use_stream else None
        self._tail_launched = False
        # self-diagnosis (bench.py --gpus N): with `timing` on every step records how long the compute stream sat behind the
        # exchange -- (tail) from the moment it reached the fence in finish() to the end of the overlapped bucket on the
        # communication stream, (head) the duration of the non-overlapped all-reduce -- as event pairs (GPU) or host seconds
        self.timing = False
        self.samples = []       # per step: (ready_event, tail_end_event, head_start_event, head_end_event) or (tail_s, head_s)
        self._tail_end = None
        self._tail_host_s = 0.0

    # -- called from the last encoder stage's backward (model/x3d.py: stage.post_backward)
    def launch_tail(self):
        if self.world == 1 or self._tail_launched:
            return
        self._tail_launched = True
        tail = self.arena.flat_grad[self.split:]
        if self.use_stream:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                dist.all_reduce(tail, op=dist.ReduceOp.SUM, group=self.group)
                if self.timing:
                    self._tail_end = torch.cuda.Event(enable_timing=True)
                    self._tail_end.record()
        else:
            t0 = time.perf_counter()
            dist.all_reduce(tail, op=dist.ReduceOp.SUM, group=self.group)
            self._tail_host_s = time.perf_counter() - t0   # no second stream on the host: the whole bucket is exposed

    def finish(self):
        """After backward: reduce what is left, average, and fence the compute stream."""
        if self.world == 1:
            return
        self.arena.check_grads_attached()   # a detached p.grad would leave its real gradient out of the exchange
        g = self.arena.flat_grad
        timed = self.timing
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if (timed and self.use_stream) else None
        t0 = time.perf_counter()
        if ev:
            ev[0].record()                                   # the compute stream is ready for the reduced gradients here
        if self._tail_launched:
            head = g[:self.split]
            if head.numel():
                dist.all_reduce(head, op=dist.ReduceOp.SUM, group=self.group)
            if ev:
                ev[1].record()
            if self.use_stream:
                torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
            if ev:
                ev[1].record()
        if timed:
            if ev:
                self.samples.append((ev[0], ev[1], self._tail_end))
            else:
                self.samples.append((self._tail_host_s, time.perf_counter() - t0))
        self._tail_end, self._tail_host_s = None, 0.0
        g.mul_(1.0 / self.world)
        self._tail_launched = False

    def exposed_ms_per_step(self):
        """Mean over the recorded steps of {"head_allreduce": the non-overlapped all-reduce, "tail_wait": the part of the overlapped
        bucket that was still running when the compute stream reached the fence} in ms (call after a device synchronise)."""
        if not self.samples:
            return None
        head = tail = 0.0
        for smp in self.samples:
            if isinstance(smp[0], float):
                tail += smp[0] * 1e3
                head += smp[1] * 1e3
            else:
                ready, head_end, tail_end = smp
                head += ready.elapsed_time(head_end)
                if tail_end is not None:
                    tail += max(0.0, head_end.elapsed_time(tail_end))   # what is left of the bucket after the head finished
        n = len(self.samples)
        return {"head_allreduce": round(head / n, 4), "tail_wait": round(tail / n, 4), "steps": n,
                "tail_bytes": int((self.arena.numel - self.split) * 4), "head_bytes": int(self.split * 4)}


_HOST_GROUP = None


def host_barrier(timeout_hours=12.0):
    """Every rank waits here for the slowest one, on the HOST (a gloo group with a long timeout, created on first use).
    Used where one rank does long rank-local work between exchanges -- rank 0 validating and writing a checkpoint after an
    epoch: the other ranks must neither sit in an RCCL collective for longer than its watchdog allows (10 min by default)
    nor tear the process group down while rank 0 still works."""
    global _HOST_GROUP
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    if _HOST_GROUP is None:
        import datetime
        _HOST_GROUP = dist.new_group(backend="gloo", timeout=datetime.timedelta(hours=timeout_hours))
    dist.barrier(group=_HOST_GROUP)


def broadcast_module_state(module, src=0, group=None):
    """Rank-`src` parameters and buffers to everyone (once, at start)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
    ops.bump_weights_version()   # in-place writes: folded-BatchNorm weights built before this are stale


def setup_data_parallel(trainer, device, overlap=True, group=None):
    """Build the arena in all-reduce-friendly order, hook the overlapped bucket, return
    (arena, grad_sync)."""
    named, n_late = ordered_hot_params(trainer)
    arena = ParamArena(named, device)
    sync = GradSync(arena, n_late, group=group)
    if overlap and sync.world > 1:
        trainer.encoder.x3d.blocks[3].post_backward = sync.launch_tail
    return arena, sync


class GradSyncGroup:
    """Several flat buffers exchanged as one unit (change captioning: the encoder's and the decoder's Adam own
    separate arenas, reference scripts/train_CC.py:436-458)."""

    def __init__(self, syncs):
        self.syncs = list(syncs)
        self.world = self.syncs[0].world

    def launch_tail(self):
        for s in self.syncs:
            s.launch_tail()

    def finish(self):
        for s in self.syncs:
            s.finish()


def setup_data_parallel_cc(trainer, device, overlap=True, group=None):
    """Change-captioning path (CC row: 5.65 M gradient-bearing parameters = 22.6 MB): returns
    ((enc_arena, enc_sync), (dec_arena, dec_sync), both) where `both.finish()` completes the exchange.

    Backward runs decoder -> res5 -> res4 -> ... -> stem, so when `blocks[4]` (res5) returns from backward the WHOLE
    decoder arena and res5 (2.9 M of the encoder's 4.4 M parameters) are final: the encoder arena is ordered
    [stem .. res4 | res5] and the hook at the end of res5's backward launches both buckets -- 4.2 M of the 5.65 M floats --
    on the communication streams while res4 .. stem backward (most of the backward pass) still run."""
    enc_named, dec_named = cc_named_params(trainer)
    tail = "encoder.x3d.blocks.4."
    late = [(n, p) for n, p in enc_named if not n.startswith(tail)]
    early = [(n, p) for n, p in enc_named if n.startswith(tail)]
    enc_arena, dec_arena = ParamArena(late + early, device), ParamArena(dec_named, device)
    enc_sync = GradSync(enc_arena, len(late), group=group)
    dec_sync = GradSync(dec_arena, 0, group=group)          # split 0: the whole decoder buffer is the overlapped bucket
    both = GradSyncGroup([dec_sync, enc_sync])
    if overlap and enc_sync.world > 1:
        trainer.encoder.x3d.blocks[4].post_backward = both.launch_tail
    return (enc_arena, enc_sync), (dec_arena, dec_sync), both
