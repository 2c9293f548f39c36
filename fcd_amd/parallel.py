"""Batch-level data parallelism: one process per GPU, gradient MEAN all-reduce over RCCL/xGMI.

Replaces Lightning's ``strategy='ddp'`` (``fcd_main.py:54-64``).  Gradients are packed
into a few large flat buckets (xGMI is point-to-point, 7 links x ~153 GB/s per GPU: few
large collectives beat many small ones), reduced asynchronously on the communicator's
stream and unpacked before the optimizer reads them.  Works with the ``gloo`` backend on
CPU tensors too (the world_size-2 tests)."""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def plan_buckets(numels: Sequence[int], bucket_elems: int) -> List[List[int]]:
    """Greedy contiguous bucketing in the given (reverse-backward) order."""
    buckets, cur, size = [], [], 0
    for i, n in enumerate(numels):
        if cur and size + n > bucket_elems:
            buckets.append(cur)
            cur, size = [], 0
        cur.append(i)
        size += n
    if cur:
        buckets.append(cur)
    return buckets


class GradSync:
    def __init__(self, bucket_mb: float = 64.0, group=None):
        self.bucket_elems = int(bucket_mb * (1 << 20) / 4)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._flat = {}
        # diagnostics (bench.py, the first real multi-GPU run): with `timing` set to a list, every finish() appends
        # (tag, host seconds spent inside finish, (event before, event after) on the current stream or None, bytes):
        # the event pair brackets exactly the stream's wait for the collectives -- the all-reduce time that backward
        # and the next forward did NOT hide
        self.timing = None

    def allreduce(self, params: List[torch.Tensor]) -> None:
        """Average ``p.grad`` over ranks, in place."""
        if self.world == 1 or not params:
            return
        params = list(reversed(params))            # last layers' grads are produced first
        grads = [p.grad for p in params]
        backend = dist.get_backend(self.group)
        op = dist.ReduceOp.AVG if backend == "nccl" else dist.ReduceOp.SUM
        work = []
        for bi, idx in enumerate(plan_buckets([g.numel() for g in grads], self.bucket_elems)):
            n = sum(grads[i].numel() for i in idx)
            bdev = grads[idx[0]].device if backend == "nccl" else torch.device("cpu")   # gloo reduces host memory
            key = (bi, n, bdev)
            flat = self._flat.get(key)
            if flat is None:
                flat = torch.empty(n, dtype=torch.float32, device=bdev)
                self._flat[key] = flat
            off = 0
            views = []
            for i in idx:
                v = flat[off:off + grads[i].numel()]
                v.copy_(grads[i].reshape(-1))
                views.append(v)
                off += grads[i].numel()
            work.append((dist.all_reduce(flat, op=op, group=self.group, async_op=True), idx, views))
        for w, idx, views in work:
            w.wait()
            for i, v in zip(idx, views):
                if op == dist.ReduceOp.SUM:
                    v.div_(self.world)
                grads[i].copy_(v.view_as(grads[i]))

    def allreduce_flat_async(self, flat: torch.Tensor):
        """Start averaging a gradient arena over ranks in place: bucket-sized views of the buffer
        itself go to the collective (nothing is packed or unpacked).  Returns a handle for
        :meth:`finish`; until then the collective runs on the communicator's stream, beside
        whatever is launched next."""
        if self.world == 1 or flat.numel() == 0:
            return []
        backend = dist.get_backend(self.group)
        op = dist.ReduceOp.AVG if backend == "nccl" else dist.ReduceOp.SUM
        work = []
        for off in range(0, flat.numel(), self.bucket_elems):
            v = flat[off:off + self.bucket_elems]
            # gloo (rehearsals on a box with fewer GPUs than ranks) reduces host memory: stage it
            t = v.cpu() if (backend != "nccl" and v.is_cuda) else v
            work.append((dist.all_reduce(t, op=op, group=self.group, async_op=True), v, t, op))
        return work

    def finish(self, handle, tag: str = "") -> None:
        """Make the current stream wait for the collectives of ``handle``."""
        rec = self.timing is not None and bool(handle)
        ev = None
        if rec:
            import time
            t0 = time.perf_counter()
            if handle[0][1].is_cuda:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
        for w, v, t, op in handle:
            w.wait()
            if op == dist.ReduceOp.SUM:
                t.div_(self.world)
            if t is not v:
                v.copy_(t)
        if rec:
            if ev is not None:
                ev[1].record()
            self.timing.append((tag, time.perf_counter() - t0, ev, sum(v.numel() * 4 for _, v, _, _ in handle)))

    def exposed_ms(self):
        """Per tag: mean milliseconds per finish() that the compute stream waited for gradient all-reduces (HIP events
        around the wait; host wall time where the tensors live on the CPU) and the bytes per finish; clears the
        record.  Call after a synchronisation."""
        out = {}
        for tag, host_s, ev, nbytes in self.timing or []:
            ms = ev[0].elapsed_time(ev[1]) if ev is not None else host_s * 1e3
            o = out.setdefault(tag, {"calls": 0, "stream_wait_ms": 0.0, "host_ms": 0.0, "bytes": 0})
            o["calls"] += 1
            o["stream_wait_ms"] += ms
            o["host_ms"] += host_s * 1e3
            o["bytes"] = nbytes
        for o in out.values():
            o["stream_wait_ms"] = round(o["stream_wait_ms"] / o["calls"], 3)
            o["host_ms"] = round(o["host_ms"] / o["calls"], 3)
        if self.timing is not None:
            self.timing.clear()
        return out

    def allreduce_flat(self, flat: torch.Tensor, tag: str = "") -> None:
        self.finish(self.allreduce_flat_async(flat), tag)

    def bucketed(self, arena, bucket_mb: Optional[float] = None) -> "BucketedReducer":
        """An overlapped reducer for one gradient arena (see :class:`BucketedReducer`)."""
        return BucketedReducer(self, arena, self.bucket_elems if bucket_mb is None else int(bucket_mb * (1 << 20) / 4))

    def broadcast_buffers(self, modules, src: int = 0) -> None:
        """Rank ``src``'s floating-point buffers (the Generator's tracked InstanceNorm running
        statistics) to every rank, as ONE flat broadcast.  DDP's ``broadcast_buffers=True``
        (Lightning's ``strategy='ddp'``, ``fcd_main.py:54-64``) re-broadcasts rank 0's buffers at
        every forward; training-mode numerics never read them, so doing it once before they are
        used (eval, validation, threshold search, checkpoint) gives every rank the statistics the
        reference's ranks would hold: rank 0's chain of updates."""
        if self.world == 1:
            return
        bufs = [b for m in modules for b in m.buffers() if b.is_floating_point() and b.numel()]
        if not bufs:
            return
        flat = torch.cat([b.detach().reshape(-1).float() for b in bufs])
        staged = dist.get_backend(self.group) != "nccl" and flat.is_cuda
        t = flat.cpu() if staged else flat
        dist.broadcast(t, src, group=self.group)
        if staged:
            flat = t.to(flat.device)
        off = 0
        with torch.no_grad():
            for b in bufs:
                b.copy_(flat[off:off + b.numel()].view_as(b))      # copy_ bumps the version: folded weights are rebuilt
                off += b.numel()

    def sum_counts(self, t: torch.Tensor) -> torch.Tensor:
        """Integer counts (confusion matrices, histograms) summed over ranks, in place."""
        if self.world == 1:
            return t
        staged = dist.get_backend(self.group) != "nccl" and t.is_cuda
        h = t.cpu() if staged else t
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
        if staged:
            t.copy_(h)
        return t

    def broadcast_params(self, modules) -> None:
        """Identical initial weights and buffers on all ranks (DDP does this at construction)."""
        if self.world == 1:
            return
        staged = dist.get_backend(self.group) != "nccl"
        for m in modules:
            for t in list(m.parameters()) + list(m.buffers()):
                if staged and t.is_cuda:
                    h = t.data.cpu()
                    dist.broadcast(h, 0, group=self.group)
                    t.data.copy_(h)
                else:
                    dist.broadcast(t.data, 0, group=self.group)
                # written through .data: bump the version so that everything keyed on (data_ptr, _version) -- packed
                # operands, the fp32x2 weight maxima, folded eval weights -- is rebuilt from the new values
                torch.autograd.graph.increment_version(t)


class BucketedReducer:
    """Gradient all-reduce overlapped with backward, for one :class:`fcd_amd.arena.ParamArena`.

    The arena is cut into spans at parameter boundaries, walking from its END (parameters are laid
    out in forward order, so backward finishes them last-to-first) and closing a span once it holds
    ``bucket_elems`` elements.  A network's parameter gradients are final only in the LAST backward
    pass of an optimizer step that touches them (the G step back-propagates through the generator 2
    or 4 times, the D step through the discriminator 2 or 3 times), so the solver declares the number
    of passes with :meth:`begin`; every hand-written backward chain calls :meth:`pass_begin`, and in
    the final pass :meth:`param_ready` right after it has enqueued the kernel that completes a
    parameter's gradient.  When the last parameter of a span is ready the span goes to the collective
    (views of the arena itself, pieces of at most ``bucket_elems``), which RCCL orders after the work
    already enqueued on the calling stream and runs on its own stream beside the rest of backward.
    :meth:`finish_rest` sends whatever is left and returns the handle for ``GradSync.finish``.

    Reporting more passes than declared means spans were reduced before their gradients were
    complete: that raises.  Fewer passes (a backward path without hooks) only loses the overlap."""

    def __init__(self, sync: GradSync, arena, bucket_elems: int):
        self.sync, self.arena, self.bucket_elems = sync, arena, max(1, int(bucket_elems))
        self.before_send = None
        n = len(arena.params)
        ends = list(arena.offsets[1:]) + [arena.flat.numel()]
        self.spans = []                 # (lo, hi) element ranges, last span of the arena first
        self.span_of = [0] * n          # parameter index -> span index
        hi, size, members = ends[-1], 0, []
        for i in range(n - 1, -1, -1):
            members.append(i)
            size += ends[i] - arena.offsets[i]
            if size >= self.bucket_elems or i == 0:
                for m in members:
                    self.span_of[m] = len(self.spans)
                self.spans.append((arena.offsets[i], hi))
                hi, size, members = arena.offsets[i], 0, []
        self.span_size = [0] * len(self.spans)
        for i in range(n):
            self.span_size[self.span_of[i]] += 1
        self.declared = 0
        self.begin(0)

    def begin(self, passes: int) -> None:
        self.declared, self.seen = int(passes), 0
        self.missing = list(self.span_size)
        self.ready = [False] * len(self.arena.params)
        self.fired = [False] * len(self.spans)
        self.handle = []
        self.fired_early = 0
        self.in_final = False

    def pass_begin(self) -> bool:
        """Called by a backward chain before its first parameter gradient; True in the final pass."""
        self.seen += 1
        if self.declared and self.seen > self.declared:
            raise RuntimeError(f"BucketedReducer: backward pass {self.seen} after {self.declared} were declared: "
                               "gradient spans may have been reduced before they were complete")
        # early sends are only valid when the kernels write into the arena themselves
        # (arena.accumulating()); gradients carried by autograd arrive after the chain has returned
        self.in_final = self.declared > 0 and self.seen == self.declared and self.arena.direct
        return self.in_final

    def _fire(self, si: int) -> None:
        lo, hi = self.spans[si]
        self.fired[si] = True
        if self.before_send is not None:       # gradient kernels on a side stream (ops.SideWork): order this one behind them
            self.before_send()
        for off in range(lo, hi, self.bucket_elems):
            self.handle += self.sync.allreduce_flat_async(self.arena.grad[off:min(hi, off + self.bucket_elems)])

    def param_ready(self, p) -> None:
        i = self.arena.index.get(id(p))
        if not self.in_final or i is None or self.ready[i]:
            return
        self.ready[i] = True
        si = self.span_of[i]
        self.missing[si] -= 1
        if self.missing[si] == 0 and not self.fired[si]:
            self._fire(si)
            self.fired_early += 1

    def describe(self):
        """Span layout for a benchmark line: bytes per span in the order they complete (end of the arena first), the
        parameters in each, and how many went out from inside backward in the last step."""
        names = {}
        for i, p in enumerate(self.arena.params):
            names.setdefault(self.span_of[i], []).append(i)
        return {"spans_mb": [round((hi - lo) * 4 / (1 << 20), 2) for lo, hi in self.spans],
                "params_per_span": [len(names.get(si, [])) for si in range(len(self.spans))],
                "piece_mb": round(self.bucket_elems * 4 / (1 << 20), 2),
                "sent_from_inside_backward_last_step": self.fired_early}

    def finish_rest(self):
        """After backward has returned: reduce the spans that were not sent from inside it."""
        for si in range(len(self.spans)):
            if not self.fired[si]:
                self._fire(si)
        h, self.handle = self.handle, []
        self.declared, self.in_final = 0, False
        return h
