"""Batch-level data parallelism: one process per GPU, gradient MEAN all-reduce over RCCL/xGMI.

Replaces Lightning's ``strategy='ddp'`` (``fcd_main.py:54-64``).  Gradients are packed
into a few large flat buckets (xGMI is point-to-point, 7 links x ~153 GB/s per GPU: few
large collectives beat many small ones), reduced asynchronously on the communicator's
stream and unpacked before the optimizer reads them.  Works with the ``gloo`` backend on
CPU tensors too (the world_size-2 tests)."""
from __future__ import annotations

from typing import List, Sequence

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

    def finish(self, handle) -> None:
        """Make the current stream wait for the collectives of ``handle``."""
        for w, v, t, op in handle:
            w.wait()
            if op == dist.ReduceOp.SUM:
                t.div_(self.world)
            if t is not v:
                v.copy_(t)

    def allreduce_flat(self, flat: torch.Tensor) -> None:
        self.finish(self.allreduce_flat_async(flat))

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
