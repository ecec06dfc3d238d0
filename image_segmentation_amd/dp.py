"""Data-parallel step for LoRA fine-tuning: one process per GPU, frames sharded by rank, and ONE
exchange step per iteration -- a mean all-reduce of the LoRA gradients only (28,672 fp32 values at
r=4).  This replaces DistributedDataParallel's reducer (train_lora_sam.py:126,172; SURVEY.md
section 2b): with every non-LoRA parameter frozen the reducer's traffic is exactly these tensors.

The gradients travel in buckets of one LoRA block each (4 tensors, 56 KiB at r=4), in the order the
backward produces them.  `attach(trunk)` registers a hook with the trunk so that a block's bucket is
all-reduced (RCCL, backend "nccl" on ROCm, on its own stream) as soon as that block's backward is
enqueued, overlapping the rest of the backward; `reduce()` then only waits, scales by 1/world and
writes the result to `.grad`.  Without `attach` (or for gradients the hook never saw) `reduce()`
all-reduces the buckets itself.  At this size the collective is latency-bound, not link-bound, so
the buckets are not split further.  On CPU the same code runs over gloo (tests/test_dp_cpu.py).

Whenever a process group exists the exchange step RUNS, whatever the group's size: a one-rank group (a 1-GPU box under
torch.distributed.run, tests/test_dp_gpu.py) still builds the RCCL communicator, broadcasts, and all-reduces every bucket
from inside the backward -- the mean over one rank is the identity, so results stay bit-equal to the plain step.  Only
without any process group (plain `python bench.py`) are the calls no-ops.  `stats` counts the collectives issued."""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class _Bucket:
    def __init__(self, params, flat):
        self.params = params
        self.flat = flat
        self.views = []
        o = 0
        for p in params:
            self.views.append(flat[o:o + p.numel()].view_as(p))
            o += p.numel()
        self.work = None      # in-flight all-reduce started by the hook
        self.filled = False   # flat holds this step's local gradients
        self.pending = []     # (increment buffer, work) of further backwards before reduce()


class LoRAGradReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], group: Optional[dist.ProcessGroup] = None,
                 tensors_per_bucket: int = 4):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("LoRAGradReducer: no trainable parameters")
        if tensors_per_bucket < 1:
            raise ValueError("LoRAGradReducer: tensors_per_bucket must be positive")
        self.group = group
        self.numel = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=p0.device)
        self.buckets: List[_Bucket] = []
        o = 0
        for i in range(0, len(self.params), tensors_per_bucket):
            ps = self.params[i:i + tensors_per_bucket]
            n = sum(p.numel() for p in ps)
            self.buckets.append(_Bucket(ps, self.flat[o:o + n]))
            o += n
        self.views = [v for b in self.buckets for v in b.views]
        self._by_first = {id(b.params[0]): b for b in self.buckets}
        self._invalidate = []   # trunks whose packed weights must be refreshed after parameters change through .data
        self.stats = {"broadcast": 0, "all_reduce": 0, "all_reduce_from_backward": 0}

    def active(self) -> bool:
        """A process group exists: collectives are issued (also on a one-rank group)."""
        return dist.is_available() and dist.is_initialized()

    def world_size(self) -> int:
        return dist.get_world_size(self.group) if self.active() else 1

    def _all_reduce(self, t, from_backward=False):
        self.stats["all_reduce"] += 1
        if from_backward:
            self.stats["all_reduce_from_backward"] += 1
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def attach(self, trunk) -> "LoRAGradReducer":
        """Overlap mode: the trunk's backward hands each LoRA block's gradients to on_bucket()."""
        trunk.set_grad_bucket_hook(self.on_bucket)
        inv = getattr(trunk, "invalidate_packed", None)
        if inv is not None:
            self._invalidate.append(inv)
        return self

    @torch.no_grad()
    def broadcast_params(self, src: int = 0):
        """DDP's constructor broadcast, restricted to the trainable tensors (frozen ones are built
        from the same checkpoint / seed on every rank)."""
        if not self.active():
            return
        torch._foreach_copy_(self.views, [p.data for p in self.params])
        dist.broadcast(self.flat, src=src, group=self.group)
        self.stats["broadcast"] += 1
        torch._foreach_copy_([p.data for p in self.params], self.views)
        for inv in self._invalidate:
            inv()

    @torch.no_grad()
    def on_bucket(self, params, grads):
        """Called from the backward (autograd thread) once the gradients of `params` are enqueued on
        the current stream: stage them in the bucket and start its all-reduce without waiting."""
        b = self._by_first.get(id(params[0]))
        if b is None or len(params) != len(b.params) or any(p is not q for p, q in zip(params, b.params)):
            return   # not one of our buckets: reduce() falls back to .grad
        if not self.active():
            return
        if b.filled:
            # a second backward before reduce() (gradient accumulation, two views per step): DDP sums the micro-batches.
            # The first all-reduce may still be in flight on b.flat, so the increment gets a staging buffer of its own
            # and is reduced separately: sum_r(g1_r) + sum_r(g2_r) = sum_r(g1_r + g2_r); reduce() adds them up.
            inc = torch.empty_like(b.flat)
            o = 0
            views = []
            for p in b.params:
                views.append(inc[o:o + p.numel()].view_as(p))
                o += p.numel()
            torch._foreach_copy_(views, list(grads))
            b.pending.append((inc, self._all_reduce(inc, True)))
            return
        torch._foreach_copy_(b.views, list(grads))
        b.filled = True
        b.work = self._all_reduce(b.flat, True)

    @torch.no_grad()
    def reduce(self):
        """grads <- mean over ranks.  Missing grads count as zero (DDP semantics)."""
        if not self.active():
            return
        ws = self.world_size()
        for b in reversed(self.buckets):   # the order the backward produced (and launched) them
            if not b.filled:
                grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in b.params]
                torch._foreach_copy_(b.views, grads)
                b.work = self._all_reduce(b.flat)
        for b in reversed(self.buckets):
            b.work.wait()
            for inc, work in b.pending:
                work.wait()
                b.flat.add_(inc)
            b.pending = []
            b.work, b.filled = None, False
        if ws > 1:
            self.flat.mul_(1.0 / ws)
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)
