"""Data-parallel step for LoRA fine-tuning: one process per GPU, frames sharded by rank,
and ONE exchange per step -- a mean all-reduce of the LoRA gradients only, as one flat fp32
buffer (28,672 floats at r=4).  This replaces DistributedDataParallel's reducer
(train_lora_sam.py:126,172): with every non-LoRA parameter frozen the
reducer's single bucket holds exactly these eight tensors.

The message is ~112 KiB, so the collective is latency-bound, not link-bound: it is issued as a
single RCCL all-reduce (backend "nccl" on ROCm) right after the backward call; on CPU the
same code runs over gloo (tests/test_dp_cpu.py)."""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class LoRAGradReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], group: Optional[dist.ProcessGroup] = None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("LoRAGradReducer: no trainable parameters")
        self.group = group
        self.numel = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=p0.device)
        self.views = []
        o = 0
        for p in self.params:
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()

    def world_size(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    @torch.no_grad()
    def broadcast_params(self, src: int = 0):
        """DDP's constructor broadcast, restricted to the trainable tensors (frozen ones are built
        from the same checkpoint / seed on every rank)."""
        if self.world_size() == 1:
            return
        torch._foreach_copy_(self.views, [p.data for p in self.params])
        dist.broadcast(self.flat, src=src, group=self.group)
        torch._foreach_copy_([p.data for p in self.params], self.views)

    @torch.no_grad()
    def reduce(self):
        """grads <- mean over ranks.  Missing grads count as zero (DDP semantics)."""
        ws = self.world_size()
        if ws == 1:
            return
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        torch._foreach_copy_(self.views, grads)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.mul_(1.0 / ws)
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)
