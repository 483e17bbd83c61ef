"""Data-parallel plumbing for the audio path: one process per GPU, torch.distributed over RCCL
(backend "nccl" on ROCm) on the GPU box, gloo in the CPU tests.

The reference is single-process (configs/accelerate/a40.yaml:3,10: distributed_type 'NO'); under
accelerate/HF Trainer a multi-process launch would wrap the model in torch DDP, whose only
collective on this path is the gradient all-reduce of the trainable parameters (projector, plus
the decoder's LoRA adapters): SURVEY.md section 8e.  The forward hot path needs no collective:
clips are independent, so a global batch is simply sharded.

`FlatGradAllReducer` is that one exchange step written for xGMI: the trainable gradients are a
few MB (projector 6.65 MB fp32), so the transfer is latency-bound, not link-bound -- one flat
buffer and ONE all-reduce per optimizer step beats per-parameter or small-bucket rings.
Gradients are views into the flat buffer, so there is no gather/scatter copy around the call.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world_size: int) -> range:
    """Contiguous, balanced shard of `n_items` clips for `rank` (the first n % world ranks get one more)."""
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def shard_batch(items: Sequence, rank: Optional[int] = None, world_size: Optional[int] = None) -> List:
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    return [items[i] for i in shard_range(len(items), rank, world_size)]


class FlatGradAllReducer:
    """Keeps every trainable parameter's .grad as a view of one flat fp32 buffer and averages it
    across ranks with a single all-reduce."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in self.params):
            raise ValueError("all parameters must share one device and dtype")
        self.group = group
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, device=dev, dtype=dt)
        off = 0
        for p in self.params:
            p.grad = self.flat[off: off + p.numel()].view_as(p)
            off += p.numel()
        self._work = None

    def zero_(self):
        self.flat.zero_()

    def rebind(self):
        """Re-attach .grad views (an optimizer's zero_grad(set_to_none=True) drops them)."""
        off = 0
        for p in self.params:
            v = self.flat[off: off + p.numel()].view_as(p)
            if p.grad is None:
                p.grad = v
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
                p.grad = v
            off += p.numel()

    def all_reduce(self, async_op: bool = False):
        """Average the gradients over the group.  With async_op the call returns at once (overlap it
        with the decoder's backward) and `wait()` must be called before the optimizer step."""
        self.rebind()
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return None
        ws = dist.get_world_size(self.group)
        self.flat.div_(ws)  # pre-divide: SUM of the pre-scaled gradients == mean, no post-pass to wait for
        self._work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        return self._work

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
