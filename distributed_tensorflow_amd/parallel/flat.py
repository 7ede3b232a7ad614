"""Flat parameter / gradient storage.

All trainable parameters of a model are re-homed into a few large contiguous
buffers so that

* the optimizer is one fused kernel launch per buffer instead of one per tensor,
* gradient collectives (all-reduce, reduce-scatter) and parameter collectives
  (all-gather) run on a handful of large, evenly divisible buckets -- the
  message sizes a point-to-point xGMI mesh wants -- instead of ~160 small ones,
* the weight-gradient kernels can write fp32 results straight into their final
  location (``param.main_grad``), with no autograd accumulation copies.

Two groups exist: ``lowp`` for parameters whose compute dtype is bf16 (conv and
dense weights; they get an fp32 master copy) and ``fp32`` for parameters that
are consumed in fp32 (batch-norm gamma/beta, biases; master == compute).  On the
CPU everything is fp32 and lands in the ``fp32`` group.

Parameters are laid out in *reverse registration order*: backward produces
gradients roughly last-layer-first, so bucket 0 fills first and its collective
can start while the rest of the backward pass is still running.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional

import torch
import torch.nn as nn

ALIGN = 64  # elements; keeps every parameter and every shard 128-byte aligned


@dataclass
class Segment:
    name: str
    param: nn.Parameter
    offset: int
    numel: int
    no_decay: bool


@dataclass
class Bucket:
    group: "FlatGroup"
    index: int
    start: int
    end: int
    segments: List[Segment] = field(default_factory=list)
    pending: int = 0

    @property
    def numel(self) -> int:
        return self.end - self.start


class FlatGroup:
    """One flat (compute, master, grad) triple."""

    def __init__(self, name: str, named_params, device, compute_dtype, grad_dtype, bucket_elems: int,
                 shard_multiple: int):
        self.name = name
        self.compute_dtype = compute_dtype
        self.segments: List[Segment] = []
        off = 0
        for pname, p in named_params:
            n = p.numel()
            self.segments.append(Segment(pname, p, off, n, p.dim() <= 1))
            off += (n + ALIGN - 1) // ALIGN * ALIGN
        # buckets: contiguous ranges, each a multiple of shard_multiple*ALIGN so that it splits evenly
        quantum = ALIGN * max(1, shard_multiple)
        self.buckets: List[Bucket] = []
        start = 0
        cur: List[Segment] = []
        for seg in self.segments:
            cur.append(seg)
            end = seg.offset + (seg.numel + ALIGN - 1) // ALIGN * ALIGN
            if end - start >= bucket_elems:
                end_pad = start + (end - start + quantum - 1) // quantum * quantum
                self.buckets.append(Bucket(self, len(self.buckets), start, end_pad, cur))
                # later segments must start after the padded end
                shift = end_pad - end
                if shift:
                    for s2 in self.segments[self.segments.index(seg) + 1:]:
                        s2.offset += shift
                start, cur = end_pad, []
        if cur:
            end = cur[-1].offset + (cur[-1].numel + ALIGN - 1) // ALIGN * ALIGN
            end_pad = start + (end - start + quantum - 1) // quantum * quantum
            self.buckets.append(Bucket(self, len(self.buckets), start, end_pad, cur))
            start = end_pad
        self.total = start
        self.master = torch.zeros(self.total, dtype=torch.float32, device=device)
        if compute_dtype == torch.float32:
            self.compute = self.master
        else:
            self.compute = torch.zeros(self.total, dtype=compute_dtype, device=device)
        self.grad = torch.zeros(self.total, dtype=grad_dtype, device=device)
        for seg in self.segments:
            p = seg.param
            sl = slice(seg.offset, seg.offset + seg.numel)
            self.master[sl].copy_(p.data.detach().reshape(-1).to(torch.float32))
            if self.compute is not self.master:
                self.compute[sl].copy_(p.data.detach().reshape(-1).to(compute_dtype))
            keep = getattr(p, "_keep_fp32", False)
            p.data = self.compute[sl].view(p.shape)
            p._keep_fp32 = keep
            p.main_grad = self.grad[sl].view(p.shape)
            p._grad_filled = False
            p._flat_segment = seg
            p.grad = None

    def decay_runs(self, lo: int, hi: int):
        """Maximal runs ``(a, b, decay)`` inside ``[lo, hi)`` whose parameters share
        the same weight-decay eligibility (matrices / filters decay, 1-D
        parameters such as batch-norm scales and biases do not)."""
        runs = []
        for i, seg in enumerate(self.segments):
            a = seg.offset
            b = self.segments[i + 1].offset if i + 1 < len(self.segments) else self.total
            a, b = max(a, lo), min(b, hi)
            if b <= a:
                continue
            decay = not seg.no_decay
            if runs and runs[-1][2] == decay and runs[-1][1] == a:
                runs[-1] = (runs[-1][0], b, decay)
            else:
                runs.append((a, b, decay))
        return runs

    def sync_compute_from_master(self) -> None:
        if self.compute is not self.master:
            if self.master.is_cuda:
                from ..ops import kernels as K

                K.cast_f32_bf16(self.master, self.compute)
            else:
                self.compute.copy_(self.master.to(self.compute.dtype))


class FlatParams:
    """Flat storage for every trainable parameter of ``model``.

    ``bucket_mb`` is the target size of a gradient bucket in MiB of gradient
    bytes; ``shard_multiple`` (the data-parallel world size for sharded modes)
    makes every bucket evenly divisible.
    """

    def __init__(self, model: nn.Module, bucket_mb: float = 32.0, shard_multiple: int = 1,
                 grad_dtype: torch.dtype = torch.float32):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        if not named:
            raise ValueError("model has no trainable parameters")
        named.reverse()
        device = named[0][1].device
        self.device = device
        lowp = [(n, p) for n, p in named if p.dtype != torch.float32]
        fp32 = [(n, p) for n, p in named if p.dtype == torch.float32]
        bucket_elems = max(ALIGN, int(bucket_mb * (1 << 20) / torch.empty((), dtype=grad_dtype).element_size()))
        self.groups: List[FlatGroup] = []
        if lowp:
            self.groups.append(FlatGroup("lowp", lowp, device, lowp[0][1].dtype, grad_dtype, bucket_elems,
                                         shard_multiple))
        if fp32:
            self.groups.append(FlatGroup("fp32", fp32, device, torch.float32, torch.float32, bucket_elems,
                                         shard_multiple))
        self.params: List[nn.Parameter] = [p for _, p in named]
        self._bucket_of: Dict[int, Bucket] = {}
        for g in self.groups:
            for b in g.buckets:
                for s in b.segments:
                    self._bucket_of[id(s.param)] = b
        self.on_bucket_ready: Optional[Callable[[Bucket], None]] = None
        for p in self.params:
            p._grad_ready_cb = self._grad_ready
        self.begin_step()

    # -- per-step bookkeeping ------------------------------------------------
    def begin_step(self) -> None:
        for g in self.groups:
            for b in g.buckets:
                b.pending = len(b.segments)
        for p in self.params:
            p._grad_filled = False

    def _grad_ready(self, p) -> None:
        # called once per parameter per step (the first time its gradient is written)
        b = self._bucket_of[id(p)]
        b.pending -= 1
        if b.pending == 0 and self.on_bucket_ready is not None:
            self.on_bucket_ready(b)

    def collect_autograd_grads(self) -> None:
        """Fold ``.grad`` tensors produced by regular autograd (CPU reference
        ops, or GPU ops without a fused writer) into the flat buffers and make
        sure every bucket is reported ready exactly once."""
        for p in self.params:
            if p.grad is not None:
                if p._grad_filled:
                    p.main_grad.add_(p.grad.to(p.main_grad.dtype))
                else:
                    p.main_grad.copy_(p.grad.to(p.main_grad.dtype))
                p.grad = None
                if not p._grad_filled:
                    p._grad_filled = True
                    self._grad_ready(p)
            elif not p._grad_filled:
                # unused this step: zero gradient
                p.main_grad.zero_()
                p._grad_filled = True
                self._grad_ready(p)

    def buckets(self) -> List[Bucket]:
        return [b for g in self.groups for b in g.buckets]

    def total_numel(self) -> int:
        return sum(g.total for g in self.groups)

    def zero_grads(self) -> None:
        for g in self.groups:
            g.grad.zero_()
