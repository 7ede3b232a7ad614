"""Fused optimizers over flat fp32 regions.

A *region* is a contiguous fp32 master buffer with a matching gradient buffer
and, when the compute dtype is bf16, a bf16 shadow buffer.  For replicated
data-parallel training a region is a whole flat group; for the sharded
parameter-server modes it is the locally owned shard (only the owner keeps
Adam / momentum slots -- the same placement TensorFlow uses when ApplyAdam runs
on the ps next to the variable; reference:
recognition_flower/distributed/train_dataset.py:147-165).

On the GPU one HIP kernel per region reads the gradient (fp32 or bf16),
updates master weights and slots and writes the bf16 copy.  On the CPU the same
arithmetic runs as torch ops.

Adam follows TensorFlow-1 ``ApplyAdam`` (bias correction folded into the step
size, epsilon outside the square root), which is what the reference trains
with (recognition_flower/single/flower_train.py:148).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Union

import torch

from .schedules import Schedule, constant


@dataclass
class OptRegion:
    name: str
    w: torch.Tensor                     # fp32 master (updated in place)
    g: torch.Tensor                     # gradient, fp32 or bf16, same length
    w16: Optional[torch.Tensor]         # bf16 shadow or None
    decay: Union[bool, torch.Tensor]    # weight-decay eligibility (bool, or a 0/1 mask on the CPU)


def regions_from_flat(flat) -> List[OptRegion]:
    """Whole-buffer regions for replicated training."""
    out = []
    for g in flat.groups:
        w16 = g.compute if g.compute is not g.master else None
        out.append(OptRegion(g.name, g.master, g.grad, w16, decay_spec(g, 0, g.total)))
    return out


def decay_spec(group, lo: int, hi: int, index: Optional[torch.Tensor] = None):
    """bool when the whole range shares one eligibility, else a 0/1 fp32 mask
    (CPU only).  ``index`` optionally maps region positions to group positions."""
    runs = group.decay_runs(0, group.total)
    if index is None:
        runs = group.decay_runs(lo, hi)
    flags = {r[2] for r in runs}
    if len(flags) <= 1:
        return bool(flags.pop()) if flags else False
    if group.master.is_cuda:
        raise ValueError("mixed weight-decay eligibility inside one GPU region")
    full = torch.zeros(group.total, dtype=torch.float32)
    for a, b, d in group.decay_runs(0, group.total):
        if d:
            full[a:b] = 1.0
    return full[index] if index is not None else full[lo:hi].clone()


class _RegionOptimizer:
    def __init__(self, regions: List[OptRegion], lr):
        self.regions = regions
        self.schedule: Schedule = lr if callable(lr) else constant(float(lr))
        self.step_count = 0
        self._hyper: Dict[str, torch.Tensor] = {}

    def current_lr(self) -> float:
        return self.schedule(self.step_count)

    def _hyper_for(self, r: OptRegion) -> torch.Tensor:
        h = self._hyper.get(r.name)
        if h is None:
            h = torch.zeros(2, dtype=torch.float32, device=r.w.device)
            self._hyper[r.name] = h
        return h

    # A step is split in a host part and a device part so that the device part can be recorded in a
    # HIP graph: ``prepare`` evaluates the schedule and uploads (step size, gradient scale) into a
    # small device array, ``launch`` enqueues kernels that read it, ``advance`` counts the step.
    def _step_size(self) -> float:
        raise NotImplementedError

    def prepare(self, grad_scale: float = 1.0) -> None:
        a = self._step_size()
        self._host_hyper = (a, float(grad_scale))
        for r in self.regions:
            if r.w.is_cuda and r.w.numel():
                from ..ops import hipk

                hipk.check(hipk.lib().dtfa_set_hyper(self._hyper_for(r).data_ptr(), float(a), float(grad_scale),
                                                     hipk.stream()), "set_hyper")

    def launch(self) -> None:
        raise NotImplementedError

    def advance(self) -> None:
        self.step_count += 1

    def step(self, grad_scale: float = 1.0) -> None:
        self.prepare(grad_scale)
        self.launch()
        self.advance()

    def slots(self) -> Dict[str, torch.Tensor]:
        raise NotImplementedError


class FusedAdam(_RegionOptimizer):
    def __init__(self, regions: List[OptRegion], lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
        super().__init__(regions, lr)
        self.b1, self.b2, self.eps = beta1, beta2, eps
        self.m = {r.name: torch.zeros_like(r.w) for r in regions}
        self.v = {r.name: torch.zeros_like(r.w) for r in regions}

    @property
    def beta1_power(self) -> float:
        """TensorFlow's ``beta1_power`` variable after ``step_count`` applies."""
        return self.b1 ** (self.step_count + 1)

    @property
    def beta2_power(self) -> float:
        return self.b2 ** (self.step_count + 1)

    def _step_size(self) -> float:
        t = self.step_count + 1
        lr = self.schedule(self.step_count)
        return lr * math.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)

    def launch(self) -> None:
        lr_t, grad_scale = self._host_hyper
        for r in self.regions:
            if r.w.numel() == 0:
                continue
            m, v = self.m[r.name], self.v[r.name]
            if r.w.is_cuda:
                from ..ops import kernels as K

                K.adam_step(r.w, m, v, r.g, r.w16, self._hyper_for(r), self.b1, self.b2, self.eps)
            else:
                gf = r.g.float() * grad_scale
                m.mul_(self.b1).add_(gf, alpha=1 - self.b1)
                v.mul_(self.b2).addcmul_(gf, gf, value=1 - self.b2)
                r.w.sub_(lr_t * m / (v.sqrt() + self.eps))
                if r.w16 is not None:
                    r.w16.copy_(r.w.to(r.w16.dtype))

    def slots(self):
        out = {}
        for r in self.regions:
            out["Adam/" + r.name] = self.m[r.name]
            out["Adam_1/" + r.name] = self.v[r.name]
        return out


class FusedMomentum(_RegionOptimizer):
    def __init__(self, regions: List[OptRegion], lr=0.1, momentum=0.9, weight_decay=0.0, nesterov=False):
        super().__init__(regions, lr)
        self.mu, self.wd, self.nesterov = momentum, weight_decay, nesterov
        self.buf = {r.name: torch.zeros_like(r.w) for r in regions}

    def _step_size(self) -> float:
        return self.schedule(self.step_count)

    def launch(self) -> None:
        lr, grad_scale = self._host_hyper
        for r in self.regions:
            if r.w.numel() == 0:
                continue
            buf = self.buf[r.name]
            if r.w.is_cuda:
                from ..ops import kernels as K

                wd = self.wd if r.decay is True else 0.0
                K.momentum_step(r.w, buf, r.g, r.w16, self._hyper_for(r), self.mu, wd, self.nesterov)
            else:
                gp = r.g.float() * grad_scale
                if self.wd:
                    if isinstance(r.decay, torch.Tensor):
                        gp = gp + self.wd * r.decay * r.w
                    elif r.decay:
                        gp = gp + self.wd * r.w
                buf.mul_(self.mu).add_(gp)
                r.w.sub_(lr * (gp + self.mu * buf if self.nesterov else buf))
                if r.w16 is not None:
                    r.w16.copy_(r.w.to(r.w16.dtype))

    def slots(self):
        return {"Momentum/" + r.name: self.buf[r.name] for r in self.regions}


def make_optimizer(name: str, regions: List[OptRegion], lr, **kw):
    name = name.lower()
    if name == "adam":
        return FusedAdam(regions, lr, **{k: v for k, v in kw.items() if k in ("beta1", "beta2", "eps")})
    if name in ("momentum", "sgd"):
        return FusedMomentum(regions, lr,
                             **{k: v for k, v in kw.items() if k in ("momentum", "weight_decay", "nesterov")})
    raise ValueError("unknown optimizer %r" % name)
