"""Differentiable ops.

GPU tensors run the gfx950 HIP kernels through ``torch.autograd.Function``
wrappers (there is no other GPU implementation: a missing kernel library is
an error).  CPU tensors run the fp32 reference ops in :mod:`ops.ref`, which is
what the CPU plumbing configurations and the numerics tests use.

Layout conventions: activations ``[B, H, W, C]`` (NHWC, contiguous), conv
weights ``[Cout, KH, KW, Cin]``, dense weights ``[Out, In]``.  On the GPU
activations and conv/dense weights are bf16; batch-norm parameters, biases and
statistics are fp32.

Weight gradients: when a parameter carries a ``main_grad`` attribute (an fp32
view into a flat gradient bucket, installed by ``parallel.flat.FlatParams``)
the backward kernels write the gradient straight into it, mark
``_grad_filled`` and fire ``_grad_ready_cb`` (used to launch bucketed
collectives while the rest of the backward pass is still running); autograd
then sees ``None`` for that input.  Without ``main_grad`` the gradient is
returned to autograd in the parameter's dtype.
"""
from __future__ import annotations

import os

from typing import Optional, Tuple

import torch
from torch.autograd import Function

from . import conv as cv
from . import kernels as K
from . import ref

BF16 = torch.bfloat16
F32 = torch.float32

_force_reference = [False]


class reference_mode:
    """Test/diagnostic context: run the plain PyTorch reference ops even for GPU
    tensors (used to calibrate bf16 noise against the fp32 CPU oracle)."""

    def __enter__(self):
        self._old = _force_reference[0]
        _force_reference[0] = True

    def __exit__(self, *a):
        _force_reference[0] = self._old


def _use_hip(t: torch.Tensor) -> bool:
    return t.is_cuda and not _force_reference[0]


def _needs_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def concat_plan(x: torch.Tensor, *tensors) -> bool:
    """True when branch outputs computed from ``x`` may be written straight into channel slices of one
    buffer: device path and no autograd graph to build (frozen trunk or ``no_grad``)."""
    return _use_hip(x) and not _needs_grad(x, *tensors)


# ----------------------------------------------------------------------------
# gradient delivery
# ----------------------------------------------------------------------------
def _notify(param) -> None:
    first = not getattr(param, "_grad_filled", False)
    param._grad_filled = True
    cb = getattr(param, "_grad_ready_cb", None)
    if first and cb is not None:
        cb(param)


def _deliver_small(param, grad: Optional[torch.Tensor]):
    """fp32 per-channel gradients (gamma / beta / bias)."""
    if param is None or grad is None:
        return None
    mg = getattr(param, "main_grad", None)
    if mg is None:
        return grad.to(param.dtype).view_as(param)
    if getattr(param, "_grad_filled", False):
        mg.add_(grad.view_as(mg))
    else:
        mg.copy_(grad.view_as(mg))
    _notify(param)
    return None


# Weight gradients on a side stream.  In the backward pass the weight gradient of a layer is off the critical
# path (nothing downstream reads it before the optimizer / the bucket's collective), while the data gradient
# and the batch-norm backward of the next layer are on it.  With DTFA_WGRAD_STREAM=1 every weight-gradient
# kernel that writes into the flat gradient buffer is enqueued on one side stream per device, behind an event
# of the main stream, and the main stream joins it where the gradients are consumed (`join_wgrad_stream`).
# Measured on ResNet-50 at batch 2048 and the flower CNN at batch 1024: 13.1 -> 13.1 .. 13.3 k img/s and
# 65.9 -> 66.0 k img/s for depths 2 .. 16, i.e. within run-to-run noise -- at these sizes every kernel fills
# the chip by itself and a second stream only fills tile-quantisation tails.  (Pairs of one launch each, side by side:
# a memory-bound 1x1 data gradient next to a 3x3 weight gradient finishes 3 .. 9 % sooner than the two in a row, two
# memory-bound launches 11 .. 25 % later -- tools/experiments/overlap_probe.py; sending only the 3x3 weight gradients to the
# side stream still left the whole step unchanged.)  Off by default.
# The operands of an in-flight weight gradient are kept referenced (not ``record_stream``-ed: blocks marked that
# way return to the allocator only after an event query, and at tens of GB of activations the allocator then
# falls back to hipMalloc / hipFree, which cost 5x in step time when tried) until the main stream has been made
# to wait for the kernel; at most ``WGRAD_DEPTH`` layers are in flight, which bounds the memory held back.
WGRAD_STREAM = os.environ.get("DTFA_WGRAD_STREAM", "0") == "1"
WGRAD_DEPTH = int(os.environ.get("DTFA_WGRAD_DEPTH", "2"))
_wgrad_streams: dict = {}
_wgrad_pending: dict = {}   # device index -> list of (event, x, dy) in issue order


def _wgrad_stream(device):
    key = torch.device(device).index
    st = _wgrad_streams.get(key)
    if st is None:
        st = _wgrad_streams[key] = torch.cuda.Stream(device=device)
    return st


def join_wgrad_stream(device) -> None:
    """Make the current stream wait for every weight-gradient kernel enqueued on the side stream so far."""
    key = torch.device(device).index
    q = _wgrad_pending.get(key)
    if q:
        torch.cuda.current_stream(device).wait_event(q[-1][0])  # one stream: the last event covers all before it
        q.clear()


def _deliver_wgrad(param, x, dy, plan):
    mg = getattr(param, "main_grad", None)
    if mg is not None:
        acc = getattr(param, "_grad_filled", False)
        if WGRAD_STREAM and x.is_cuda:
            st = _wgrad_stream(x.device)
            cur = torch.cuda.current_stream(x.device)
            q = _wgrad_pending.setdefault(x.device.index, [])
            while len(q) >= WGRAD_DEPTH:   # oldest in-flight layer: the main stream waits for it, its operands go
                cur.wait_event(q.pop(0)[0])
            st.wait_event(cur.record_event())
            with torch.cuda.stream(st):
                cv.conv_wgrad(x, dy, plan, out=mg, accumulate=acc)
                done = st.record_event()
            q.append((done, x, dy))
        else:
            cv.conv_wgrad(x, dy, plan, out=mg, accumulate=acc)
        _notify(param)
        return None
    out = torch.empty(param.shape, dtype=param.dtype, device=param.device)
    cv.conv_wgrad(x, dy, plan, out=out)
    return out


class GradSlot:
    """Side channel between the backward passes of two layers of a residual block.

    In ``out = relu(bn(conv3(h)) + r)`` the gradient that flows into the residual ``r`` is
    ``g = dout * (out > 0)``.  Letting autograd materialise and sum such tensors costs full passes
    over the activation; instead layers hand each other what they need through a slot:

    * ``grad`` (+ ``mask``): a gradient tensor for another layer to add inside the epilogue of its
      data-gradient kernel; with ``mask`` (one bit per element) only the set elements are added, so
      ``g`` itself is never written -- the incoming ``dout`` and the stored sign bits are enough;
    * ``mask`` alone: the sign bits of the activation a layer's output was added into; that layer's
      batch-norm backward gates its incoming gradient with them;
    * ``stride`` (sy, sx) with ``grad``: the gradient lives on the pixels ``[::sy, ::sx]`` of the consumer's tensor
      (a strided shortcut read the same input); the consumer adds it there, or scatters it if it cannot.

    Producers run before consumers by data dependency, except for ``input_grad_out_slot`` (a sibling
    branch): there ``consumed`` tells the producer that the consumer already ran, and it falls back to
    returning its gradient through autograd."""

    __slots__ = ("grad", "mask", "consumed", "stride")

    def __init__(self):
        self.grad = None
        self.mask = None
        self.consumed = False
        self.stride = (1, 1)

    def take(self):
        g, m = self.grad, self.mask
        self.grad = self.mask = None
        self.consumed = True
        return g, m

    def take_strided(self):
        """``(grad, mask, stride)``; the slot is left empty."""
        stride, self.stride = self.stride, (1, 1)
        return self.take() + (stride,)


def _scatter_strided(d, stride, shape):
    """``d`` placed on the pixels ``[::sy, ::sx]`` of a zero tensor of ``shape``."""
    full = torch.zeros(shape, dtype=d.dtype, device=d.device)
    full[:, ::stride[0], ::stride[1], :] = d
    return full


# ----------------------------------------------------------------------------
# parameter fence
# ----------------------------------------------------------------------------
# An engine that refreshes parameters asynchronously (the all-gather of the sharded mode, the chunked pull of the
# GPU-resident parameter server) lets the refresh of the last layers' parameters run while the first layers already
# compute.  Every place that reads a layer's parameters calls ``param_fence(module)`` first; the engine's callback
# makes the current stream wait for exactly the refreshes that cover this module's parameters.  Without such an
# engine the call is one global lookup.
_param_fence_cb = None


def set_param_fence(cb) -> None:
    """``cb(modules)``, a bound method of the engine, or None.  One engine at a time owns the fence (the one created
    last); it is held weakly, so a discarded engine takes its fence with it."""
    global _param_fence_cb
    import weakref

    _param_fence_cb = None if cb is None else weakref.WeakMethod(cb)


def param_fence(*modules) -> None:
    global _param_fence_cb
    if _param_fence_cb is not None:
        cb = _param_fence_cb()
        if cb is None:
            _param_fence_cb = None
        else:
            cb(modules)


class Links:
    """The slots of one conv + batch-norm layer (all optional)."""

    __slots__ = ("residual_grad", "residual_mask", "input_grad", "output_mask", "input_grad_out", "output_tap",
                 "input_tap")

    def __init__(self, residual_grad=None, residual_mask=None, input_grad=None, output_mask=None,
                 input_grad_out=None, output_tap=None, input_tap=None):
        self.residual_grad = residual_grad      # park (dout, mask) of the residual for an upstream dgrad epilogue
        self.residual_mask = residual_mask      # park the mask for the layer that produced the residual
        self.input_grad = input_grad            # add this to the input gradient inside the dgrad epilogue
        self.output_mask = output_mask          # gate the incoming gradient with this mask
        self.input_grad_out = input_grad_out    # park the input gradient here instead of returning it
        # conv.BnBackwardTap shared by a layer (output_tap) and the only consumer of its output (input_tap):
        # the consumer's data-gradient kernel emits the layer's batch-norm backward partial sums
        self.output_tap = output_tap
        self.input_tap = input_tap


_NO_LINKS = Links()


# Independent branches of a block on separate HIP streams.  At small batches one convolution of an Inception
# branch is a few hundred workgroups at most -- one partial wave over the 256 CUs, ~20 us of mostly latency --
# so the four branches of a block are enqueued on four streams and run side by side; captured in a HIP graph
# the fork / join events become parallel graph branches (measured, Inception-v3 head training: +25 % at batch
# 32, +9 % at batch 256; serving 2.36 -> 1.82 ms at batch 1).  Eager launches at such sizes are bound by the
# host enqueueing kernels, where switching streams only adds cost (-24 % at batch 32), so by default the fork
# happens only while a graph is being captured (DTFA_BRANCH_PARALLEL=always|capture|never).  Above
# ``BRANCH_PARALLEL_MAX_ROWS`` rows per layer every kernel fills the chip on its own and the branches stay on
# one stream.
BRANCH_PARALLEL = os.environ.get("DTFA_BRANCH_PARALLEL", "capture")
BRANCH_PARALLEL_MAX_ROWS = int(os.environ.get("DTFA_BRANCH_PARALLEL_MAX_ROWS", str(1 << 20)))
_branch_streams: dict = {}


def branch_streams(device, n: int):
    key = torch.device(device).index
    pool = _branch_streams.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


def run_branches(x: torch.Tensor, jobs) -> None:
    """Run the callables ``jobs`` (each enqueues device work that reads ``x`` and writes memory of its own).
    While a HIP graph is being captured (and the tensors have few enough rows for the kernels not to fill the
    chip one by one): job 0 stays on the current stream, the others fork onto side streams and are joined back
    before returning.  Otherwise plain sequential calls."""
    rows = x.shape[0] * x.shape[1] * x.shape[2] if x.dim() == 4 else 0
    parallel = (x.is_cuda and len(jobs) > 1 and 0 < rows <= BRANCH_PARALLEL_MAX_ROWS and BRANCH_PARALLEL != "never"
                and (BRANCH_PARALLEL == "always" or torch.cuda.is_current_stream_capturing()))
    if not parallel:
        for job in jobs:
            job()
        return
    cur = torch.cuda.current_stream(x.device)
    fork = cur.record_event()
    side = branch_streams(x.device, len(jobs) - 1)
    jobs[0]()
    for st, job in zip(side, jobs[1:]):
        st.wait_event(fork)
        with torch.cuda.stream(st):
            job()
        cur.wait_event(st.record_event())


# ----------------------------------------------------------------------------
# batch statistics of small forward-only layers: fp64 accumulators instead of partial rows + a reduction launch
# ----------------------------------------------------------------------------
# A frozen trunk evaluated with batch statistics at a small batch (the reference's configuration: Inception-v3 built
# with is_training=True and fed 32 images, recognition_flower/single/flower_train.py:111,177-187) is ~94 layers
# of 10 .. 20 us convolutions; reducing each layer's partial statistics rows in a kernel of its own was a tenth of
# the step.  Inside a ``stat_scope`` such layers (at most STAT_ACC_MAX_ROWS output pixels, nothing requiring grad)
# let the convolution add its column sums to a slice of one zeroed fp64 buffer and the normalisation kernel derive
# its coefficients from the slice: two launches per layer, and one fill per forward pass.
STAT_ACC_MAX_ROWS = int(os.environ.get("DTFA_STAT_ACC_MAX_ROWS", str(1 << 16)))
_STAT_CHUNK = 1 << 17   # fp64 values per buffer (1 MB); Inception-v3 needs ~2 x 17 k, several times that in replicas


class _StatScope:
    """One zeroed fp64 buffer per forward pass.  The buffer is filled when the scope is entered, on the entering
    stream, so the fill precedes every fork of ``run_branches``; a stream other than the filling one waits for the
    fill's event before it hands a slice to a kernel (inside a graph capture that wait is the edge from the fill node
    to the branch's convolution nodes -- without it a replay may zero the buffer after a side branch has added its
    sums).  A buffer that runs out is replaced the same way, fill + event on the stream that needed it."""

    def __init__(self, device):
        self.device = device
        self.buf = None
        self.used = 0
        self.fill_stream = None
        self.fill_event = None
        self._fill(_STAT_CHUNK)

    def _fill(self, n: int) -> None:
        self.buf = torch.zeros(max(_STAT_CHUNK, n), dtype=torch.float64, device=self.device)
        self.used = 0
        self.fill_stream = torch.cuda.current_stream(self.device)
        self.fill_event = self.fill_stream.record_event()

    def take(self, n: int) -> torch.Tensor:
        if self.used + n > self.buf.numel():
            self._fill(n)
        cur = torch.cuda.current_stream(self.device)
        if cur != self.fill_stream:
            cur.wait_event(self.fill_event)
        out = self.buf[self.used:self.used + n]
        self.used += (n + 1) & ~1   # keep slices 16-byte aligned
        return out


_stat_scope: Optional[_StatScope] = None


class stat_scope:
    """Context manager: forward-only batch-statistics layers of few output pixels inside it use accumulator slices
    (see above).  Re-entrant use keeps the outer scope; a no-op off the device."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.mine = False

    def __enter__(self):
        global _stat_scope
        if _stat_scope is None and self.device.type == "cuda" and STAT_ACC_MAX_ROWS > 0:
            _stat_scope, self.mine = _StatScope(self.device), True
        return self

    def __exit__(self, *exc):
        global _stat_scope
        if self.mine:
            _stat_scope, self.mine = None, False
        return False


def _stat_slice(plan, rows: int):
    """An accumulator slice for a layer of ``rows`` output pixels, or None (no scope / too many rows / too wide)."""
    if _stat_scope is None or rows > STAT_ACC_MAX_ROWS or plan.cout > K.BN_SUMS_MAX_C:
        return None
    return _stat_scope.take(2 * plan.cout * cv.stat_replicas(rows))


def _conv_bn_forward_only(x, weight, gamma, beta, plan, run_mean, run_var, eps, momentum, relu, out):
    """Batch-statistics ``act(bn(conv(x)))`` with nothing kept for a backward pass."""
    rows = x.shape[0] * plan.oh * plan.ow
    acc = _stat_slice(plan, rows)
    if acc is not None:
        y, _ = cv.conv_fwd(x, weight, plan, stat_acc=acc)
        return K.bn_apply_sums(y, acc, rows, eps, gamma, beta, run_mean, run_var, momentum, relu, out=out)
    y, part = cv.conv_fwd(x, weight, plan, want_stats=True)
    _, _, scale, shift = K.bn_finalize(part, rows, eps, gamma, beta, run_mean, run_var, momentum)
    return K.bn_apply(y, scale, shift, relu, None, out=out)


# ----------------------------------------------------------------------------
# conv + batch-norm (+ residual) (+ ReLU)
# ----------------------------------------------------------------------------
class _ConvBnActFn(Function):
    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, plan, run_mean, run_var, eps, momentum, relu, training,
                params, links):
        need_grad = any(ctx.needs_input_grad)
        y, part = cv.conv_fwd(x, weight, plan, want_stats=training)
        M = y.numel() // plan.cout
        if training:
            mean, invstd, scale, shift = K.bn_finalize(part, M, eps, gamma, beta, run_mean, run_var, momentum)
        else:
            scale, shift = K.bn_eval_coeffs(gamma, beta, run_mean, run_var, eps)
            mean = invstd = None
            if (gamma is not None and ctx.needs_input_grad[2]) or ctx.needs_input_grad[3]:
                # fine-tuning on frozen statistics: the backward pass needs them for the scale's gradient
                mean = run_mean.detach().float().clone()
                invstd = torch.rsqrt(run_var.detach().float() + eps)
        # ReLU backward needs the sign of the output only.  Without a residual it is recomputed from y;
        # with one it is stored as a bitmask (1/16 of the activation's bytes) written by the same pass.
        want_mask = need_grad and training and relu and residual is not None
        if want_mask:
            z, mask = K.bn_apply(y, scale, shift, relu, residual, want_mask=True)
        else:
            z, mask = K.bn_apply(y, scale, shift, relu, residual), None
        ctx.plan, ctx.relu, ctx.training = plan, relu, training
        ctx.has_res = residual is not None
        ctx.params = params  # (weight_param, gamma_param, beta_param) python objects
        ctx.links = links
        if need_grad and training and links.output_tap is not None:
            tap = links.output_tap
            tap.y, tap.mean, tap.invstd, tap.scale, tap.shift, tap.mask = y, mean, invstd, scale, shift, mask
            tap.relu_mode = 0 if not relu else (3 if mask is not None else 2)
            tap.partials = None
        if need_grad:
            keep_z = relu and not training
            ctx.save_for_backward(x, weight, y, z if keep_z else None, mask, gamma, mean, invstd, scale, shift)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, weight, y, z, mask, gamma, mean, invstd, scale, shift = ctx.saved_tensors
        plan, links = ctx.plan, ctx.links
        wparam, gparam, bparam = ctx.params
        dz = dz.contiguous()
        want_dres = ctx.has_res and ctx.needs_input_grad[4]
        addend = addend_mask = None
        addend_stride = (1, 1)
        if links.input_grad is not None:
            addend, addend_mask, addend_stride = links.input_grad.take_strided()
            if addend is not None and addend_stride != (1, 1) and not (
                    addend_mask is None and cv.dgrad_coarse_addend_ok(plan, addend_stride)):
                addend, addend_stride = _scatter_strided(addend, addend_stride, x.shape), (1, 1)
        gate = mask
        if links.output_mask is not None and links.output_mask.mask is not None:
            if mask is not None:
                raise ValueError("a layer with its own residual ReLU cannot also be gated from outside")
            gate, links.output_mask.mask = links.output_mask.mask, None
        direct = False
        dres = None
        partials = None
        if ctx.training and links.output_tap is not None:
            if gate is mask:
                partials = links.output_tap.partials
            links.output_tap.clear()
        if ctx.training:
            # per-channel gradients go straight into the flat gradient buffer when both parameters live there
            g_mg = getattr(gparam, "main_grad", None) if (gamma is not None and ctx.needs_input_grad[2]) else None
            b_mg = getattr(bparam, "main_grad", None) if ctx.needs_input_grad[3] else None
            filled = getattr(bparam, "_grad_filled", False)
            direct = (b_mg is not None and b_mg.dtype == torch.float32
                      and (gamma is None or (g_mg is not None and g_mg.dtype == torch.float32
                                             and getattr(gparam, "_grad_filled", False) == filled)))
            dy, _, dgamma, dbeta = K.bn_backward(dz, y, None, mean, invstd, gamma, ctx.relu, False,
                                                 dgamma_out=g_mg if direct else None,
                                                 dbeta_out=b_mg if direct else None,
                                                 accumulate=filled if direct else False,
                                                 scale=scale, shift=shift, mask=gate, partials=partials)
            if want_dres:
                if mask is None:
                    dres = dz  # no activation between the sum and this layer's output
                elif links.residual_grad is not None:
                    links.residual_grad.grad, links.residual_grad.mask = dz, mask
                elif links.residual_mask is not None:
                    links.residual_mask.mask = mask
                    dres = dz  # the producer of the residual gates it with the parked mask
                else:
                    dres = K.mask_mul(dz, mask)
        else:
            g = K.relu_bwd(dz, z) if ctx.relu else dz
            if gate is not None and gate is not mask:
                g = K.mask_mul(g, gate)
            if want_dres:
                if links.residual_grad is not None:
                    links.residual_grad.grad = g
                else:
                    dres = g
            dy = (g.float() * scale).to(BF16)
            dgamma = dbeta = None
            if mean is not None:
                # z = gamma * (y - moving_mean) * moving_invstd + beta: sums of g and of g * xhat over the pixels
                dgamma, dbeta = K.bn_frozen_param_grads(g.contiguous(), y, mean, invstd)
        gx = gw = gg = gb = None
        if ctx.needs_input_grad[1]:
            gw = _deliver_wgrad(wparam if wparam is not None else weight, x, dy, plan)
        if direct:
            if gamma is not None:
                _notify(gparam)
            _notify(bparam)
        else:
            if gamma is not None and ctx.needs_input_grad[2]:
                gg = _deliver_small(gparam if gparam is not None else gamma, dgamma)
            if ctx.needs_input_grad[3]:
                gb = _deliver_small(bparam, dbeta) if bparam is not None else dbeta
        if ctx.needs_input_grad[0]:
            # the gradient written here is complete (nothing else flows into x) when every contribution
            # expected through the slot has arrived; then the upstream layer's reduction rides along
            tap = links.input_tap
            if tap is not None and not (tap.armed and cv.dgrad_can_tap(plan)
                                        and (links.input_grad is None or addend is not None)
                                        and (tap.gate_store or cv.dgrad_tap_pays(plan, dy.shape[0]))):
                tap = None
            gx = cv.conv_dgrad(dy, weight, plan, addend=addend, addend_mask=addend_mask, bn_tap=tap,
                               addend_stride=addend_stride)
            out_slot = links.input_grad_out
            if out_slot is not None and not out_slot.consumed:
                out_slot.grad, gx = gx, None  # a sibling consumer of x adds it in its own dgrad epilogue
        return gx, gw, gg, gb, dres, None, None, None, None, None, None, None, None, None


# ----------------------------------------------------------------------------
# expanding 1x1 conv + batch-norm + residual + ReLU with the normalisation folded THROUGH the convolution
# ----------------------------------------------------------------------------
# The tail of a bottleneck block, z = relu(bn(x W^T) + r) with x [M, C] and W [N, C], N = 4C, moves most of the
# block's bytes when done as conv -> stats -> apply: the N-wide convolution output y is written, re-read for the
# normalisation, kept for the backward pass and read there three more times.  But y is *linear* in the narrow
# tensor x, so everything the batch-norm needs of it can be had from C-wide quantities:
#
#   forward   S = x^T x (C x C), s = x^T 1:    mean = W s / M,  E[y^2]_n = w_n S w_n^T / M
#             -> scale a, shift b ->  ONE kernel: z = relu((x W^T) a + b + r), sign bits alongside; y never exists.
#   backward  with g = dz gated by the sign bits and P = g^T x (N x C, the ordinary weight-gradient GEMM on g):
#             dbeta = sum_m g,   sum_m g y = rowsum(W * P)  ->  dgamma and the coefficients of dy = c0 g + c1 y + c2
#             dW = dy^T x = diag(c0) P + diag(c1) W S + c2 s^T
#             dx = dy W   = g (diag(c0) W) + x (W^T diag(c1) W) + 1 (c2^T W)      -- one GEMM over [g | x]
#
# Traffic of the block tail in units of one C-wide tensor sweep: forward 17.25 -> ~11, backward 27 -> 12 (and the
# stored activations shrink by the N-wide y).  The gated gradient g is written directly by the data-gradient
# kernel that produces dz (BnBackwardTap.gate_store), which also emits sum_m g.
def _col_sums(x: torch.Tensor) -> torch.Tensor:
    return K.sum_partial_rows(K.bn_stats_partials(x))[0]


def _fold_stats(x, weight, gamma, beta, run_mean, run_var, eps, momentum):
    """Batch statistics of ``y = x W^T`` from the moments of ``x``: (W32, T = W S, s, mean, invstd, scale, shift);
    updates the running statistics."""
    N, C = weight.shape[0], x.shape[-1]
    M = x.numel() // C
    gplan = cv.get_plan(x.shape[1], x.shape[2], C, C, 1, 1, 1, 1, (0, 0, 0, 0), x.device)
    if cv.wgrad_colsum_ok(gplan, M):
        S, s = cv.conv_wgrad(x, x, gplan, colsum=True)   # x^T x and x^T 1 from one pass over x
        S = S.view(C, C)
    else:
        S = torch.empty((C, C), dtype=F32, device=x.device)
        cv.conv_wgrad(x, x, gplan, out=S)
        s = _col_sums(x)
    W32 = weight.reshape(N, C).float()
    T = W32 @ S
    mean, invstd, scale, shift = K.fold_forward(weight, T, s, gamma, beta, run_mean, run_var, M, eps, momentum)
    return W32, T, s, mean, invstd, scale, shift


def _grad_buffer(param, wanted: bool):
    """(fp32 flat-buffer view to write the gradient of ``param`` into, accumulate flag) or (None, False)."""
    mg = getattr(param, "main_grad", None) if (wanted and param is not None) else None
    if mg is not None and mg.dtype == F32 and mg.is_contiguous():
        return mg, getattr(param, "_grad_filled", False)
    return None, False


def _fold_backward(needs, params, x, g, sum_g, plan, weight, W32, T, s, mean, invstd, scale, gamma):
    """Backward of ``bn(x W^T)`` given the gradient ``g`` of its output and ``sum_g``, its column sums.
    ``needs`` = (dx, dW, dgamma, dbeta) wanted.  Gradients of parameters that live in a flat fp32 gradient buffer are
    written there by the kernel.  Returns ``(gw, gg, gb, wd, q)``: parameter gradients for autograd (None when
    delivered) and the matrix ([C, N + C] bf16) / bias ([C] fp32) of the data-gradient GEMM over ``[g | x]``."""
    need_x, need_w, need_g, need_b = needs
    need_g = need_g and gamma is not None
    N, C = plan.cout, plan.cin
    M = x.numel() // C
    wparam, gparam, bparam = params
    P = torch.empty((N, C), dtype=F32, device=x.device)
    cv.conv_wgrad(x, g, plan, out=P)
    wbuf, wacc = _grad_buffer(wparam, need_w)
    gbuf, gacc = _grad_buffer(gparam, need_g)
    bbuf, bacc = _grad_buffer(bparam, need_b)
    dgamma, dbeta, dW, wd, W1, c2 = K.fold_backward(weight, P, T, s, sum_g, mean, invstd, scale, M, gbuf, bbuf, wbuf,
                                                    gacc, bacc, wacc, want_dx=need_x)
    gw = gg = gb = None
    if need_w:
        if wbuf is not None:
            _notify(wparam)
        else:
            gw = _deliver_small(wparam if wparam is not None else weight, dW)
            if gw is not None:
                gw = gw.view(weight.shape)
    if need_g:
        if gbuf is not None:
            _notify(gparam)
        else:
            gg = _deliver_small(gparam if gparam is not None else gamma, dgamma)
    if need_b:
        if bbuf is not None:
            _notify(bparam)
        else:
            gb = _deliver_small(bparam, dbeta) if bparam is not None else dbeta
    q = None
    if need_x:
        wd[:, N:].copy_(W32.t() @ W1)   # Q = W^T diag(c1) W (symmetric)
        q = c2 @ W32
    return gw, gg, gb, wd, q


def _gated_gradient(dz, mask, tap):
    """``(g, sum_m g)``: the incoming gradient gated by the sign bits -- taken as is when the kernel that produced
    ``dz`` already stored it gated and summed it (BnBackwardTap.gate_store), else computed here."""
    if tap is not None and tap.gated and tap.partials is not None:
        g = dz
        sum_g = K.sum_partial_rows(tap.partials)[0]
    else:
        g = K.mask_mul(dz, mask)
        sum_g = _col_sums(g)
    if tap is not None:
        tap.clear()
    return g, sum_g


class _ExpandBnResFn(Function):
    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, plan, run_mean, run_var, eps, momentum, params, links):
        need_grad = any(ctx.needs_input_grad)
        N, C = plan.cout, plan.cin
        M = x.numel() // C
        W32, T, s, mean, invstd, scale, shift = _fold_stats(x, weight, gamma, beta, run_mean, run_var, eps, momentum)
        mask = torch.empty(M * N // 8, dtype=torch.uint8, device=x.device) if need_grad else None
        z, _ = cv.conv_fwd(x, weight, plan, bias=shift, relu=True, scale=scale, residual=residual, mask_out=mask)
        ctx.plan, ctx.params, ctx.links = plan, params, links
        ctx.has_res = residual is not None
        if need_grad and links.output_tap is not None:
            tap = links.output_tap
            tap.clear()
            tap.mask, tap.relu_mode, tap.gate_store = mask, 3, True
        if need_grad:
            ctx.save_for_backward(x, weight, mask, gamma, mean, invstd, scale, T, s, W32)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, weight, mask, gamma, mean, invstd, scale, T, s, W32 = ctx.saved_tensors
        plan, links = ctx.plan, ctx.links
        g, sum_g = _gated_gradient(dz.contiguous(), mask, links.output_tap)
        dres = None
        if ctx.has_res and ctx.needs_input_grad[4]:
            if links.residual_grad is not None:
                links.residual_grad.grad, links.residual_grad.mask = g, None  # already gated: added without a mask
            else:
                dres = g   # (a producer of the residual that waits for sign bits in ``residual_mask`` gets none)
        gw, gg, gb, wd, q = _fold_backward(ctx.needs_input_grad[:4], ctx.params, x, g, sum_g, plan, weight, W32, T, s,
                                           mean, invstd, scale, gamma)
        gx = None
        if ctx.needs_input_grad[0]:
            itap = links.input_tap
            if itap is not None and not (itap.armed and cv.dgrad_can_tap(plan)):
                itap = None
            gx = cv.conv_dgrad(g, weight, plan, bn_tap=itap, wd=wd, x2=x, dbias=q)
        return gx, gw, gg, gb, dres, None, None, None, None, None, None, None


class _ExpandDualFn(Function):
    """``relu(bn(x W^T) + bn_s(x_s W_s^T))``: the tail of a projection block, both batch-norms folded through
    their 1x1 convolutions and the two convolutions summed in ONE GEMM over ``[x | x_s]`` (``x_s`` = the block input
    at the shortcut's stride).  Neither convolution output nor the shortcut activation exists in memory."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, xs_full, weight_s, gamma_s, beta_s, plan, stride_s, run, run_s, eps,
                momentum, params, params_s, links):
        need_grad = any(ctx.needs_input_grad)
        sy, sx = stride_s
        if sy == 1 and sx == 1:
            xs = xs_full
        elif xs_full.is_contiguous() and xs_full.dtype == BF16:
            xs = K.subsample(xs_full, (sy, sx))   # (the tensor library's strided copy runs at less than half the rate)
        else:
            xs = xs_full[:, ::sy, ::sx, :].contiguous()
        N, C, Cs = plan.cout, plan.cin, xs.shape[-1]
        M = x.numel() // C
        if tuple(xs.shape[:3]) != tuple(x.shape[:3]):
            raise ValueError("shortcut input %s does not land on the block's output grid %s" % (tuple(xs.shape), tuple(x.shape)))
        W32, T, s, mean, invstd, scale, shift = _fold_stats(x, weight, gamma, beta, run[0], run[1], eps, momentum)
        Ws32, Ts, ss, mean_s, invstd_s, scale_s, shift_s = _fold_stats(xs, weight_s, gamma_s, beta_s, run_s[0], run_s[1],
                                                                       eps, momentum)
        wcat = torch.cat((scale[:, None] * W32, scale_s[:, None] * Ws32), dim=1).to(BF16).contiguous()
        mask = torch.empty(M * N // 8, dtype=torch.uint8, device=x.device) if need_grad else None
        z, _ = cv.conv_fwd(x, wcat, plan, bias=shift + shift_s, relu=True, x2=xs, mask_out=mask)
        ctx.plan, ctx.params, ctx.params_s, ctx.links, ctx.stride_s = plan, params, params_s, links, (sy, sx)
        ctx.full_shape = tuple(xs_full.shape)
        if need_grad and links.output_tap is not None:
            tap = links.output_tap
            tap.clear()
            tap.mask, tap.relu_mode, tap.gate_store = mask, 3, True
        if need_grad:
            ctx.save_for_backward(x, weight, gamma, mean, invstd, scale, T, s,
                                  xs, weight_s, gamma_s, mean_s, invstd_s, scale_s, Ts, ss, mask, W32, Ws32)
        return z

    @staticmethod
    def backward(ctx, dz):
        (x, weight, gamma, mean, invstd, scale, T, s,
         xs, weight_s, gamma_s, mean_s, invstd_s, scale_s, Ts, ss, mask, W32, Ws32) = ctx.saved_tensors
        plan, links = ctx.plan, ctx.links
        sy, sx = ctx.stride_s
        g, sum_g = _gated_gradient(dz.contiguous(), mask, links.output_tap)
        need = ctx.needs_input_grad
        # main branch, then the shortcut branch (same g)
        gw, gg, gb, wd, q = _fold_backward(need[:4], ctx.params, x, g, sum_g, plan, weight, W32, T, s, mean, invstd,
                                           scale, gamma)
        Cs = xs.shape[-1]
        plan_s = cv.get_plan(plan.ih, plan.iw, Cs, plan.cout, 1, 1, 1, 1, (0, 0, 0, 0), x.device)
        gws, ggs, gbs, wds, qs = _fold_backward(need[4:8], ctx.params_s, xs, g, sum_g, plan_s, weight_s, Ws32, Ts, ss,
                                                mean_s, invstd_s, scale_s, gamma_s)
        gx = gxs = None
        if need[4]:
            gxs = cv.conv_dgrad(g, weight_s, plan_s, wd=wds, x2=xs, dbias=qs)
            slot = links.input_grad_out
            if slot is not None and not slot.consumed:
                # the sibling consumer of the block input adds it, on the shortcut's pixel grid, in its dgrad epilogue
                slot.grad, slot.stride, gxs = gxs, (sy, sx), None
            elif (sy, sx) != (1, 1):
                gxs = _scatter_strided(gxs, (sy, sx), ctx.full_shape)
        if need[0]:
            itap = links.input_tap
            if itap is not None and not (itap.armed and cv.dgrad_can_tap(plan)):
                itap = None
            gx = cv.conv_dgrad(g, weight, plan, bn_tap=itap, wd=wd, x2=x, dbias=q)
        return (gx, gw, gg, gb, gxs, gws, ggs, gbs) + (None,) * 9


EXPAND_FOLD = os.environ.get("DTFA_EXPAND_FOLD", "1") == "1"


def _expand_fold_ok(plan, x, residual, relu, training, links) -> bool:
    """The folded form needs a 1x1 / stride-1 layer that widens (C -> N > C: the C x C moments are then the cheap
    side), a ReLU behind the sum, batch statistics, and only the slots it implements."""
    return (EXPAND_FOLD and training and relu and residual is not None and plan.is_pointwise and plan.cout > plan.cin
            and links.input_grad is None and links.output_mask is None and links.input_grad_out is None)



POOL_WINNERS = os.environ.get("DTFA_POOL_WINNERS", "1") == "1"


class _ConvBnActPoolFn(Function):
    """conv -> batch-norm -> ReLU -> max-pool; the full-resolution activation is never stored."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, plan, run_mean, run_var, eps, momentum, relu, training, params, pool):
        need_grad = any(ctx.needs_input_grad)
        y, part = cv.conv_fwd(x, weight, plan, want_stats=training)
        M = y.numel() // plan.cout
        if training:
            mean, invstd, scale, shift = K.bn_finalize(part, M, eps, gamma, beta, run_mean, run_var, momentum)
        else:
            scale, shift = K.bn_eval_coeffs(gamma, beta, run_mean, run_var, eps)
            mean = invstd = None
        k, stride, pads = pool
        ywin = None
        if need_grad and training and POOL_WINNERS:
            # the winners' values: a quarter of y's bytes (stride-2 pool) written once, read once by the backward sums
            out, idx, ywin = K.bn_relu_maxpool_fwd(y, scale, shift, relu, k, stride, pads, want_winners=True)
        else:
            out, idx = K.bn_relu_maxpool_fwd(y, scale, shift, relu, k, stride, pads, want_index=need_grad)
        ctx.plan, ctx.relu, ctx.training, ctx.params, ctx.pool = plan, relu, training, params, pool
        if need_grad:
            ctx.save_for_backward(x, weight, y, idx, gamma, mean, invstd, scale, shift, ywin)
        return out

    @staticmethod
    def backward(ctx, dp):
        x, weight, y, idx, gamma, mean, invstd, scale, shift, ywin = ctx.saved_tensors
        plan = ctx.plan
        wparam, gparam, bparam = ctx.params
        k, stride, pads = ctx.pool
        dp = dp.contiguous()
        if not ctx.training:
            raise NotImplementedError("backward through a frozen-statistics conv+pool block")
        g_mg = getattr(gparam, "main_grad", None) if (gamma is not None and ctx.needs_input_grad[2]) else None
        b_mg = getattr(bparam, "main_grad", None) if ctx.needs_input_grad[3] else None
        filled = getattr(bparam, "_grad_filled", False)
        direct = (b_mg is not None and b_mg.dtype == torch.float32
                  and (gamma is None or (g_mg is not None and g_mg.dtype == torch.float32
                                         and getattr(gparam, "_grad_filled", False) == filled)))
        dy, dgamma, dbeta = K.bn_pool_backward(dp, idx, y, mean, invstd, gamma, scale, shift, ctx.relu, k, stride, pads,
                                               dgamma_out=g_mg if direct else None, dbeta_out=b_mg if direct else None,
                                               accumulate=filled if direct else False, ywin=ywin)
        gx = gw = gg = gb = None
        if ctx.needs_input_grad[1]:
            gw = _deliver_wgrad(wparam if wparam is not None else weight, x, dy, plan)
        if direct:
            if gamma is not None:
                _notify(gparam)
            _notify(bparam)
        else:
            if gamma is not None and ctx.needs_input_grad[2]:
                gg = _deliver_small(gparam if gparam is not None else gamma, dgamma)
            if ctx.needs_input_grad[3]:
                gb = _deliver_small(bparam, dbeta) if bparam is not None else dbeta
        if ctx.needs_input_grad[0]:
            gx = cv.conv_dgrad(dy, weight, plan)
        return gx, gw, gg, gb, None, None, None, None, None, None, None, None, None


def conv_bn_act_pool(x, weight, gamma, beta, run_mean, run_var, *, stride, padding, eps, momentum, relu, training,
                     pool_kernel=(3, 3), pool_stride=(2, 2), pool_padding="VALID"):
    """``maxpool(act(bn(conv(x, weight))))``.  On the device the pool reads the convolution output and
    applies the normalisation and the activation on the fly (csrc/hip/bn_pool.hip)."""
    if not _use_hip(x):
        z = conv_bn_act(x, weight, gamma, beta, run_mean, run_var, stride=stride, padding=padding, eps=eps,
                        momentum=momentum, relu=relu, training=training)
        return max_pool(z, pool_kernel, pool_stride, pool_padding)
    kh, kw = weight.shape[1], weight.shape[2]
    sy, sx = stride
    pads = cv.resolve_pads(padding, x.shape[1], x.shape[2], kh, kw, sy, sx)
    plan = cv.get_plan(x.shape[1], x.shape[2], x.shape[3], weight.shape[0], kh, kw, sy, sx, pads, x.device)
    ppads = cv.resolve_pads(pool_padding, plan.oh, plan.ow, pool_kernel[0], pool_kernel[1], pool_stride[0],
                            pool_stride[1])
    if _needs_grad(x, weight, gamma, beta) and not training:
        z = conv_bn_act(x, weight, gamma, beta, run_mean, run_var, stride=stride, padding=padding, eps=eps,
                        momentum=momentum, relu=relu, training=training)
        return max_pool(z, pool_kernel, pool_stride, pool_padding)
    return _ConvBnActPoolFn.apply(x, weight, gamma, beta, plan, run_mean, run_var, eps, momentum, relu, training,
                                  (weight, gamma, beta), (tuple(pool_kernel), tuple(pool_stride), tuple(ppads)))


def conv_bn_act(x, weight, gamma, beta, run_mean, run_var, *, stride, padding, eps, momentum, relu, training,
                residual=None, links=None, out=None, eval_coeffs=None):
    """``act(bn(conv(x, weight)) + residual)``; batch statistics when ``training``.
    ``links`` (:class:`Links`) wires the layer's backward pass to its neighbours in a residual block.

    Inference (``training`` false, nothing requiring grad) on the device is ONE kernel: the normalisation is
    folded into the convolution's epilogue as a per-channel multiply-add on the fp32 accumulators, a residual
    is added there before the activation.  ``eval_coeffs`` = cached ``(scale, shift)`` of the running statistics (computed if absent).

    ``out`` (device path, nothing requiring grad) is a channel slice of a wider NHWC tensor that
    receives the result directly -- the way parallel branches are concatenated without a copy."""
    kh, kw = weight.shape[1], weight.shape[2]
    sy, sx = stride
    pads = cv.resolve_pads(padding, x.shape[1], x.shape[2], kh, kw, sy, sx)
    if out is not None:
        if not _use_hip(x) or _needs_grad(x, weight, gamma, beta, residual):
            raise ValueError("out= is a forward-only device path")
        plan = cv.get_plan(x.shape[1], x.shape[2], x.shape[3], weight.shape[0], kh, kw, sy, sx, pads, x.device)
        if not training:
            scale, shift = eval_coeffs or K.bn_eval_coeffs(gamma, beta, run_mean, run_var, eps)
            return cv.conv_fwd(x, weight, plan, bias=shift, relu=relu, scale=scale, out=out, residual=residual)[0]
        if residual is None:
            return _conv_bn_forward_only(x, weight, gamma, beta, plan, run_mean, run_var, eps, momentum, relu, out)
        y, part = cv.conv_fwd(x, weight, plan, want_stats=True)
        _, _, scale, shift = K.bn_finalize(part, y.numel() // plan.cout, eps, gamma, beta, run_mean, run_var, momentum)
        return K.bn_apply(y, scale, shift, relu, residual, out=out)
    if not _use_hip(x):
        y = ref.conv2d(x, weight, (sy, sx), pads)
        if training:
            z, _, _ = ref.batch_norm_train(y, gamma, beta, run_mean, run_var, eps, momentum)
        else:
            z = ref.batch_norm_eval(y, gamma, beta, run_mean, run_var, eps)
        if residual is not None:
            z = z + residual
        return torch.relu(z) if relu else z
    plan = cv.get_plan(x.shape[1], x.shape[2], x.shape[3], weight.shape[0], kh, kw, sy, sx, pads, x.device)
    if not training and not _needs_grad(x, weight, gamma, beta, residual):
        scale, shift = eval_coeffs or K.bn_eval_coeffs(gamma, beta, run_mean, run_var, eps)
        res = residual.contiguous() if residual is not None else None
        return cv.conv_fwd(x, weight, plan, bias=shift, relu=relu, scale=scale, residual=res)[0]
    if training and residual is None and links is None and not _needs_grad(x, weight, gamma, beta):
        return _conv_bn_forward_only(x, weight, gamma, beta, plan, run_mean, run_var, eps, momentum, relu, None)
    links = links if links is not None else _NO_LINKS
    if links.input_grad is not None and (plan.dgrad_needs_zero or not x.requires_grad):
        raise ValueError("input_grad needs an input that requires grad and a dgrad that writes every pixel")
    if _expand_fold_ok(plan, x, residual, relu, training, links):
        return _ExpandBnResFn.apply(x, weight, gamma, beta, residual.contiguous(), plan, run_mean, run_var, eps,
                                    momentum, (weight, gamma, beta), links)
    return _ConvBnActFn.apply(x, weight, gamma, beta, residual, plan, run_mean, run_var, eps, momentum, relu,
                              training, (weight, gamma, beta), links)


def expand_dual_ok(main, shortcut, x, x_in, training: bool) -> bool:
    """True when the tail of a projection block (``main`` = the expanding 1x1 layer, ``shortcut`` = the strided 1x1
    projection of the block input, both ``ConvBnAct``-like modules) can run as one folded GEMM on the device."""
    return (EXPAND_FOLD and training and _use_hip(x) and torch.is_grad_enabled()
            and main.weight.shape[1] == 1 and main.weight.shape[2] == 1 and main.stride == (1, 1)
            and shortcut.weight.shape[1] == 1 and shortcut.weight.shape[2] == 1
            and main.relu and not shortcut.relu and main.weight.shape[0] > main.weight.shape[3]
            and main.weight.shape[0] == shortcut.weight.shape[0] and x_in.shape[-1] % 8 == 0)


def expand_dual(x, main, x_in, shortcut, links=None):
    """``relu(main.bn(main.conv(x)) + shortcut.bn(shortcut.conv(x_in)))`` for 1x1 layers (see ``_ExpandDualFn``).
    ``links``: ``input_tap`` / ``output_tap`` as for ``conv_bn_act``; ``input_grad_out`` receives the gradient of
    ``x_in`` through the shortcut (for the data-gradient epilogue of the other consumer of ``x_in``)."""
    links = links if links is not None else _NO_LINKS
    plan = cv.get_plan(x.shape[1], x.shape[2], x.shape[3], main.weight.shape[0], 1, 1, 1, 1, (0, 0, 0, 0), x.device)
    param_fence(main, shortcut)
    return _ExpandDualFn.apply(x, main.weight, main.gamma, main.beta, x_in, shortcut.weight, shortcut.gamma,
                               shortcut.beta, plan, tuple(shortcut.stride), (main.running_mean, main.running_var),
                               (shortcut.running_mean, shortcut.running_var), main.eps, main.momentum,
                               (main.weight, main.gamma, main.beta), (shortcut.weight, shortcut.gamma, shortcut.beta),
                               links)


# ----------------------------------------------------------------------------
# plain conv (+bias) (+ReLU)
# ----------------------------------------------------------------------------
class _ConvFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, plan, relu, params):
        rows = x.shape[0] * x.shape[1] * x.shape[2]
        if plan.is_pointwise and cv.dense_skinny_ok(rows, plan.cin, plan.cout):
            # a dense layer (or 1x1 convolution) over a handful of rows
            y = cv.dense_skinny(x.reshape(rows, plan.cin), weight.reshape(plan.cout, plan.cin), bias, relu)
            y = y.view(x.shape[0], x.shape[1], x.shape[2], plan.cout)
        else:
            y, _ = cv.conv_fwd(x, weight, plan, bias=bias, relu=relu)
        ctx.plan, ctx.relu, ctx.params = plan, relu, params
        ctx.save_for_backward(x, weight, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        plan = ctx.plan
        wparam, bparam = ctx.params
        dy = dy.contiguous()
        if ctx.relu:
            dy = K.relu_bwd(dy, y)
        gx = gw = gb = None
        if ctx.needs_input_grad[1]:
            gw = _deliver_wgrad(wparam, x, dy, plan)
            if gw is not None:
                gw = gw.view(weight.shape)
        if bparam is not None and ctx.needs_input_grad[2]:
            gb = _deliver_small(bparam, dy.reshape(-1, plan.cout).float().sum(0))
        if ctx.needs_input_grad[0]:
            gx = cv.conv_dgrad(dy, weight, plan)
        return gx, gw, gb, None, None, None


def conv2d(x, weight, bias=None, *, stride=(1, 1), padding="VALID", relu=False):
    kh, kw = weight.shape[1], weight.shape[2]
    sy, sx = stride
    pads = cv.resolve_pads(padding, x.shape[1], x.shape[2], kh, kw, sy, sx)
    if not _use_hip(x):
        return ref.conv2d(x, weight, (sy, sx), pads, bias, relu)
    plan = cv.get_plan(x.shape[1], x.shape[2], x.shape[3], weight.shape[0], kh, kw, sy, sx, pads, x.device)
    return _ConvFn.apply(x, weight, bias, plan, relu, (weight, bias))


def dense(x, weight, bias=None, *, relu=False):
    """``x [B, In] @ weight[Out, In]^T + bias``; In and Out multiples of 8 on the GPU."""
    if not _use_hip(x):
        return ref.dense(x, weight, bias, relu)
    B, n_in = x.shape
    n_out = weight.shape[0]
    plan = cv.get_plan(1, 1, n_in, n_out, 1, 1, 1, 1, (0, 0, 0, 0), x.device)
    y = _ConvFn.apply(x.view(B, 1, 1, n_in), weight.view(n_out, 1, 1, n_in), bias, plan, relu, (weight, bias))
    return y.view(B, n_out)


# ----------------------------------------------------------------------------
# pooling
# ----------------------------------------------------------------------------
class _MaxPoolFn(Function):
    @staticmethod
    def forward(ctx, x, k, stride, pads):
        need = ctx.needs_input_grad[0]
        y, idx = K.maxpool_fwd(x, k, stride, pads, want_index=need)
        ctx.cfg = (tuple(x.shape), k, stride, pads)
        if need:
            ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        shape, k, stride, pads = ctx.cfg
        return K.maxpool_bwd(dy.contiguous(), idx, shape, k, stride, pads), None, None, None


class _AvgPoolFn(Function):
    @staticmethod
    def forward(ctx, x, k, stride, pads, include_pad):
        ctx.cfg = (tuple(x.shape), k, stride, pads, include_pad)
        return K.avgpool_fwd(x, k, stride, pads, include_pad)

    @staticmethod
    def backward(ctx, dy):
        shape, k, stride, pads, include_pad = ctx.cfg
        return K.avgpool_bwd(dy.contiguous(), shape, k, stride, pads, include_pad), None, None, None, None


class _GapFn(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        return K.gap_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        return K.gap_bwd(dy.contiguous(), ctx.shape)


def _pool_pads(x, k, stride, padding):
    return cv.resolve_pads(padding, x.shape[1], x.shape[2], k[0], k[1], stride[0], stride[1])


def max_pool(x, k=(3, 3), stride=(2, 2), padding="VALID"):
    pads = _pool_pads(x, k, stride, padding)
    if not _use_hip(x):
        return ref.max_pool(x, k, stride, pads)
    return _MaxPoolFn.apply(x, tuple(k), tuple(stride), pads)


def avg_pool(x, k=(3, 3), stride=(1, 1), padding="SAME", count_include_pad=False):
    pads = _pool_pads(x, k, stride, padding)
    if not _use_hip(x):
        return ref.avg_pool(x, k, stride, pads, count_include_pad)
    return _AvgPoolFn.apply(x, tuple(k), tuple(stride), pads, count_include_pad)


def global_avg_pool(x):
    if not _use_hip(x):
        return ref.global_avg_pool(x)
    return _GapFn.apply(x)


# ----------------------------------------------------------------------------
# dropout
# ----------------------------------------------------------------------------
class _DropoutFn(Function):
    @staticmethod
    def forward(ctx, x, rate, seed):
        ctx.rate, ctx.seed = rate, seed
        return K.dropout(x.contiguous(), rate, seed)

    @staticmethod
    def backward(ctx, dy):
        return K.dropout(dy.contiguous(), ctx.rate, ctx.seed), None, None


_dropout_counter = [0]


def dropout(x, rate: float, training: bool, seed: Optional[int] = None):
    """Inverted dropout with *drop* probability ``rate``."""
    if not training or rate <= 0.0:
        return x
    if not _use_hip(x):
        return ref.dropout(x, rate, training)
    if seed is None:
        _dropout_counter[0] += 1
        seed = (torch.initial_seed() * 0x9E3779B97F4A7C15 + _dropout_counter[0] * 0xD1B54A32D192ED03) & ((1 << 64) - 1)
    return _DropoutFn.apply(x, rate, seed)


# ----------------------------------------------------------------------------
# activity regulariser
# ----------------------------------------------------------------------------
class _ActL2Fn(Function):
    """``(h, coef/2 * sum(h^2))``: the activation passes through, so that its two gradients -- from the layers
    above and from the regulariser -- are combined in one kernel instead of autograd's separate multiply and add."""

    @staticmethod
    def forward(ctx, h, coef):
        ctx.coef = coef
        ctx.save_for_backward(h)
        return h.view_as(h), K.act_l2_fwd(h, coef)[0]

    @staticmethod
    def backward(ctx, dh, dreg):
        (h,) = ctx.saved_tensors
        if dreg is None:
            return dh, None
        g = dreg.detach().reshape(1).to(F32)
        return K.act_l2_bwd(h, dh.contiguous() if dh is not None else None, g, ctx.coef), None


def activity_l2(h, coef: float):
    """L2 activity regulariser of the reference's head (``l2_regularizer(coef)(h)`` = ``coef * sum(h^2) / 2`` over
    the whole batch; recognition_flower/single/flower_train.py:71,77).  Returns ``(h, reg)``: use the returned
    ``h`` downstream."""
    if not _use_hip(h) or h.numel() % 8 or not h.is_contiguous():
        return h, 0.5 * coef * (h.float() ** 2).sum()
    return _ActL2Fn.apply(h, coef)


# ----------------------------------------------------------------------------
# loss
# ----------------------------------------------------------------------------
class _XentFn(Function):
    @staticmethod
    def forward(ctx, logits, labels, n_classes, act_l2):
        B = logits.shape[0]
        # the kernel folds the activity term into the per-row loss, which is then averaged: B * act_l2 per row
        # makes it the batch SUM the regulariser is defined as
        loss, dl, top1, score, correct = K.softmax_xent(logits, labels, n_classes, 1.0 / B, act_l2 * B)
        ctx.save_for_backward(dl)
        ctx.mark_non_differentiable(top1, score, correct)
        return K.mean_f32(loss)[0], top1, score, correct

    @staticmethod
    def backward(ctx, g, *_):
        (dl,) = ctx.saved_tensors
        # dl already holds d(mean loss)/d(logits); g is the upstream scalar (1.0 for a plain .backward()).
        return (dl.float() * g).to(BF16), None, None, None


def softmax_xent(logits, labels, n_classes: Optional[int] = None, act_l2: float = 0.0):
    """Mean sparse softmax cross-entropy, plus ``act_l2 * sum(logits**2) / 2`` over the WHOLE batch: the
    reference adds ``l2_regularizer(0.0005)(flower_logits)`` = ``0.0005 * tf.nn.l2_loss(logits)``, a batch sum,
    un-averaged, to ``reduce_mean(xent)`` (recognition_flower/single/flower_train.py:84,143-145).
    Returns ``(loss, top1, top1_score, correct)``."""
    n = n_classes if n_classes is not None else logits.shape[-1]
    if not _use_hip(logits):
        loss, top1, score, correct = ref.softmax_xent(logits, labels, n, act_l2 * logits.shape[0])
        return loss.mean(), top1, score, correct
    return _XentFn.apply(logits.contiguous(), labels.to(torch.int32).contiguous(), n, act_l2)
