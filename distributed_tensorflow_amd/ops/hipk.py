"""Low-level bindings to ``libdtfa_hip.so``.

Every function here takes torch CUDA(=HIP) tensors, checks the shapes / dtypes /
contiguity the kernel and its grid assume *on the host* (a faulting kernel can
reset the whole node), and launches on the current torch stream.  Nothing in
this module falls back to another implementation: a missing library or a bad
argument raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from .. import _lib

c_void = C.c_void_p
c_int = C.c_int
c_ll = C.c_longlong
c_float = C.c_float


class ConvArgs(C.Structure):
    _fields_ = [
        ("x", c_void), ("w", c_void), ("y", c_void), ("bias", c_void), ("addend", c_void), ("addend_mask", c_void), ("stats", c_void),
        ("zero", c_void), ("taps", c_void),
        ("IH", c_int), ("IW", c_int), ("Cin", c_int),
        ("OH", c_int), ("OW", c_int), ("Cout", c_int),
        ("OYn", c_int), ("OXn", c_int),
        ("osy", c_int), ("osx", c_int), ("oy0", c_int), ("ox0", c_int),
        ("isy", c_int), ("isx", c_int),
        ("T", c_int), ("M", c_int), ("K", c_int),
        ("cin_magic", C.c_uint), ("relu", c_int), ("dense_out", c_int), ("pointwise", c_int),
        ("bn_y", c_void), ("bn_mean", c_void), ("bn_invstd", c_void), ("bn_scale", c_void), ("bn_shift", c_void),
        ("bn_mask", c_void), ("bn_relu", c_int),
        ("ldo", c_int), ("oscale", c_void), ("stream_out", c_int), ("omask", c_void), ("dbias", c_void), ("bn_gate_store", c_int),
        ("x2", c_void), ("K1", c_int), ("addend_shy", c_int), ("addend_shx", c_int), ("stat_acc", c_void),
        ("tile_m_base", c_int), ("tile_m_count", c_int), ("stat_row_base", c_int), ("stat_rep", c_int),
    ]


class WgradArgs(C.Structure):
    _fields_ = [
        ("x", c_void), ("dy", c_void), ("slab", c_void), ("zero", c_void), ("taps", c_void),
        ("IH", c_int), ("IW", c_int), ("Cin", c_int),
        ("OH", c_int), ("OW", c_int), ("Cout", c_int),
        ("isy", c_int), ("isx", c_int),
        ("T", c_int), ("M", c_int), ("K", c_int),
        ("cin_magic", C.c_uint), ("ktiles_per_split", c_int), ("splits", c_int),
        ("step_b", c_int), ("step_y", c_int), ("step_x", c_int),
        ("slab_stride", c_ll), ("colsum", c_int),
    ]


class SkinnyArgs(C.Structure):
    _fields_ = [("x", c_void), ("w", c_void), ("bias", c_void), ("y", c_void), ("M", c_int), ("N", c_int),
                ("K", c_int), ("relu", c_int)]


class PoolArgs(C.Structure):
    _fields_ = [
        ("B", c_int), ("IH", c_int), ("IW", c_int), ("C", c_int), ("OH", c_int), ("OW", c_int),
        ("kh", c_int), ("kw", c_int), ("sy", c_int), ("sx", c_int), ("pt", c_int), ("pl", c_int),
        ("count_include_pad", c_int),
    ]


_sigs_done = False


def lib():
    global _sigs_done
    L = _lib.hip()
    if not _sigs_done:
        for name in dir(_Sigs):
            if name.startswith("dtfa_"):
                fn = getattr(L, name)
                fn.argtypes = getattr(_Sigs, name)
                fn.restype = c_int
        _sigs_done = True
    return L


class _Sigs:
    dtfa_conv_igemm = [C.POINTER(ConvArgs), c_int, c_int, c_int, c_int, c_void]
    dtfa_conv_igemm_ks = [C.POINTER(ConvArgs), c_int, c_int, c_int, c_int, c_int, c_void]
    dtfa_conv_p8 = [C.POINTER(ConvArgs), c_int, c_int, c_void]
    dtfa_conv_p8_buffer = [c_int, c_int, c_int, c_int, c_int, c_int]
    dtfa_conv_stats_rows = [c_int, c_int]
    dtfa_dense_skinny = [C.POINTER(SkinnyArgs), c_void]
    dtfa_conv_wgrad = [C.POINTER(WgradArgs), c_int, c_int, c_int, c_int, c_void]
    dtfa_conv_wp3 = [C.POINTER(WgradArgs), c_void]
    dtfa_conv_wp3_slabs = [c_int]
    dtfa_conv_wp8 = [C.POINTER(WgradArgs), c_int, c_int, c_void]
    dtfa_conv_wp8_slabs = [c_int]
    dtfa_wgrad_reduce = [c_void, c_int, c_ll, c_void, c_void, c_int, c_void]
    dtfa_partial_reduce = [c_void, c_void, c_int, c_int, c_int, c_int, c_void]
    dtfa_bn_finalize = [c_void, c_int, c_int, c_float, c_float, c_void, c_void, c_void, c_void, c_void, c_void,
                        c_void, c_void, c_float, c_void]
    dtfa_bn_eval_coeffs = [c_int, c_float, c_void, c_void, c_void, c_void, c_void, c_void, c_void]
    dtfa_bn_apply = [c_void, c_void, c_void, c_void, c_void, c_void, c_ll, c_int, c_int, c_int, c_void]
    dtfa_bn_apply_sums = [c_void, c_void, c_void, c_float, c_float, c_void, c_void, c_void, c_void, c_float, c_ll, c_int,
                          c_int, c_int, c_int, c_void]
    dtfa_col_partials_rows = [c_ll, c_int]
    dtfa_bn_stats_partials = [c_void, c_void, c_ll, c_int, c_void]
    dtfa_bn_bwd_partials = [c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_ll, c_int, c_int, c_void]
    dtfa_bn_bwd_finalize = [c_void, c_int, c_int, c_float, c_void, c_void, c_void, c_void, c_void, c_void, c_void,
                            c_void, c_int, c_void]
    dtfa_bn_bwd_apply = [c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_ll, c_int, c_int,
                         c_void]
    dtfa_fold_fwd = [c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_int, c_int,
                     c_float, c_float, c_float, c_void]
    dtfa_fold_bwd = [c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_float, c_void, c_void, c_void,
                     c_void, c_void, c_void, c_int, c_int, c_int, c_int, c_int, c_int, c_void]
    dtfa_relu_bwd = [c_void, c_void, c_void, c_ll, c_void]
    dtfa_add_bf16 = [c_void, c_void, c_void, c_ll, c_void]
    dtfa_bn_relu_maxpool_fwd = [c_void, c_void, c_void, c_void, c_void, c_void, C.POINTER(PoolArgs), c_int, c_void]
    dtfa_bn_pool_partials_rows = [C.POINTER(PoolArgs)]
    dtfa_bn_pool_bwd_partials = [c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_void,
                                 C.POINTER(PoolArgs), c_int, c_void]
    dtfa_bn_pool_bwd_apply = [c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_void,
                              C.POINTER(PoolArgs), c_int, c_void]
    dtfa_maxpool_fwd = [c_void, c_void, c_void, C.POINTER(PoolArgs), c_void]
    dtfa_maxpool_bwd = [c_void, c_void, c_void, C.POINTER(PoolArgs), c_void]
    dtfa_avgpool_fwd = [c_void, c_void, C.POINTER(PoolArgs), c_void]
    dtfa_avgpool_bwd = [c_void, c_void, C.POINTER(PoolArgs), c_void]
    dtfa_gap_fwd = [c_void, c_void, c_int, c_int, c_int, c_void]
    dtfa_gap_bwd = [c_void, c_void, c_int, c_int, c_int, c_void]
    dtfa_softmax_xent = [c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_int, c_int, c_int, c_float,
                         c_float, c_void]
    dtfa_mean_f32 = [c_void, c_void, c_int, c_float, c_void]
    dtfa_act_l2_blocks = [c_ll]
    dtfa_act_l2_fwd = [c_void, c_ll, c_float, c_void, c_void, c_void]
    dtfa_act_l2_bwd = [c_void, c_void, c_void, c_float, c_void, c_ll, c_void]
    dtfa_adam = [c_void, c_void, c_void, c_void, c_int, c_void, c_ll, c_void, c_float, c_float, c_float, c_void]
    dtfa_momentum = [c_void, c_void, c_void, c_int, c_void, c_ll, c_void, c_float, c_float, c_int, c_void]
    dtfa_cast_f32_bf16 = [c_void, c_void, c_ll, c_void]
    dtfa_mask_mul = [c_void, c_void, c_void, c_ll, c_void]
    dtfa_set_hyper = [c_void, c_float, c_float, c_void]
    dtfa_weight_permute = [c_void, c_void, c_void, c_int, c_int, c_int, c_int, c_void]
    dtfa_dropout = [c_void, c_void, c_ll, C.c_ulonglong, c_void, c_float, c_void]
    dtfa_subsample = [c_void, c_void, c_int, c_int, c_int, c_int, c_int, c_int, c_void]
    dtfa_p2p_create = [c_int, c_int, c_ll, C.c_double, C.POINTER(c_void)]
    dtfa_p2p_handle = [c_void, C.c_char_p]
    dtfa_p2p_connect = [c_void, C.c_char_p]
    dtfa_p2p_allreduce = [c_void, c_void, c_void, c_ll, c_int, c_float, c_void]
    dtfa_p2p_allreduce2 = [c_void, c_void, c_void, c_ll, c_int, c_float, c_void]
    dtfa_p2p_status = [c_void]


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError("HIP kernel launch failed (%s): code %d" % (what, rc))


def require(t: torch.Tensor, dtype, what: str) -> None:
    if not t.is_cuda:
        raise ValueError("%s must be a GPU tensor" % what)
    if t.dtype != dtype:
        raise ValueError("%s must be %s, got %s" % (what, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % what)


_zero_pages = {}


def zero_page(device) -> torch.Tensor:
    key = torch.device(device).index
    z = _zero_pages.get(key)
    if z is None:
        z = torch.zeros(256, dtype=torch.bfloat16, device=device)
        _zero_pages[key] = z
    return z


_workspaces = {}


def workspace(device, nbytes: int) -> torch.Tensor:
    """Grow-only scratch buffer (fp32 elements), one per device and stream: launches on different streams may
    overlap, launches on one stream use theirs one after the other."""
    key = (torch.device(device).index, torch.cuda.current_stream(device).cuda_stream)
    w = _workspaces.get(key)
    need = (nbytes + 3) // 4
    if w is None or w.numel() < need:
        w = torch.empty(max(need, 1 << 22), dtype=torch.float32, device=device)
        _workspaces[key] = w
    return w
