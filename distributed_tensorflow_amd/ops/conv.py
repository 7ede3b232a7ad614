"""Convolution on the gfx950 implicit-GEMM kernels.

A :class:`ConvPlan` captures the geometry of one convolution layer at one input
size and owns the small device tables the kernels need (tap offsets, the tap
subsets of every data-gradient phase).  The three entry points map onto the two
kernels in ``csrc/hip``:

* :func:`conv_fwd`    -- ``conv_igemm`` with the forward tap table
* :func:`conv_dgrad`  -- ``conv_igemm`` run on ``dY`` with permuted weights, one
  launch per output-stride phase so no multiply is wasted on the zeros a
  strided convolution's transpose would insert
* :func:`conv_wgrad`  -- ``conv_wgrad`` (split over pixels) + slab reduction

TensorFlow padding rules ("SAME" puts the odd pixel at the bottom/right) are
honoured through explicit ``(top, bottom, left, right)`` pads.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from . import hipk, tune


def same_pads(size: int, k: int, s: int) -> Tuple[int, int]:
    """TensorFlow SAME padding for one axis -> (before, after)."""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return total // 2, total - total // 2


def out_size(size: int, k: int, s: int, p0: int, p1: int) -> int:
    return (size + p0 + p1 - k) // s + 1


def resolve_pads(padding, ih: int, iw: int, kh: int, kw: int, sy: int, sx: int) -> Tuple[int, int, int, int]:
    """padding: 'SAME' | 'VALID' | int | (ph, pw) | (top, bottom, left, right)."""
    if isinstance(padding, str):
        p = padding.upper()
        if p == "VALID":
            return (0, 0, 0, 0)
        if p == "SAME":
            t, b = same_pads(ih, kh, sy)
            l, r = same_pads(iw, kw, sx)
            return (t, b, l, r)
        raise ValueError("unknown padding %r" % padding)
    if isinstance(padding, int):
        return (padding, padding, padding, padding)
    if len(padding) == 2:
        return (padding[0], padding[0], padding[1], padding[1])
    return tuple(padding)  # type: ignore[return-value]


def _pack_taps(taps: List[Tuple[int, int]]) -> List[int]:
    out = []
    for dy, dx in taps:
        if not (-32768 <= dy < 32768 and -32768 <= dx < 32768):
            raise ValueError("tap offset out of int16 range")
        v = ((dy & 0xFFFF) << 16) | (dx & 0xFFFF)
        if v >= 1 << 31:
            v -= 1 << 32
        out.append(v)
    return out


def _magic(d: int) -> int:
    return ((1 << 32) + d - 1) // d if d > 1 else 0xFFFFFFFF


@dataclass
class _Phase:
    taps: torch.Tensor      # packed (dy, dx) offsets into dY
    src_tap: torch.Tensor   # forward tap index of every phase tap
    T: int
    OYn: int
    OXn: int
    oy0: int
    ox0: int


class ConvPlan:
    """Geometry + device tables of one convolution at one spatial size."""

    def __init__(self, ih, iw, cin, cout, kh, kw, sy, sx, pads, device):
        if cin % 8 or cout % 8:
            raise ValueError("conv kernels need Cin %% 8 == 0 and Cout %% 8 == 0 (got %d, %d)" % (cin, cout))
        if kh * kw > 64:
            raise ValueError("at most 64 taps supported")
        if kh * kw * cin >= 1 << 16:
            raise ValueError("K = kh*kw*Cin must be < 65536")
        self.ih, self.iw, self.cin, self.cout = ih, iw, cin, cout
        self.kh, self.kw, self.sy, self.sx = kh, kw, sy, sx
        self.pads = tuple(pads)
        pt, pb, pl, pr = self.pads
        self.oh = out_size(ih, kh, sy, pt, pb)
        self.ow = out_size(iw, kw, sx, pl, pr)
        if self.oh <= 0 or self.ow <= 0:
            raise ValueError("empty convolution output")
        self.T = kh * kw
        self.K = self.T * cin
        self.device = torch.device(device)
        fwd = [(ky - pt, kx - pl) for ky in range(kh) for kx in range(kw)]
        self.fwd_taps = torch.tensor(_pack_taps(fwd), dtype=torch.int32, device=device)
        # data-gradient phases
        self.phases: List[_Phase] = []
        self.dgrad_needs_zero = False
        for ph in range(sy):
            for pw in range(sx):
                oyn = (ih - ph + sy - 1) // sy if ih > ph else 0
                oxn = (iw - pw + sx - 1) // sx if iw > pw else 0
                if oyn <= 0 or oxn <= 0:
                    continue
                taps, src = [], []
                for ky in range(kh):
                    if (ph + pt - ky) % sy:
                        continue
                    for kx in range(kw):
                        if (pw + pl - kx) % sx:
                            continue
                        taps.append(((ph + pt - ky) // sy, (pw + pl - kx) // sx))
                        src.append(ky * kw + kx)
                if not taps:
                    self.dgrad_needs_zero = True
                    continue
                self.phases.append(_Phase(
                    torch.tensor(_pack_taps(taps), dtype=torch.int32, device=device),
                    torch.tensor(src, dtype=torch.int32, device=device),
                    len(taps), oyn, oxn, ph, pw))
        self.cin_magic = _magic(cin)
        self.cout_magic = _magic(cout)
        self.is_pointwise = kh == 1 and kw == 1 and sy == 1 and sx == 1 and not any(self.pads)
        # same-size 3x3: every tap of the forward and of the data-gradient launch lies within one pixel
        self.near_taps = kh == 3 and kw == 3 and sy == 1 and sx == 1 and self.pads == (1, 1, 1, 1)


_plans: Dict[tuple, ConvPlan] = {}


def get_plan(ih, iw, cin, cout, kh, kw, sy, sx, pads, device) -> ConvPlan:
    key = (ih, iw, cin, cout, kh, kw, sy, sx, tuple(pads), str(device))
    p = _plans.get(key)
    if p is None:
        p = ConvPlan(ih, iw, cin, cout, kh, kw, sy, sx, pads, device)
        _plans[key] = p
    return p


# --------------------------------------------------------------------------
# tile selection
# --------------------------------------------------------------------------
_FWD_TILES = {(128, 192), (128, 128), (128, 64), (64, 64)}


# 256 x 256 macro tile for wide, deep layers.
_BIG = os.environ.get("DTFA_CONV_BIG", "1") == "1"
_BIG_MIN_BLOCKS = int(os.environ.get("DTFA_CONV_BIG_MIN_BLOCKS", "192"))
_BIG_MIN_K = int(os.environ.get("DTFA_CONV_BIG_MIN_K", "768"))
# 256 x 128 macro tile (512-thread workgroups) for compute-bound layers.
# (the 256 x 128 macro tile is a candidate of the measured table only: timed inside training steps at batch 256 .. 2048
# it was the fastest choice in 1 of the 97 launches the rule below used to give it, 20 % behind on average)
_WIDE = os.environ.get("DTFA_CONV_WIDE", "0") == "1"
_WIDE_MIN_K = int(os.environ.get("DTFA_CONV_WIDE_MIN_K", "256"))
_WIDE_MIN_TILES = int(os.environ.get("DTFA_CONV_WIDE_MIN_TILES", "1024"))
_WIDE_BK = int(os.environ.get("DTFA_CONV_WIDE_BK", "32"))


def _env_tile(name: str):
    v = os.environ.get(name)
    if not v:
        return None
    return tuple(int(t) for t in v.split(","))


_SMALL_GRID_TILES = int(os.environ.get("DTFA_SMALL_GRID_TILES", "256"))


def pick_fwd_tile(M: int, cout: int, K: int, dense: bool = True) -> Tuple[int, int, int, int]:
    """(BM, BN, BK, LDS stages) of the implicit-GEMM kernel.  Chosen from a sweep over the ResNet-50 layer shapes
    and Inception-v3 layer shapes on MI355X (tools/sweep_conv.py, profiles/tile_sweep_b256.md); a 256-row
    tile never won."""
    forced = _env_tile("DTFA_CONV_TILE")
    if forced:
        return tuple(forced) if len(forced) == 4 else tuple(forced) + (2,)  # type: ignore[return-value]
    # 256 x 256 macro tile (512 threads, each wave 64 x 128, two 64 KB stages, one workgroup per CU): the fewest
    # L2 -> LDS bytes per FLOP, which is what bounds the wide deep layers (14x14 / 7x7 stages: +5 .. 25 % at batch
    # 1024).  Measured inside training steps (profiles/tuned_launches.md) it wins for every output wider than 128
    # channels -- 160 .. 768, multiples of 256 or not, strided data-gradient phases included -- once the reduction
    # is deep and there is about a workgroup per CU; layers with K <= 512 and the heavy residual / batch-norm-tap
    # epilogues are faster on 128 x 128 x 32.
    if _BIG and cout > 128 and K >= _BIG_MIN_K and -(-M // 256) * -(-cout // 256) >= _BIG_MIN_BLOCKS:
        return 256, 256, 64, 2
    # k-depth: deep reductions amortise the per-tile barrier with BK = 64 (two 32 KB stages, two workgroups per
    # CU); below K = 1024 three 16 KB stages (two k-tiles of loads in flight, three workgroups per CU) win.
    bk = 64 if K >= 1024 else 32
    # width: channel counts that are multiples of 192 (Inception: 192, 384) get the 192-wide tile, which
    # wastes no columns; otherwise 128, or 64 for narrow outputs
    bn = 192 if cout % 192 == 0 else (128 if cout > 64 else 64)
    if bn == 192:
        bk = 32   # two 80 KB stages of a 192-wide BK = 64 tile leave room for ONE workgroup per CU: never the fastest
    bm = 128
    tiles = -(-M // bm) * -(-cout // bn)
    if bn == 64 and tiles < 512:
        bm = 64
    if tiles < _SMALL_GRID_TILES and (bm, bn) != (64, 64):
        # fewer tiles than CUs (the 17x17 / 8x8 Inception layers at batch 32, dense layers): the launch is a chain of
        # k-steps on a mostly idle chip, so the smallest tile -- the most workgroups, the shortest k-step -- wins
        # (measured at batch 32: 17x17 1x7 192->192 23.2 -> 15.2 us, 8x8 3x3 448->384 55.7 -> 32.3 us)
        return 64, 64, (64 if K >= 256 else 32), 2
    if _WIDE and bn == 128 and K >= _WIDE_MIN_K and tiles >= _WIDE_MIN_TILES:
        # 256-row macro tile: two 128-row wave groups share one weight tile (fewer L2 -> LDS bytes per FLOP)
        return 256, 128, _WIDE_BK, (2 if _WIDE_BK == 64 else 3)
    return bm, bn, bk, (2 if bk == 64 else 3)


# Tail re-cut.  The macro tiles run one (256 x 256) or two (256 x 128) workgroups per CU, so a launch is a sequence of
# rounds of 256 / 512 tiles, and a last round that fills a fraction of the chip still lasts a whole round: at batch
# 2048 the 7x7 512 -> 512 layers are 784 tiles = 3.06 rounds (measured: 505 us against 429 us for the 768 tiles of
# batch 2006; with the tail re-cut 444 us), the 14x14 256 -> 256 layers 6.125 (504 -> 472 us).  When the last round would be at most ``_TAIL_MAX`` full, the rows of
# that round are given to a second launch with the tile ``pick_fwd_tile`` chooses for a layer of just those rows
# (small tiles, several workgroups per CU: a fraction of a round).
_TAIL = os.environ.get("DTFA_CONV_TAIL", "1") == "1"
_TAIL_MAX = float(os.environ.get("DTFA_CONV_TAIL_MAX", "0.6"))
_TAIL_MAX_ROUNDS = int(os.environ.get("DTFA_CONV_TAIL_MAX_ROUNDS", "8"))


def split_tail(M: int, cout: int, K: int, bm: int, bn: int, dense: bool):
    """(rows of the main launch, tile of the tail launch or None)."""
    if not _TAIL or bm != 256 or _env_tile("DTFA_CONV_TILE"):
        return M, None
    slots = 256 if bn == 256 else 512
    nt = -(-cout // bn)
    tiles = -(-M // bm) * nt
    full = tiles // slots
    rem = tiles - full * slots
    if full < 1 or full > _TAIL_MAX_ROUNDS or rem == 0 or rem > _TAIL_MAX * slots:
        return M, None   # (beyond a handful of rounds the fraction saved is within the noise: measured at 12 and 24)
    m_main = (full * slots // nt) * bm
    if m_main <= 0 or m_main >= M:
        return M, None
    tbm, tbn, tbk, tnst = pick_fwd_tile(M - m_main, cout, K, dense)
    if m_main % (128 if tbm == 256 else tbm):
        return M, None
    return m_main, (tbm, tbn, tbk, tnst)


_WGRAD_WIDE = os.environ.get("DTFA_WGRAD_WIDE", "1") == "1"
_WGRAD_ROUNDS = os.environ.get("DTFA_WGRAD_ROUNDS", "1") == "1"
# weight-gradient instantiation -> (workgroups resident per CU, rounds of them a layer with taps is split into)
_WGRAD_RESIDENT = {(256, 256, 32): (1, 4), (256, 256, 64): (1, 4), (256, 128, 32): (2, 4), (128, 256, 32): (2, 4), (64, 256, 32): (2, 4), (128, 128, 32): (3, 3),
                   (64, 128, 32): (4, 3), (128, 64, 32): (4, 3), (64, 64, 32): (6, 4), (128, 128, 64): (2, 4),
                   (64, 128, 64): (3, 3), (128, 64, 64): (3, 3), (64, 64, 64): (5, 4)}


def pick_wgrad_tile(M: int, cout: int, K: int, cin: int = 0) -> Tuple[int, int, int, int]:
    forced = _env_tile("DTFA_WGRAD_TILE")
    bmw = 128 if cout > 64 else 64
    bnw = 128 if (K > 64 and cout > 64) else 64  # narrow outputs: 64x64 tiles won every such shape in the sweep
    if 0 < cin <= 16 and K >= 256:
        bnw = 128  # image-like inputs (the 7x7 stem): many taps of a few channels each, -15 % with the wide K tile
    # a small output over very many pixels with a 128-wide X tile (the 56x56 256->64 / 256->128 1x1 layers at
    # large batch) is bound by streaming dY and X: a 64-pixel k-step halves the barriers per byte moved
    # (+4 .. 10 % there); everywhere else 32 pixels with three stages measured faster
    bkp = 64 if (128 <= K <= 256 and cout <= 128 and M >= 1 << 20) else 32
    # 256-wide tiles (each wave 128 x 64 or 64 x 128 / 32 x 128: more MFMAs per fragment read) where they waste no
    # columns: measured at batch 2048, -10 % on the 14x14 3x3 layers, -4 .. 8 % on the 1x1 layers with 256 or 1024
    # output channels, -8 % on 56x56 256 -> 64; slower wherever K or Cout does not fill them (3x3 with 64 / 128
    # channels: K = 576 / 1152 is 2.25 / 4.5 tiles)
    if _WGRAD_WIDE and cout % 256 == 0:
        bmw, bnw, bkp = 256, 128, 32
    elif _WGRAD_WIDE and cout <= 64 and cin > 0 and K == cin and K % 256 == 0:
        bmw, bnw, bkp = 64, 256, 32
    if forced:
        bmw, bnw, bkp = forced[:3]
    tiles = -(-cout // bmw) * -(-K // bnw)
    kt_total = -(-M // bkp)
    # Workgroups to aim for.  Every split writes its tile as fp32 partials that a second kernel sums, so splits cost
    # traffic: a 1x1 layer is bound by streaming its two operands once and measured fastest with two workgroups per
    # CU (one when a single large tile covers the whole gradient: -5 .. 12 % against four per CU); layers with taps
    # re-read their operands from cache, run at a lower rate per workgroup, and want eight per CU.
    if K != cin or cin <= 0:
        # whole rounds of resident workgroups: a last round that fills a fraction of the slots runs as long as a full
        # one.  Slots per CU follow from the registers and LDS of each instantiation; rounds measured at batch 2048 /
        # 1024 over the ResNet-50 and flower CNN layers (tools/experiments/bench_wgrad_rounds.py): four for the 64 x 64 tile
        # (56x56 3x3 64->64: -25 % against 1.3 rounds, 112x112 32->64: -21 % against 3.3), three for 128 x 128
        # (-5 %), and for the 256-wide tiles one round per nine output tiles up to four (-5 .. 13 %)
        per_cu, rounds = _WGRAD_RESIDENT.get((bmw, bnw, bkp), (2, 4))
        if bmw == 256 or bnw == 256:
            rounds = max(1, min(4, (tiles + 4) // 9))
        want = 256 * per_cu * rounds if cin > 16 else 2048   # (image-like inputs: the stems keep their measured count)
        splits = max(1, min(want // tiles, max(1, kt_total // 8), 4096))
        if splits >= 16:
            splits -= splits % 8   # the same number of splits for each of the 8 XCDs (see _wgrad_candidates)
        if not _WGRAD_ROUNDS:   # the rule this replaced: 2048 workgroups whatever the tile
            splits = max(1, min(-(-2048 // tiles), max(1, kt_total // 8), 1024))
    else:
        want = 256 if (tiles <= 2 and bmw * bnw >= 128 * 128) else 512
        splits = max(1, min(-(-want // tiles), max(1, kt_total // 8), 1024))
    if forced and len(forced) > 3:
        splits = forced[3]
    return bmw, bnw, bkp, splits


# --------------------------------------------------------------------------
# launches
# --------------------------------------------------------------------------
# Outputs at least this large (the 256 MB memory-side cache) are stored non-temporally: their consumer finds nothing of
# them in a cache anyway, and they stop evicting the operands being re-read (ResNet-50 at batch 2048: -0.7 % step time)
STREAM_OUT_BYTES = int(os.environ.get("DTFA_CONV_STREAM_OUT_BYTES", str(256 << 20)))

def _launch_igemm(x, w2d, y, taps, T, ih, iw, cin, oh, ow, cout, oyn, oxn, osy, osx, oy0, ox0, isy, isx, B,
                  cin_magic, bias=None, relu=False, stats=False, addend=None, zero_taps=False, addend_mask=None,
                  bn_tap=None, oscale=None, ldo=0, omask=None, dbias=None, x2=None, addend_shift=(0, 0),
                  stat_acc=None, near_taps=False, grid_taps=None):
    M = B * oyn * oxn
    K = T * cin + (x2.shape[-1] if x2 is not None else 0)
    dense_out = osy == 1 and osx == 1 and oy0 == 0 and ox0 == 0 and oyn == oh and oxn == ow
    L = hipk.lib()
    a = hipk.ConvArgs()
    a.x, a.w, a.y = x.data_ptr(), w2d.data_ptr(), y.data_ptr()
    a.bias = hipk.ptr(bias)
    a.addend = hipk.ptr(addend)
    a.addend_mask = hipk.ptr(addend_mask)
    if bn_tap is not None:
        a.bn_y, a.bn_mean, a.bn_invstd = hipk.ptr(bn_tap.y), hipk.ptr(bn_tap.mean), hipk.ptr(bn_tap.invstd)
        a.bn_scale, a.bn_shift, a.bn_mask = hipk.ptr(bn_tap.scale), hipk.ptr(bn_tap.shift), hipk.ptr(bn_tap.mask)
        a.bn_relu = bn_tap.relu_mode
        a.bn_gate_store = 1 if bn_tap.gate_store else 0
    a.omask, a.dbias, a.x2 = hipk.ptr(omask), hipk.ptr(dbias), hipk.ptr(x2)
    a.stream_out = 1 if M * cout * 2 >= STREAM_OUT_BYTES else 0
    a.K1 = T * cin
    a.addend_shy, a.addend_shx = addend_shift
    a.stat_acc = hipk.ptr(stat_acc)
    a.stat_rep = stat_acc.numel() // (2 * cout) if stat_acc is not None else 0
    a.zero = hipk.zero_page(x.device).data_ptr()
    a.taps = taps.data_ptr()
    a.IH, a.IW, a.Cin = ih, iw, cin
    a.OH, a.OW, a.Cout = oh, ow, cout
    a.OYn, a.OXn = oyn, oxn
    a.osy, a.osx, a.oy0, a.ox0 = osy, osx, oy0, ox0
    a.isy, a.isx = isy, isx
    a.T, a.M, a.K = T, M, K
    a.cin_magic = cin_magic
    a.relu = 1 if relu else 0
    a.oscale = hipk.ptr(oscale)
    a.ldo = ldo or cout
    a.dense_out = 1 if dense_out else 0
    # 1x1 / stride 1 / no padding: input pixel index == output pixel index, the kernel skips the pixel decode
    a.pointwise = 1 if (zero_taps and a.dense_out and T == 1 and isy == 1 and isx == 1
                        and ih == oyn and iw == oxn) else 0
    want_part = stats or bn_tap is not None

    def launch(cfg):
        """One configuration ``(bm, bn, bk, stages, tail[, k-split])``: the main launch and, with ``tail``, the second
        launch that re-cuts a partly filled last round (``split_tail``).  Returns the partial-statistics rows."""
        bm, bn, bk, nst, use_tail = cfg[:5]
        ksp = cfg[5] if len(cfg) > 5 else 1
        m_main, tail = (split_tail(M, cout, K, bm, bn, dense_out) if (use_tail and stat_acc is None) else (M, None))
        part = None
        if want_part:
            rows = L.dtfa_conv_stats_rows(m_main, bm) + (L.dtfa_conv_stats_rows(M - m_main, tail[0]) if tail else 0)
            part = torch.empty((rows, 2, cout), dtype=torch.float32, device=x.device)
        a.stats = hipk.ptr(part)
        a.tile_m_base = a.stat_row_base = 0
        a.tile_m_count = m_main // (128 if bm == 256 else bm) if tail is not None else 0
        if ksp != 1:
            hipk.check(L.dtfa_conv_igemm_ks(C.byref(a), bm, bn, bk, nst, ksp, hipk.stream()),
                       "conv_igemm %dx%dx%d/%d k-split %d" % (bm, bn, bk, nst, ksp))
        else:
            hipk.check(L.dtfa_conv_igemm(C.byref(a), bm, bn, bk, nst, hipk.stream()),
                       "conv_igemm %dx%dx%d/%d" % (bm, bn, bk, nst))
        if tail is not None:
            tbm, tbn, tbk, tnst = tail
            a.tile_m_count = 0
            a.tile_m_base = m_main // (128 if tbm == 256 else tbm)
            a.stat_row_base = L.dtfa_conv_stats_rows(m_main, bm)
            hipk.check(L.dtfa_conv_igemm(C.byref(a), tbm, tbn, tbk, tnst, hipk.stream()),
                       "conv_igemm tail %dx%dx%d/%d" % (tbm, tbn, tbk, tnst))
        return part

    rule = pick_fwd_tile(M, cout, K, dense_out) + (1,)
    rule += (pick_ksplit(rule, M, cout, K),)
    cfg = rule
    if tune.MODE != "off" and stat_acc is None and not _env_tile("DTFA_CONV_TILE"):
        # epilogue class as the kernel dispatches it: its cost differs per tile shape, so it is part of the key
        epi = 3 if bn_tap is not None else 1 if stats else 2 if (bias is not None or relu or oscale is not None
                                                                   or (ldo or cout) != cout) else 0
        key = "%d,%d,%d,%d,%d,%d,%d,%d" % (M, cout, K, T, dense_out, a.pointwise, epi, x2 is not None)
        found = tune.lookup("igemm", key)
        if found is None and tune.measuring() and M * cout >= _TUNE_MIN_OUT:
            found = tune.measure("igemm", key, _igemm_candidates(M, cout, K), launch, rule)
        if found is not None:   # entries measured before the reduction split existed carry no split: the rule adds it
            cfg = found if len(found) > 5 else tuple(found) + (pick_ksplit(found, M, cout, K),)
    if (grid_taps is not None and p8_takes(cin, cout, grid_taps, ih, iw, oyn, oxn, isy, isx, M) and dense_out and x2 is None
            and stat_acc is None and bn_tap is None and not _env_tile("DTFA_CONV_TILE")):
        # conv_p8_kernel: persistent workgroups over 8-channel pixels, the pixels a tile touches staged once in LDS
        part = None
        if want_part:
            part = torch.empty((L.dtfa_conv_stats_rows(M, 128), 2, cout), dtype=torch.float32, device=x.device)
        a.stats = hipk.ptr(part)
        a.tile_m_base = a.stat_row_base = a.tile_m_count = 0
        hipk.check(L.dtfa_conv_p8(C.byref(a), grid_taps[0], grid_taps[1], hipk.stream()), "conv_p8")
        return part
    if (_P3 and near_taps and cin in P3_CINS and cout <= P3_MAX_COUT and x2 is None and dense_out and iw <= P3_MAX_WIDTH
            and stat_acc is None and bn_tap is None and M >= _P3_MIN_ROWS and not _env_tile("DTFA_CONV_TILE")):
        cfg = (128, 64, 64, 3, 0, -1)   # conv_p3_kernel: persistent workgroups, weights in registers, pixel patches in LDS
    return launch(cfg)


# Reduction split inside the workgroup (conv_igemm.hip, KSP): for launches of few tiles -- the small-launch path's
# 64 x 64 x 64 tile -- with a deep reduction, 2 or 4 wave groups each take a slice of every k-tile.
# conv_p3_kernel (csrc/hip/conv_p3.hip: same-size 3x3 over 64 channels by two persistent workgroups per CU, the
# weights resident in registers, the pixels a tile touches over all nine taps staged once in LDS): same reduction order
# as the ordinary kernel, so the same bits.  Two pixel patches of 128 + 2 (width + 1) rows fit in half a CU's LDS up to 87
# pixels of width; wider images (to 240) get one patch buffer per workgroup, still two workgroups per CU (112x112
# 64 -> 32, the flower CNN's second data gradient at batch 1024: 1521 -> 1093 us; with two buffers and ONE workgroup per
# CU 1776).
# Not for the data gradient with batch-norm sums (its epilogue needs the registers the resident weights hold).
_P3 = os.environ.get("DTFA_CONV_P3", "1") == "1"
_P3_MIN_ROWS = int(os.environ.get("DTFA_CONV_P3_MIN_ROWS", str(1 << 16)))
P3_MAX_WIDTH = int(os.environ.get("DTFA_CONV_P3_MAX_WIDTH", "240"))
P3_MAX_COUT = int(os.environ.get("DTFA_CONV_P3_MAX_COUT", "128"))   # (slices of 64 columns, one workgroup each)
P3_CINS = tuple(int(c) for c in os.environ.get("DTFA_CONV_P3_CINS", "32,64").split(",") if c)   # input channels it is built for
_P3_DGRAD = os.environ.get("DTFA_CONV_P3_DGRAD", "1") == "1"
# conv_p8_kernel (csrc/hip/conv_p8.hip): strided layers over 8-channel pixels whose input is exactly stride times their
# output (the pixel-pair stem of ResNet-50, 7 x 4 taps).  Not for 32 output channels, where half of its waves idle: the
# flower CNN's first 3x3 / stride-2 layer measured 523 against 490 us.
_P8 = os.environ.get("DTFA_CONV_P8", "1") == "1"
_P8_TAPS = (9, 10, 27, 28)   # tap counts the kernel is instantiated for (5 or 14 k-steps of two taps)


def p8_takes(cin: int, cout: int, grid, ih: int, iw: int, oh: int, ow: int, sy: int, sx: int, M: int) -> bool:
    kh, kw = grid
    return (_P8 and cin == 8 and 32 < cout <= 64 and kh * kw in _P8_TAPS and ih == oh * sy and iw == ow * sx and iw > kw
            and M >= _P3_MIN_ROWS and hipk.lib().dtfa_conv_p8_buffer(iw, ow, sy, sx, kh, kw) > 0)


_KSPLIT = os.environ.get("DTFA_CONV_KSPLIT", "auto")
_KSPLIT_MAX_TILES = int(os.environ.get("DTFA_CONV_KSPLIT_MAX_TILES", "512"))


def pick_ksplit(tile, M: int, cout: int, K: int) -> int:
    if tuple(tile[:4]) != (64, 64, 64, 2) or _KSPLIT == "0":
        return 1
    if _KSPLIT in ("2", "4"):
        return int(_KSPLIT)
    tiles = -(-M // 64) * -(-cout // 64)
    if tiles > _KSPLIT_MAX_TILES:
        return 1
    # graph-timed at batch 32 (tools/experiments/sweep_ksplit.py): up to a workgroup per CU, 8x8 2048->384 17.6 -> 12.5 (two
    # groups) -> 11.7 us (four), 8x8 3x3 448->384 (K = 4032) 34.0 -> 22.7 -> 19.8; with 435 tiles (the 17x17 layers)
    # two groups gain 10 % and four, at one 1024-thread workgroup per CU, lose 30 %
    if tiles <= 256:
        return 4 if K >= 1536 else (2 if K >= 512 else 1)
    return 2 if K >= 768 else 1


# configurations an autotuning run tries (tools/tune_conv.py); launches with less output than this keep the rules
_TUNE_MIN_OUT = int(os.environ.get("DTFA_TUNE_MIN_OUT", str(1 << 22)))


def _igemm_candidates(M: int, cout: int, K: int):
    out = []
    for bm, bn in ((128, 192), (128, 128), (128, 64), (64, 64)):
        if bn >= 128 and cout <= bn // 2:
            continue   # at least half of the tile's columns would be padding
        out += [(bm, bn, 32, 3, 0), (bm, bn, 64, 2, 0), (bm, bn, 32, 2, 0)]
    if cout > 64:
        out += [(256, 128, 32, 3, 0), (256, 128, 32, 3, 1), (256, 128, 64, 2, 0)]
    if cout > 128 and K >= 128:
        out += [(256, 256, 64, 2, 0), (256, 256, 64, 2, 1)]
    if K >= 256 and -(-M // 64) * -(-cout // 64) <= 2048:
        out += [(64, 64, 64, 2, 0, 2), (64, 64, 64, 2, 0, 4)]
    return out


def stat_replicas(rows: int) -> int:
    """Replicas of the fp64 statistics accumulators for a layer of ``rows`` output pixels: every 64-row tile of a
    column adds to the same addresses and those atomics are served one after the other (35x35 at batch 32, 613
    tiles: 12 us of a 19 us launch), so beyond ~100 tiles per address the tiles are dealt over 2 .. 64 replicas."""
    tiles = -(-rows // 64)
    rep = 1
    while rep < 64 and tiles > _STAT_TILES_PER_REPLICA * rep:
        rep *= 2
    return rep


_STAT_TILES_PER_REPLICA = int(os.environ.get("DTFA_STAT_TILES_PER_REPLICA", "96"))


def conv_fwd(x: torch.Tensor, w: torch.Tensor, plan: ConvPlan, bias: Optional[torch.Tensor] = None,
             relu: bool = False, want_stats: bool = False, scale: Optional[torch.Tensor] = None,
             out: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
             mask_out: Optional[torch.Tensor] = None, x2: Optional[torch.Tensor] = None,
             stat_acc: Optional[torch.Tensor] = None):
    """x [B,IH,IW,Cin] bf16, w [Cout,KH,KW,Cin] bf16 -> (y [B,OH,OW,Cout] bf16, stats partials or None).

    ``stat_acc`` (fp64 [R * 2 * Cout], R a power of two, zeroed by the caller; excludes every other epilogue option):
    the column sums and sums of squares of the output are *added* to it by atomics -- for launches of few tiles, whose
    consumer (``kernels.bn_apply_sums``) then needs no reduction pass over partial rows.  Row tile t adds to replica
    ``t mod R`` (``stat_replicas`` says how many a layer wants: atomics on one address are served one at a time).

    Pointwise layers: with ``x2`` ([B,OH,OW,C2] bf16) the reduction runs over ``[x | x2]`` and ``w`` is
    ``[Cout, Cin + C2]`` (two 1x1 convolutions summed in one GEMM).

    ``mask_out`` (uint8 [numel / 8], with ``relu`` and a dense output) receives the sign of the activation, one
    bit per element: what a training step keeps of a ReLU output for its backward pass.

    ``scale`` / ``bias`` ([Cout] fp32) turn the epilogue into ``act(conv * scale + bias)`` -- a convolution with
    its inference-mode batch-norm folded in; ``residual`` ([B,OH,OW,Cout] bf16, with ``bias`` or ``scale``) is added
    before the activation (the tail of a residual block).  ``out`` is a channel slice ``buf[..., c0:c0+Cout]`` of a contiguous
    NHWC tensor (``c0`` and the buffer's channel count multiples of 8) that receives the rows directly."""
    hipk.require(x, torch.bfloat16, "x")
    hipk.require(w, torch.bfloat16, "w")
    B = x.shape[0]
    if tuple(x.shape[1:]) != (plan.ih, plan.iw, plan.cin):
        raise ValueError("input shape %s does not match plan (%d,%d,%d)" % (tuple(x.shape), plan.ih, plan.iw, plan.cin))
    c2 = 0
    if x2 is not None:
        hipk.require(x2, torch.bfloat16, "x2")
        c2 = x2.shape[-1]
        if not plan.is_pointwise or tuple(x2.shape[:-1]) != (B, plan.oh, plan.ow) or c2 % 8:
            raise ValueError("a second operand needs a pointwise layer and the output's pixel grid")
    if w.numel() != plan.cout * (plan.K + c2) or w.shape[0] != plan.cout:
        raise ValueError("weight shape %s does not match plan" % (tuple(w.shape),))
    if bias is not None:
        hipk.require(bias, torch.float32, "bias")
        if bias.numel() != plan.cout:
            raise ValueError("bias size")
        if want_stats:
            raise ValueError("batch-norm statistics and bias are mutually exclusive")
    if scale is not None:
        hipk.require(scale, torch.float32, "scale")
        if scale.numel() != plan.cout or want_stats:
            raise ValueError("scale needs Cout entries and excludes batch-norm statistics")
    if residual is not None:
        hipk.require(residual, torch.bfloat16, "residual")
        if tuple(residual.shape) != (B, plan.oh, plan.ow, plan.cout) or want_stats or (bias is None and scale is None):
            raise ValueError("residual needs the output's shape and the bias / scale epilogue")
    if B * plan.oh * plan.ow >= 1 << 31:
        raise ValueError("too many output pixels for 32-bit row indices")
    if stat_acc is not None:
        rep = stat_acc.numel() // (2 * plan.cout)
        if (stat_acc.dtype != torch.float64 or rep < 1 or stat_acc.numel() != rep * 2 * plan.cout or rep & (rep - 1)
                or rep > 64 or not stat_acc.is_contiguous() or stat_acc.device != x.device):
            raise ValueError("stat_acc must hold 2^k x 2 * Cout contiguous fp64 values on the input's device")
        if (want_stats or bias is not None or scale is not None or relu or residual is not None or out is not None
                or mask_out is not None):
            raise ValueError("stat_acc excludes the other epilogue options")
    ldo = 0
    if out is None:
        y = torch.empty((B, plan.oh, plan.ow, plan.cout), dtype=torch.bfloat16, device=x.device)
    else:
        y = out
        ldo = y.stride(2)
        if (y.dtype != torch.bfloat16 or y.device != x.device or tuple(y.shape) != (B, plan.oh, plan.ow, plan.cout)
                or y.stride(3) != 1 or ldo < plan.cout or ldo % 8 or y.stride(1) != plan.ow * ldo
                or y.stride(0) != plan.oh * plan.ow * ldo or y.data_ptr() % 16 or want_stats):
            raise ValueError("out must be a 16-byte aligned channel slice of a contiguous NHWC bf16 tensor")
    if mask_out is not None:
        if not relu or out is not None or mask_out.dtype != torch.uint8 or not mask_out.is_contiguous() \
                or mask_out.numel() * 8 != y.numel() or mask_out.device != x.device:
            raise ValueError("mask_out needs a ReLU, a dense output and numel / 8 bytes")
    part = _launch_igemm(x, w, y, plan.fwd_taps, plan.T, plan.ih, plan.iw, plan.cin, plan.oh, plan.ow, plan.cout,
                         plan.oh, plan.ow, 1, 1, 0, 0, plan.sy, plan.sx, B, plan.cin_magic, bias, relu, want_stats,
                         addend=residual, zero_taps=plan.is_pointwise, oscale=scale, ldo=ldo, omask=mask_out, x2=x2,
                         stat_acc=stat_acc, near_taps=plan.near_taps, grid_taps=(plan.kh, plan.kw))
    return y, part


# --------------------------------------------------------------------------
# dense layers over a handful of rows (csrc/hip/dense_skinny.hip)
# --------------------------------------------------------------------------
SKINNY_MAX_ROWS = int(os.environ.get("DTFA_SKINNY_MAX_ROWS", "128"))


def dense_skinny_ok(rows: int, n_in: int, n_out: int) -> bool:
    return 0 < rows <= SKINNY_MAX_ROWS and n_in % 16 == 0 and n_out % 8 == 0


def dense_skinny(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, relu: bool = False) -> torch.Tensor:
    """``act(x [M, K] @ w [N, K]^T + bias)`` for few rows: the reduction, not the output, is spread over the waves.
    (The two dense layers of the reference's head at its batch of 32,
    recognition_flower/single/flower_train.py:72-84, and every dense layer when serving single images.)"""
    hipk.require(x, torch.bfloat16, "x")
    hipk.require(w, torch.bfloat16, "w")
    M, Kd = x.shape
    N = w.shape[0]
    if w.dim() != 2 or w.shape[1] != Kd or Kd % 16 or N % 8:
        raise ValueError("dense_skinny: w must be [N, K] with K %% 16 == 0 and N %% 8 == 0, got %s" % (tuple(w.shape),))
    if bias is not None:
        hipk.require(bias, torch.float32, "bias")
        if bias.numel() != N:
            raise ValueError("bias size")
    y = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    a = hipk.SkinnyArgs()
    a.x, a.w, a.bias, a.y = x.data_ptr(), w.data_ptr(), hipk.ptr(bias), y.data_ptr()
    a.M, a.N, a.K, a.relu = M, N, Kd, 1 if relu else 0
    hipk.check(hipk.lib().dtfa_dense_skinny(C.byref(a), hipk.stream()), "dense_skinny")
    return y


# --------------------------------------------------------------------------
class BnBackwardTap:
    """What a data-gradient kernel needs to emit the backward partial sums of the batch-norm layer its
    output flows into: that layer's input ``y`` and statistics, and how its ReLU sign is obtained
    (``relu_mode`` 0: no activation, 2: recomputed from ``y*scale+shift``, 3: stored bits in ``mask``).
    The producer of the gradient stores the partial sums in ``partials``.

    A tap with ``gate_store`` (a layer whose batch-norm is folded through its convolution, see
    ``ops.functional._ExpandBnResFn``) asks for less and more: no ``y`` (only ``sum g`` is emitted), and the
    producer *stores* the gated gradient ``g`` instead of the raw one; it then sets ``gated``."""

    __slots__ = ("y", "mean", "invstd", "scale", "shift", "mask", "relu_mode", "partials", "gate_store", "gated")

    def __init__(self):
        self.y = self.mean = self.invstd = self.scale = self.shift = self.mask = self.partials = None
        self.relu_mode = 0
        self.gate_store = self.gated = False

    @property
    def armed(self) -> bool:
        return self.y is not None or (self.gate_store and self.mask is not None)

    def clear(self) -> None:
        """Drop the tensor references (the autograd graph may outlive the backward pass of a layer)."""
        self.y = self.mean = self.invstd = self.scale = self.shift = self.mask = self.partials = None
        self.gate_store = self.gated = False

    def validate(self, shape, channels: int) -> None:
        n = 1
        for d in shape:
            n *= d
        if self.gate_store:
            if self.y is not None or self.relu_mode != 3:
                raise ValueError("a gate-and-store tap carries sign bits only")
        else:
            if tuple(self.y.shape) != tuple(shape) or self.y.dtype != torch.bfloat16 or not self.y.is_contiguous():
                raise ValueError("batch-norm tap does not match the gradient's shape")
            for t in (self.mean, self.invstd) + ((self.scale, self.shift) if self.relu_mode == 2 else ()):
                if t is None or t.dtype != torch.float32 or t.numel() != channels:
                    raise ValueError("batch-norm tap statistics have the wrong size")
        if self.relu_mode == 3 and (self.mask is None or self.mask.dtype != torch.uint8
                                    or self.mask.numel() * 8 != n):
            raise ValueError("batch-norm tap bitmask has the wrong size")
        if self.relu_mode not in (0, 2, 3):
            raise ValueError("bad relu_mode")


_TAP_PHASES = os.environ.get("DTFA_TAP_PHASES", "1") == "1"


def dgrad_can_tap(plan: ConvPlan) -> bool:
    """Every input pixel is written by exactly one data-gradient launch (one launch at unit stride, one per
    output-stride phase otherwise), whose epilogue therefore sees the complete gradient of its pixels: the launches
    can emit the batch-norm backward partial sums of the layer below (their rows simply add up)."""
    return (len(plan.phases) == 1 or (_TAP_PHASES and len(plan.phases) > 1)) and not plan.dgrad_needs_zero


def p3_takes(plan: ConvPlan, B: int, cin: int, cout: int) -> bool:
    """The persistent 3x3 kernel serves a same-size convolution ``cin -> cout`` of this plan's geometry over ``B``
    images (the forward of the plan, or its data gradient with the channel counts swapped)."""
    return (_P3 and plan.near_taps and cin in P3_CINS and cout <= P3_MAX_COUT and plan.iw <= P3_MAX_WIDTH
            and B * plan.ih * plan.iw >= _P3_MIN_ROWS and not _env_tile("DTFA_CONV_TILE"))


def dgrad_tap_pays(plan: ConvPlan, B: int) -> bool:
    """Whether the data gradient of ``plan`` should emit the batch-norm backward sums of the layer below in its
    epilogue.  Not where the persistent 3x3 kernel can run instead: that kernel without the sums plus a separate
    pass over the gradient (0.55 + 0.33 ms at 56x56x64, batch 2048) beats the ordinary kernel with them (1.18 ms)."""
    return not (_P3_DGRAD and p3_takes(plan, B, plan.cout, plan.cin))


def dgrad_coarse_addend_ok(plan: ConvPlan, stride) -> bool:
    """The data gradient of ``plan`` can add a gradient that lives on the grid ``[::sy, ::sx]`` of its output: one launch
    over a dense output, strides powers of two up to 16."""
    sy, sx = stride
    return (dgrad_can_tap(plan) and len(plan.phases) == 1 and plan.sy == 1 and plan.sx == 1 and 1 <= sy <= 16 and 1 <= sx <= 16
            and sy & (sy - 1) == 0 and sx & (sx - 1) == 0)


def conv_dgrad(dy: torch.Tensor, w: torch.Tensor, plan: ConvPlan, addend: Optional[torch.Tensor] = None,
               addend_mask: Optional[torch.Tensor] = None, bn_tap: Optional[BnBackwardTap] = None,
               wd: Optional[torch.Tensor] = None, x2: Optional[torch.Tensor] = None,
               dbias: Optional[torch.Tensor] = None, addend_stride: Tuple[int, int] = (1, 1)) -> torch.Tensor:
    """dy [B,OH,OW,Cout] -> dx [B,IH,IW,Cin] (+ ``addend``, e.g. the gradient arriving over a
    residual connection, fused into the epilogue instead of a separate add pass; ``addend_mask`` holds
    one bit per element of ``addend`` and only set elements are added).

    ``addend_stride`` (sy, sx), powers of two: ``addend`` is [B, ceil(IH/sy), ceil(IW/sx), Cin] and lands on the pixels
    ``dx[:, ::sy, ::sx]`` -- the gradient of a strided 1x1 shortcut reading the same input, added without ever being
    scattered into a full-resolution tensor (see ``dgrad_coarse_addend_ok``).

    Pointwise layers only: ``wd`` ([Cin, K] bf16) replaces the permuted weight matrix; with ``x2`` ([B,OH,OW,C2]
    bf16) the reduction runs over ``[dy | x2]`` (K = Cout + C2), i.e. ``dx = dy @ wd[:, :Cout].T + x2 @
    wd[:, Cout:].T``; ``dbias`` ([Cin] fp32) is added before rounding."""
    hipk.require(dy, torch.bfloat16, "dy")
    hipk.require(w, torch.bfloat16, "w")
    if wd is not None or x2 is not None or dbias is not None:
        if not plan.is_pointwise:
            raise ValueError("explicit data-gradient matrices are for pointwise layers")
        c2 = 0
        if x2 is not None:
            hipk.require(x2, torch.bfloat16, "x2")
            c2 = x2.shape[-1]
            if tuple(x2.shape[:-1]) != tuple(dy.shape[:-1]) or c2 % 8:
                raise ValueError("second data-gradient operand shape")
        if wd is None or wd.dtype != torch.bfloat16 or not wd.is_contiguous() or tuple(wd.shape) != (plan.cin, plan.cout + c2):
            raise ValueError("explicit data-gradient matrix must be [Cin, Cout + C2] bf16")
        if dbias is not None:
            hipk.require(dbias, torch.float32, "dbias")
            if dbias.numel() != plan.cin:
                raise ValueError("data-gradient bias size")
    B = dy.shape[0]
    if tuple(dy.shape[1:]) != (plan.oh, plan.ow, plan.cout):
        raise ValueError("dy shape %s does not match plan" % (tuple(dy.shape),))
    if w.numel() != plan.cout * plan.K:
        raise ValueError("weight shape does not match plan")
    L = hipk.lib()
    if addend is not None:
        hipk.require(addend, torch.bfloat16, "addend")
        asy, asx = addend_stride
        if (asy, asx) != (1, 1) and (not dgrad_coarse_addend_ok(plan, addend_stride) or addend_mask is not None):
            raise ValueError("a coarse-grid addend needs a single-launch unit-stride data gradient and power-of-two strides")
        if tuple(addend.shape) != (B, -(-plan.ih // asy), -(-plan.iw // asx), plan.cin):
            raise ValueError("addend shape does not match dx")
        if plan.dgrad_needs_zero:
            raise ValueError("addend fusion needs every input pixel to be written by a dgrad phase")
        if addend_mask is not None and (addend_mask.dtype != torch.uint8 or addend_mask.numel() * 8 != addend.numel()):
            raise ValueError("addend bitmask size mismatch")
    elif addend_mask is not None:
        raise ValueError("addend_mask without addend")
    if bn_tap is not None:
        if not dgrad_can_tap(plan):
            raise ValueError("batch-norm partials need a data gradient that writes every pixel")
        bn_tap.validate((B, plan.ih, plan.iw, plan.cin), plan.cin)
    if plan.dgrad_needs_zero:
        dx = torch.zeros((B, plan.ih, plan.iw, plan.cin), dtype=torch.bfloat16, device=dy.device)
    else:
        dx = torch.empty((B, plan.ih, plan.iw, plan.cin), dtype=torch.bfloat16, device=dy.device)
    parts = []
    for ph in plan.phases:
        if wd is None:
            wp = torch.empty((plan.cin, ph.T * plan.cout), dtype=torch.bfloat16, device=dy.device)
            hipk.check(L.dtfa_weight_permute(w.data_ptr(), wp.data_ptr(), ph.src_tap.data_ptr(), plan.cout, plan.T,
                                             plan.cin, ph.T, hipk.stream()), "weight_permute")
        else:
            wp = wd
        part = _launch_igemm(dy, wp, dx, ph.taps, ph.T, plan.oh, plan.ow, plan.cout, plan.ih, plan.iw, plan.cin,
                             ph.OYn, ph.OXn, plan.sy, plan.sx, ph.oy0, ph.ox0, 1, 1, B, plan.cout_magic,
                             addend=addend, zero_taps=plan.is_pointwise, addend_mask=addend_mask, bn_tap=bn_tap,
                             x2=x2, dbias=dbias, near_taps=plan.near_taps,
                             addend_shift=(addend_stride[0].bit_length() - 1, addend_stride[1].bit_length() - 1)
                             if addend is not None else (0, 0))
        parts.append(part)
    if bn_tap is not None:
        bn_tap.partials = parts[0] if len(parts) == 1 else torch.cat(parts, 0)
        bn_tap.gated = bn_tap.gate_store
    return dx


_WGRAD_COLSUM_TILES = {(256, 128, 32), (128, 128, 32), (128, 128, 64), (64, 64, 32), (64, 64, 64)}


def wgrad_colsum_ok(plan: ConvPlan, M: int) -> bool:
    """The weight-gradient launch of ``plan`` over ``M`` pixels can also sum the columns of its X operand."""
    return plan.T == 1 and pick_wgrad_tile(M, plan.cout, plan.K, plan.cin)[:3] in _WGRAD_COLSUM_TILES


def conv_wgrad(x: torch.Tensor, dy: torch.Tensor, plan: ConvPlan, out: Optional[torch.Tensor] = None,
               accumulate: bool = False, colsum: bool = False):
    """Weight gradient [Cout,KH,KW,Cin].  ``out`` may be an fp32 or bf16 buffer
    of the right size (written, or added to when ``accumulate``); by default a
    new fp32 tensor is returned.

    ``colsum`` (1x1 layers, no ``out``): returns ``(dW, s)`` with ``s[ci] = sum over pixels of x[..., ci]`` (fp32),
    computed by the same launch from the X fragments it already holds (see ``wgrad_colsum_ok``)."""
    hipk.require(x, torch.bfloat16, "x")
    hipk.require(dy, torch.bfloat16, "dy")
    B = x.shape[0]
    if tuple(x.shape[1:]) != (plan.ih, plan.iw, plan.cin) or tuple(dy.shape) != (B, plan.oh, plan.ow, plan.cout):
        raise ValueError("wgrad operand shapes do not match plan")
    M = B * plan.oh * plan.ow
    if M >= 1 << 31 or B * plan.ih * plan.iw >= 1 << 31:
        raise ValueError("too many pixels for 32-bit indices")
    n = plan.cout * plan.K
    if colsum:
        if out is not None or not wgrad_colsum_ok(plan, M):
            raise ValueError("column sums ride on 1x1 weight-gradient launches with one of the tiles built for them")
        both = torch.empty(n + plan.K, dtype=torch.float32, device=x.device)
        out = both[:n].view(plan.cout, plan.kh, plan.kw, plan.cin)
        accumulate = False
    if out is None:
        out = torch.empty((plan.cout, plan.kh, plan.kw, plan.cin), dtype=torch.float32, device=x.device)
        accumulate = False
    if out.numel() != n or not out.is_contiguous() or out.dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("bad wgrad output buffer")
    if out.dtype == torch.bfloat16 and accumulate:
        raise ValueError("accumulation needs an fp32 gradient buffer")
    per_split = n + (plan.K if colsum else 0)
    L = hipk.lib()
    a = hipk.WgradArgs()
    a.x, a.dy = x.data_ptr(), dy.data_ptr()
    a.zero = hipk.zero_page(x.device).data_ptr()
    a.taps = plan.fwd_taps.data_ptr()
    a.IH, a.IW, a.Cin = plan.ih, plan.iw, plan.cin
    a.OH, a.OW, a.Cout = plan.oh, plan.ow, plan.cout
    a.isy, a.isx = plan.sy, plan.sx
    a.T, a.M, a.K = plan.T, M, plan.K
    a.cin_magic = plan.cin_magic
    a.colsum = 1 if colsum else 0

    def launch(cfg, dst=None, dst_both=None, acc=None):
        """Slab kernel + reduction of one configuration ``(bmw, bnw, bkp, splits)`` into ``dst`` (default: the real
        output; an autotuning run passes a scratch buffer so that repeated launches do not accumulate)."""
        bmw, bnw, bkp, splits = cfg
        dst = out if dst is None else dst
        dst_both = (both if colsum else None) if dst_both is None else dst_both
        acc = accumulate if acc is None else acc
        # an unsplit reduction (few pixels: dense layers at small batch) writes the fp32 gradient in place
        direct = splits == 1 and dst.dtype == torch.float32 and not acc
        slab = (dst_both if colsum else dst) if direct else hipk.workspace(x.device, splits * per_split * 4)
        a.slab = slab.data_ptr()
        hipk.check(L.dtfa_conv_wgrad(C.byref(a), bmw, bnw, bkp, splits, hipk.stream()), "conv_wgrad")
        if direct:
            return
        if colsum:
            hipk.check(L.dtfa_wgrad_reduce(slab.data_ptr(), splits, per_split, dst_both.data_ptr(), None, 0, hipk.stream()),
                       "wgrad_reduce")
            return
        f32 = dst.data_ptr() if dst.dtype == torch.float32 else None
        b16 = dst.data_ptr() if dst.dtype == torch.bfloat16 else None
        hipk.check(L.dtfa_wgrad_reduce(slab.data_ptr(), splits, n, f32, b16, 1 if acc else 0, hipk.stream()),
                   "wgrad_reduce")

    if wgrad_p3_takes(plan, M) and not colsum:
        # persistent kernel: the whole 64 x 576 gradient in the registers of each resident workgroup, one slab each
        slabs = L.dtfa_conv_wp3_slabs(M)
        slab = hipk.workspace(x.device, slabs * n * 4)
        a.slab = slab.data_ptr()
        hipk.check(L.dtfa_conv_wp3(C.byref(a), hipk.stream()), "conv_wp3")
        f32 = out.data_ptr() if out.dtype == torch.float32 else None
        b16 = out.data_ptr() if out.dtype == torch.bfloat16 else None
        hipk.check(L.dtfa_wgrad_reduce(slab.data_ptr(), slabs, n, f32, b16, 1 if accumulate else 0, hipk.stream()),
                   "wgrad_reduce")
        return out
    if wgrad_p8_takes(plan, M) and not colsum:
        # persistent kernel over 8-channel pixels: the whole 64 x 224 gradient in registers, two slabs per workgroup
        slabs = L.dtfa_conv_wp8_slabs(M)
        slab = hipk.workspace(x.device, slabs * n * 4)
        a.slab = slab.data_ptr()
        hipk.check(L.dtfa_conv_wp8(C.byref(a), plan.kh, plan.kw, hipk.stream()), "conv_wp8")
        f32 = out.data_ptr() if out.dtype == torch.float32 else None
        b16 = out.data_ptr() if out.dtype == torch.bfloat16 else None
        hipk.check(L.dtfa_wgrad_reduce(slab.data_ptr(), slabs, n, f32, b16, 1 if accumulate else 0, hipk.stream()),
                   "wgrad_reduce")
        return out
    rule = pick_wgrad_tile(M, plan.cout, plan.K, plan.cin)
    cfg = rule
    if tune.MODE != "off" and not _env_tile("DTFA_WGRAD_TILE"):
        key = "%d,%d,%d,%d,%d,%d,%d" % (M, plan.cout, plan.K, plan.cin, plan.sy, plan.sx, colsum)
        found = tune.lookup("wgrad", key)
        if found is None and tune.measuring() and M * (plan.cout + plan.K) >= _TUNE_MIN_OUT:
            scratch_both = torch.empty(per_split, dtype=torch.float32, device=x.device)
            scratch = scratch_both[:n]
            found = tune.measure("wgrad", key, _wgrad_candidates(M, plan.cout, plan.K),
                                 lambda c: launch(c, scratch, scratch_both, False), rule)
        if found is not None:
            cfg = found
    launch(cfg)
    if colsum:
        return out, both[n:]
    return out


# conv_wp3_kernel (csrc/hip/conv_wp3.hip): the weight gradient of a same-size 3x3 layer 64 -> 64 by persistent
# workgroups that hold the whole gradient in registers and stage every pixel once per tile.
_WGRAD_P3 = os.environ.get("DTFA_WGRAD_P3", "1") == "1"
_WGRAD_P3_MIN_ROWS = int(os.environ.get("DTFA_WGRAD_P3_MIN_ROWS", str(1 << 16)))


def wgrad_p3_takes(plan: ConvPlan, M: int) -> bool:
    return (_WGRAD_P3 and plan.near_taps and plan.cin == 64 and plan.cout == 64 and 16 <= plan.iw <= 180
            and M >= _WGRAD_P3_MIN_ROWS and not _env_tile("DTFA_WGRAD_TILE"))


# conv_wp8_kernel (csrc/hip/conv_wp8.hip): the weight gradient of the pixel-pair stem (8-channel pixels, 7 x 4 taps, input
# = stride x output) by persistent workgroups that stage the input range of a tile once.
_WGRAD_P8 = os.environ.get("DTFA_WGRAD_P8", "1") == "1"


def wgrad_p8_takes(plan: ConvPlan, M: int) -> bool:
    return (_WGRAD_P8 and plan.cin == 8 and plan.cout == 64 and plan.T == 28 and plan.ih == plan.oh * plan.sy
            and plan.iw == plan.ow * plan.sx and 32 <= plan.ow <= 160 and plan.iw > plan.kw
            and M >= _WGRAD_P3_MIN_ROWS and not _env_tile("DTFA_WGRAD_TILE"))


def _wgrad_candidates(M: int, cout: int, K: int):
    out = []
    for (bmw, bnw, bkp), (per_cu, _) in _WGRAD_RESIDENT.items():
        if (bmw >= 128 and cout <= bmw // 2) or (bnw >= 128 and K <= bnw // 2):
            continue   # at least half of the tile would be padding
        tiles = -(-cout // bmw) * -(-K // bnw)
        kt_total = -(-M // bkp)
        cap = min(max(1, kt_total // 8), 4096)
        for r2 in (1, 2, 4, 6, 8):   # half rounds of resident workgroups
            splits = max(1, min(256 * per_cu * r2 // (2 * tiles), cap))
            out.append((bmw, bnw, bkp, splits))
            # ... and the neighbouring multiples of 8.  The kernel deals the splits over the 8 XCDs round-robin (all
            # tiles of a split on one XCD, whose L2 then serves their common operand rows), so with 28 splits four
            # XCDs work through four splits' tiles and four through three: measured on the 256 x 256 tile of the
            # 14x14 3x3 layers, 28 splits 971 us, 56 splits 585 us.
            for s8 in (splits // 8 * 8, -(-splits // 8) * 8):
                if 8 <= s8 <= cap:
                    out.append((bmw, bnw, bkp, s8))
    return sorted(set(out))


