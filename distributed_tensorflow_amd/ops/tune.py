"""Measured launch configurations of the convolution kernels (tile, LDS stages, tail re-cut, weight-gradient splits).

``pick_fwd_tile`` / ``pick_wgrad_tile`` are rules distilled from sweeps at a few batch sizes; a sweep over the very
shapes a job runs finds configurations 10 .. 30 % faster on a fifth of the layers (batch 2048 ResNet-50: the 14x14
256 -> 1024 layers on the 256 x 256 tile, the 28x28 3x3 layers on the plain 128 x 128 tile; Inception's 17x17
layers on the 192-wide tile with three stages).  This module keeps such measurements as a table:

* ``tuned_gfx950.json`` next to this file ships the winners for the shapes of the models at their benchmark batch
  sizes, measured on an MI355X by ``tools/tune_conv.py``.  A lookup is deterministic -- every rank and every run of
  a job launches the same configuration -- and shapes that are not in the table fall back to the rules.
* ``DTFA_AUTOTUNE=1`` measures every shape that is not in the table the first time it is launched (outside HIP
  graph capture): each candidate runs once to warm up and ``REPS`` times between two events; the fastest is used
  for the rest of the process and can be written out with :func:`save`.  ``DTFA_AUTOTUNE=off`` ignores the table.

All candidates without a reduction split compute the same bits of the convolution output (the reduction order over k
does not depend on the tile); only the grouping of the batch-norm partial sums differs.  A sixth element 2 or 4 splits
every k-tile over that many wave groups of the workgroup, which changes the fp32 summation order.
"""
from __future__ import annotations

import json
import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

MODE = os.environ.get("DTFA_AUTOTUNE", "0")
REPS = int(os.environ.get("DTFA_AUTOTUNE_REPS", "3"))
_PATH = os.environ.get("DTFA_TUNE_DB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_gfx950.json"))

_db: Dict[str, Dict[str, List[int]]] = {"igemm": {}, "wgrad": {}}
_measured: Dict[str, Dict[str, dict]] = {"igemm": {}, "wgrad": {}}   # what this process measured (for ``save``)
_loaded = False


def _load() -> None:
    global _loaded
    _loaded = True
    if MODE == "off" or not os.path.exists(_PATH):
        return
    try:
        with open(_PATH) as f:
            d = json.load(f)
        for kind in ("igemm", "wgrad"):
            for k, v in d.get(kind, {}).items():
                _db[kind][k] = [int(t) for t in (v["best"] if isinstance(v, dict) else v)]
                _index(kind, k, _db[kind][k])
    except (OSError, ValueError, KeyError, TypeError):
        pass   # an unreadable table is the same as none: the rules apply


def measuring() -> bool:
    return MODE == "1"


# keys start with the number of GEMM rows (pixels); the rest identifies the layer.  A shape that is not in the table
# borrows the entry of the same layer at the nearest row count within a factor of NEAR (another batch size of the
# same model), which is closer to a measurement than the rules are.
NEAR = float(os.environ.get("DTFA_TUNE_NEAR", "2.0"))
_by_layer: Dict[str, Dict[str, List[Tuple[int, List[int]]]]] = {"igemm": {}, "wgrad": {}}
_near_cache: Dict[Tuple[str, str], Optional[Tuple[int, ...]]] = {}


def _index(kind: str, key: str, cfg: List[int]) -> None:
    m, _, rest = key.partition(",")
    _by_layer[kind].setdefault(rest, []).append((int(m), cfg))
    _near_cache.clear()


def lookup(kind: str, key: str) -> Optional[Tuple[int, ...]]:
    if not _loaded:
        _load()
    v = _db[kind].get(key)
    if v is not None:
        return tuple(v)
    if measuring() or NEAR <= 1.0:
        return None
    ck = (kind, key)
    if ck not in _near_cache:
        m, _, rest = key.partition(",")
        m = int(m)
        best = None
        for m2, cfg in _by_layer[kind].get(rest, ()):
            ratio = max(m, m2) / max(1, min(m, m2))
            if ratio <= NEAR and (best is None or ratio < best[0]):
                best = (ratio, cfg)
        _near_cache[ck] = tuple(best[1]) if best else None
    return _near_cache[ck]


def measure(kind: str, key: str, candidates: Sequence[Tuple[int, ...]], launch: Callable[[Tuple[int, ...]], None],
            default: Tuple[int, ...]) -> Tuple[int, ...]:
    """Time ``launch(c)`` for every candidate on the current stream; remember and return the fastest.  A candidate the
    kernels reject (a tile that is not built, a shape it cannot take) raises in ``launch`` and is skipped."""
    dev = torch.cuda.current_device()
    if torch.cuda.is_current_stream_capturing():
        return default
    times: Dict[str, float] = {}
    best, best_t = default, float("inf")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    seen = set()
    for c in [tuple(default)] + [tuple(c) for c in candidates]:
        if c in seen:
            continue
        seen.add(c)
        try:
            launch(c)
            e0.record()
            for _ in range(REPS):
                launch(c)
            e1.record()
            e1.synchronize()
        except (RuntimeError, ValueError):
            continue
        t = e0.elapsed_time(e1) / REPS
        times[",".join(map(str, c))] = round(t * 1e3, 1)
        if t < best_t:
            best, best_t = c, t
    _db[kind][key] = list(best)
    _index(kind, key, list(best))
    _measured[kind][key] = {"best": list(best), "us": times, "rule": list(default), "device": dev}
    return best


def save(path: Optional[str] = None, merge: bool = True) -> str:
    """Write the table (what was loaded plus what this process measured; with timings for the latter)."""
    path = path or _PATH
    out: Dict[str, Dict[str, object]] = {"igemm": {}, "wgrad": {}}
    if merge and os.path.exists(path):
        try:
            with open(path) as f:
                old = json.load(f)
            for kind in list(out):
                out[kind].update(old.get(kind, {}))
            for k, v in old.items():   # notes kept beside the tables ("_about")
                out.setdefault(k, v)
        except (OSError, ValueError):
            pass
    for kind in ("igemm", "wgrad"):
        for k, v in _db[kind].items():
            if k not in out[kind]:
                out[kind][k] = {"best": v}
        out[kind].update(_measured[kind])
    tmp = path + ".tmp"
    with open(tmp, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    os.replace(tmp, path)
    return path


def stats() -> dict:
    if not _loaded:
        _load()
    return {"mode": MODE, "table": _PATH, "igemm": len(_db["igemm"]), "wgrad": len(_db["wgrad"]),
            "measured_here": len(_measured["igemm"]) + len(_measured["wgrad"])}
