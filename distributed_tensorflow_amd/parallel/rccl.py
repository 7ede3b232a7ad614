"""RCCL settings for one xGMI node, and the measured bucket-size table.

An MI355X node is eight GPUs on a full xGMI mesh: seven links per GPU, one to each peer.  A single ring drives one
egress link per GPU, so what a collective reaches depends on how many *channels* (rings / trees running side by
side, each on its own set of links and workgroups) RCCL uses.  RCCL picks a channel count from the topology it
detects; the variables below override it.  Nothing is overridden unless asked for: the defaults of the RCCL build
stay in force, and the launchers only make the choice *visible* (the benchmark record lists every ``NCCL_*`` /
``RCCL_*`` variable that was set) and *tunable*:

``DTFA_RCCL_CHANNELS=n`` (or ``--rccl_channels n``)
    sets ``NCCL_MIN_NCHANNELS = NCCL_MAX_NCHANNELS = n`` before the process group is created.
``DTFA_RCCL_LOG_TOPOLOGY=1``
    sets ``NCCL_DEBUG=INFO`` / ``NCCL_DEBUG_SUBSYS=INIT,GRAPH`` so that RCCL prints the rings and trees it built
    (``tools/rccl_topology.py`` runs one collective that way and summarises the log).

The gradient bucket size is the other half of "choose sizes for point-to-point links, not for a switch"
(one bucket of B bytes is B / N per link and step of a ring, or per peer of a direct exchange).
``tools/tune_buckets.py`` measures whole training steps over a range of bucket sizes on the machine at hand and
writes the winners to ``tuned_buckets.json`` next to this file; :func:`tuned_bucket_mb` looks a configuration up
there and the engine's automatic choice (``bucket_mb=0``) uses the entry when there is one, its rule otherwise.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional

_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_buckets.json")


def configure_rccl(channels: Optional[int] = None) -> Dict[str, str]:
    """Apply the overrides described above to ``os.environ`` (call before ``init_process_group``); returns the
    ``NCCL_*`` / ``RCCL_*`` variables that are set afterwards."""
    n = channels if channels else int(os.environ.get("DTFA_RCCL_CHANNELS", "0") or 0)
    if n > 0:
        os.environ["NCCL_MIN_NCHANNELS"] = str(n)
        os.environ["NCCL_MAX_NCHANNELS"] = str(n)
    if os.environ.get("DTFA_RCCL_LOG_TOPOLOGY") == "1":
        os.environ.setdefault("NCCL_DEBUG", "INFO")
        os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH")
    return rccl_env()


def rccl_env() -> Dict[str, str]:
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith(("NCCL_", "RCCL_"))}


def _load_table() -> list:
    try:
        with open(_TABLE) as f:
            return json.load(f).get("entries", [])
    except (OSError, ValueError):
        return []


def tuned_bucket_mb(world: int, mode: str, gradient_mb: float) -> Optional[float]:
    """The measured best bucket size (MiB) for ``world`` ranks in ``mode`` and a model of about ``gradient_mb`` MiB
    of gradient: the entry of the same world and mode whose gradient size is closest, within a factor of two."""
    best, dist_ = None, None
    for e in _load_table():
        if int(e.get("world", 0)) != int(world) or e.get("mode") != mode:
            continue
        g = float(e.get("gradient_mb", 0.0))
        if g <= 0 or not (0.5 <= gradient_mb / g <= 2.0):
            continue
        d = abs(gradient_mb - g)
        if dist_ is None or d < dist_:
            best, dist_ = float(e["bucket_mb"]), d
    return best


def record_bucket(world: int, mode: str, model: str, gradient_mb: float, bucket_mb: float, rows: list) -> None:
    """Add or replace the table entry of (world, mode, model); ``rows`` are the measurements it was chosen from."""
    entries = [e for e in _load_table()
               if not (int(e.get("world", 0)) == int(world) and e.get("mode") == mode and e.get("model") == model)]
    entries.append({"world": int(world), "mode": mode, "model": model, "gradient_mb": round(float(gradient_mb), 2),
                    "bucket_mb": float(bucket_mb), "measured": rows})
    entries.sort(key=lambda e: (e["world"], e["mode"], e["model"]))
    with open(_TABLE, "w") as f:
        json.dump({"entries": entries}, f, indent=1)
        f.write("\n")
