"""Peer-to-peer mailbox all-reduce for small and medium gradient buckets (``csrc/hip/p2p.hip``).

The reference pushes the 8.4 MB of head gradients to the ps as four ``RecvTensor`` RPCs per
step (recognition_flower/distributed/train_dataset.py:147-165); on an xGMI mesh the
small-message case is latency-bound, and a ring collective spends 2(N-1) hops on it.  Here every
rank exposes a mailbox through ``hipIpc``; an all-reduce is one kernel that publishes the local
data, raises a flag at every peer, waits for the peers' flags and sums all mailboxes in rank
order (bit-identical results on every rank).  That *one-shot* form moves N-1 remote reads per
element, which is the right trade up to a few hundred kilobytes.  Above that (``two_shot_min_bytes``, worlds of
three or more) the *two-shot* form runs: rank r sums only slice r and publishes it, then every rank collects the
N reduced slices -- a direct reduce-scatter and a direct all-gather over all links of the mesh at once,
2 (N-1)/N of the buffer per rank instead of N-1 times it, with the same result bits.  Buckets beyond
``max_bytes`` stay on RCCL.

IPC handles are exchanged through the process group (``all_gather_object``), which works with any
backend; the data path itself never touches ``torch.distributed``.
"""
from __future__ import annotations

import ctypes as C
import torch
import torch.distributed as dist

from ..ops import hipk


class P2PAllReduce:
    def __init__(self, max_bytes: int = 1 << 20, group=None, timeout_s: float = 5.0, device=None,
                 two_shot_min_bytes: int = 256 << 10):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if self.world > 8:
            raise ValueError("the mailbox layout holds at most 8 ranks (one xGMI node)")
        self.max_bytes = (int(max_bytes) + 15) // 16 * 16
        self.two_shot_min_bytes = int(two_shot_min_bytes)
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._ctx = C.c_void_p()
        L = hipk.lib()
        with torch.cuda.device(self.device):
            hipk.check(L.dtfa_p2p_create(self.world, self.rank, self.max_bytes, float(timeout_s), C.byref(self._ctx)),
                       "p2p_create")
            buf = C.create_string_buffer(64)
            hipk.check(L.dtfa_p2p_handle(self._ctx, buf), "p2p_handle")
            handles = [None] * self.world
            dist.all_gather_object(handles, buf.raw, group=group)
            hipk.check(L.dtfa_p2p_connect(self._ctx, b"".join(handles)), "p2p_connect")
        dist.barrier(group=group)  # every mailbox is mapped everywhere before the first flag is written

    def fits(self, t: torch.Tensor) -> bool:
        nbytes = t.numel() * t.element_size()
        return (t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.bfloat16)
                and nbytes <= self.max_bytes and nbytes % 16 == 0 and t.data_ptr() % 16 == 0)

    def all_reduce_(self, t: torch.Tensor, scale: float = 1.0, algo: str = "auto") -> torch.Tensor:
        """``t <- scale * sum over ranks of t`` on the current stream (asynchronous to the host).  ``algo``:
        ``one_shot``, ``two_shot`` or ``auto`` (two-shot from ``two_shot_min_bytes`` in worlds of three or more;
        with two ranks both forms move the same bytes and the one-shot form has one flag round less)."""
        if self._ctx is None:
            raise RuntimeError("closed")
        if not self.fits(t):
            raise ValueError("tensor does not fit the peer-to-peer path (dtype, alignment or size)")
        nbytes = t.numel() * t.element_size()
        if algo == "auto":
            algo = "two_shot" if (self.world >= 3 and nbytes >= self.two_shot_min_bytes) else "one_shot"
        if algo not in ("one_shot", "two_shot"):
            raise ValueError("algo must be auto, one_shot or two_shot")
        fn = hipk.lib().dtfa_p2p_allreduce2 if algo == "two_shot" else hipk.lib().dtfa_p2p_allreduce
        hipk.check(fn(self._ctx, t.data_ptr(), t.data_ptr(), nbytes, 1 if t.dtype == torch.bfloat16 else 0,
                      float(scale), hipk.stream()), "p2p_allreduce (%s)" % algo)
        return t

    def check(self) -> None:
        """Synchronise and raise if any flag wait timed out (a peer never arrived)."""
        if self._ctx is not None and hipk.lib().dtfa_p2p_status(self._ctx) != 0:
            raise RuntimeError("peer-to-peer all-reduce timed out waiting for a peer")

    def close(self) -> None:
        if self._ctx is None:
            return
        torch.cuda.synchronize(self.device)
        dist.barrier(group=self.group)  # no peer may still be reading this rank's mailbox
        lib = hipk.lib()
        lib.dtfa_p2p_destroy.argtypes = [C.c_void_p]
        lib.dtfa_p2p_destroy.restype = None
        lib.dtfa_p2p_destroy(self._ctx)
        self._ctx = None
