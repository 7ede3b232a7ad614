"""ctypes loaders for the two in-tree native libraries.

``hip()`` returns ``libdtfa_hip.so`` (gfx950 kernels).  It is only legal to
call once PyTorch has initialised its HIP runtime: the library links against
``libamdhip64.so.7`` and must bind to the copy PyTorch already mapped, so that
streams and device pointers are shared.  There is no fallback: if the library
is missing on a machine with a GPU the call raises.

``host()`` returns ``libdtfa_host.so`` (file formats, JPEG entropy decode,
record loader) and builds it on demand when a compiler is available.
"""
from __future__ import annotations

import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_lock = threading.Lock()
_hip = None
_host = None


class NativeLibraryError(RuntimeError):
    pass


def hip_path() -> str:
    # DTFA_HIP_LIB points experiments (tools/build_variant.py) at an alternative build of the kernels
    return os.environ.get("DTFA_HIP_LIB") or os.path.join(_HERE, "libdtfa_hip.so")


def host_path() -> str:
    return os.path.join(_HERE, "libdtfa_host.so")


def hip() -> ctypes.CDLL:
    global _hip
    if _hip is not None:
        return _hip
    with _lock:
        if _hip is None:
            import torch  # noqa: F401  (maps libamdhip64 first)

            path = hip_path()
            if not os.path.exists(path):
                raise NativeLibraryError(
                    "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(or `python -m distributed_tensorflow_amd._build`) before using GPU ops" % path
                )
            lib = ctypes.CDLL(path)
            lib.dtfa_hip_arch.restype = ctypes.c_char_p
            if lib.dtfa_hip_arch() != b"gfx950":
                raise NativeLibraryError("libdtfa_hip.so was not built for gfx950")
            _hip = lib
    return _hip


def host() -> ctypes.CDLL:
    global _host
    if _host is not None:
        return _host
    with _lock:
        if _host is None:
            path = host_path()
            if not os.path.exists(path):
                from .. import _build

                _build.build_host()
            _host = ctypes.CDLL(path)
    return _host


def hip_available() -> bool:
    return os.path.exists(hip_path())
