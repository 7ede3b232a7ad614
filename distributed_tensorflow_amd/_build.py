"""In-tree native build.

Two shared libraries are produced under ``distributed_tensorflow_amd/_lib``:

* ``libdtfa_hip.so``  -- every ``csrc/hip/*.hip`` kernel file compiled with
  ``hipcc --offload-arch=gfx950`` (MI355X only; hipcc cross-compiles without a
  GPU), exposing ``extern "C"`` launchers that take raw device pointers and a
  ``hipStream_t``.
* ``libdtfa_host.so`` -- the host runtime in ``csrc/host/*.cc`` (crc32c,
  TFRecord, tf.Example codec, SSTable / tensor-bundle checkpoints, event files,
  baseline-JPEG entropy decoder, threaded record loader) compiled with g++.

The libraries are loaded with ctypes (see ``_lib``), so there is no dependency
on the PyTorch C++ headers and a full rebuild takes well under a minute.
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "csrc")
LIBDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib")
OBJDIR = os.path.join(ROOT, "build", "obj")

HIP_ARCH = "gfx950"
HIP_FLAGS = ["-O3", "-fPIC", "-std=c++17", "-fno-gpu-rdc", "-Wno-unused-result"]
HOST_FLAGS = ["-O3", "-fPIC", "-std=c++17", "-msse4.2", "-pthread", "-fno-math-errno", "-ffp-contract=off", "-Wall",
              "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; the HIP kernels cannot be built")


def _sources(sub: str, ext: str):
    d = os.path.join(CSRC, sub)
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(ext))


def _deps_digest(paths) -> str:
    h = hashlib.sha1()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(p.encode())
            h.update(f.read())
    return h.hexdigest()


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def _build_lib(name, srcs, headers, compile_cmd, link_cmd, force=False, verbose=False, jobs=8):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    out = os.path.join(LIBDIR, name)
    stamp = out + ".stamp"
    digest = _deps_digest(list(srcs) + list(headers)) + "|" + " ".join(compile_cmd("S", "O"))
    if not force and os.path.exists(out) and os.path.exists(stamp):
        with open(stamp) as f:
            if f.read() == digest:
                return out
    objs = []
    hdr_digest = _deps_digest(headers)

    def one(src):
        obj = os.path.join(OBJDIR, name + "." + os.path.basename(src) + ".o")
        ostamp = obj + ".stamp"
        d = _deps_digest([src]) + hdr_digest + " ".join(compile_cmd("S", "O"))
        if not force and os.path.exists(obj) and os.path.exists(ostamp) and open(ostamp).read() == d:
            return obj
        if verbose:
            print("[dtfa build] compiling", os.path.relpath(src, ROOT), flush=True)
        _run(compile_cmd(src, obj))
        with open(ostamp, "w") as f:
            f.write(d)
        return obj

    with cf.ThreadPoolExecutor(max_workers=max(1, min(jobs, len(srcs)))) as ex:
        objs = list(ex.map(one, srcs))
    if verbose:
        print("[dtfa build] linking", name, flush=True)
    _run(link_cmd(objs, out))
    with open(stamp, "w") as f:
        f.write(digest)
    return out


def build_hip(force=False, verbose=False) -> str:
    hipcc = _hipcc()
    srcs = _sources("hip", ".hip")
    headers = _sources("hip", ".h")

    def cc(src, obj):
        return [hipcc, "--offload-arch=" + HIP_ARCH, *HIP_FLAGS, "-c", src, "-o", obj]

    def ld(objs, out):
        return [hipcc, "--offload-arch=" + HIP_ARCH, "-shared", "-fPIC", *objs, "-o", out]

    return _build_lib("libdtfa_hip.so", srcs, headers, cc, ld, force, verbose)


def build_host(force=False, verbose=False) -> str:
    cxx = os.environ.get("CXX") or shutil.which("g++") or "g++"
    srcs = _sources("host", ".cc")
    headers = _sources("host", ".h")

    def cc(src, obj):
        return [cxx, *HOST_FLAGS, "-c", src, "-o", obj]

    def ld(objs, out):
        return [cxx, "-shared", "-fPIC", "-pthread", *objs, "-o", out]

    return _build_lib("libdtfa_host.so", srcs, headers, cc, ld, force, verbose)


def build_all(force=False, verbose=False):
    return build_host(force, verbose), build_hip(force, verbose)


if __name__ == "__main__":
    force = "--force" in sys.argv
    for p in build_all(force=force, verbose=True):
        print(p)
