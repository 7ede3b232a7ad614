"""Build an experimental variant of libdtfa_hip.so with extra compiler flags.

    python tools/build_variant.py dbg1 -DDTFA_DBG=1
    DTFA_HIP_LIB=variants/dbg1/libdtfa_hip.so python tools/bench_conv.py ...

Variants live in ``variants/<name>/`` (travels with gpurun snapshots, ignored by git)."""
import concurrent.futures as cf
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from distributed_tensorflow_amd import _build  # noqa: E402


def main():
    name, extra = sys.argv[1], sys.argv[2:]
    out_dir = os.path.join(ROOT, "variants", name)
    os.makedirs(out_dir, exist_ok=True)
    hipcc = _build._hipcc()
    srcs = _build._sources("hip", ".hip")

    def one(src):
        obj = os.path.join(out_dir, os.path.basename(src) + ".o")
        subprocess.run([hipcc, "--offload-arch=" + _build.HIP_ARCH, *_build.HIP_FLAGS, *extra, "-c", src, "-o", obj],
                       check=True)
        return obj

    with cf.ThreadPoolExecutor(8) as ex:
        objs = list(ex.map(one, srcs))
    out = os.path.join(out_dir, "libdtfa_hip.so")
    subprocess.run([hipcc, "--offload-arch=" + _build.HIP_ARCH, "-shared", "-fPIC", *objs, "-o", out], check=True)
    for o in objs:
        os.remove(o)
    print(out)


if __name__ == "__main__":
    main()
