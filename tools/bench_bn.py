"""Bandwidth of the batch-norm kernels on ResNet-50 activation shapes (batch as argv[1], default 256)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import kernels as K  # noqa: E402
from tools.bench_conv import timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SHAPES = [(112, 64), (56, 64), (56, 256), (28, 128), (28, 512), (14, 256), (14, 1024), (7, 512), (7, 2048)]
print("%-14s | %-22s | %-22s | %-22s | %-22s" % ("shape", "apply relu (2N)", "apply relu+res (3N)", "bwd recompute (2N+3N)",
                                                "bwd z+dres (3N+5N)"))
for hw, c in SHAPES:
    M = B * hw * hw
    nbytes = M * c * 2
    y = torch.randn(M, c, device="cuda").bfloat16()
    res = torch.randn(M, c, device="cuda").bfloat16()
    dz = torch.randn(M, c, device="cuda").bfloat16()
    gamma = torch.ones(c, device="cuda")
    beta = torch.zeros(c, device="cuda")
    part = K.bn_stats_partials(y)
    mean, invstd, scale, shift = K.bn_finalize(part, M, 1e-5, gamma, beta, None, None, 0.1)
    z = K.bn_apply(y, scale, shift, True, res)
    t1 = timeit(lambda: K.bn_apply(y, scale, shift, True), 10)
    t2 = timeit(lambda: K.bn_apply(y, scale, shift, True, res), 10)
    t3 = timeit(lambda: K.bn_backward(dz, y, None, mean, invstd, gamma, True, False, scale=scale, shift=shift), 10)
    t4 = timeit(lambda: K.bn_backward(dz, y, z, mean, invstd, gamma, True, True), 10)

    def f(t, n):
        return "%7.1f us %5.2f TB/s" % (t * 1e6, n * nbytes / t / 1e12)

    print("%3dx%-3d c%-5d | %s | %s | %s | %s" % (hw, hw, c, f(t1, 2), f(t2, 3), f(t3, 5), f(t4, 8)), flush=True)
