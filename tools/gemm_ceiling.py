"""What the vendor GEMM reaches on the implicit-GEMM shapes of the compute-bound layers (a ceiling estimate for
the hand-written convolution main loop; the library multiplies a materialised [M, K] matrix, which the
convolution never has)."""
import torch

shapes = [("28x28 128->128 3x3", 802816, 128, 1152), ("14x14 256->256 3x3", 200704, 256, 2304),
          ("7x7 512->512 3x3", 50176, 512, 4608), ("14x14 1024->256 1x1", 200704, 256, 1024),
          ("14x14 256->1024 1x1", 200704, 1024, 256), ("7x7 2048->512 1x1", 50176, 512, 2048),
          ("square 8192", 8192, 8192, 8192)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        c = a @ b.t()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(10):
        c = a @ b.t()
    en.record()
    torch.cuda.synchronize()
    us = st.elapsed_time(en) * 100
    print("%-22s M=%-7d N=%-5d K=%-5d %8.1f us  %7.1f TF/s" % (name, M, N, K, us, 2.0 * M * N * K / us / 1e6), flush=True)
    del a, b, c
