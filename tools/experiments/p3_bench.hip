// Standalone timing of conv_p3_kernel on the ResNet-50 56x56 64 -> 64 layer (no framework): prints microseconds per
// launch for the statistics epilogue and for the plain data-gradient epilogue with an addend.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DP3_DBG=n] tools/experiments/p3_bench.hip -o /tmp/p3_bench
#include "../../csrc/hip/conv_p3.hip"

#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 2048, H = 56, W = 56, C = 64;
  const size_t M = (size_t)B * H * W;
  bf16 *x, *w, *y, *bny, *zero;
  float *stats, *vec;
  int* taps;
  hipMalloc(&x, M * C * 2); hipMalloc(&y, M * C * 2); hipMalloc(&bny, M * C * 2);
  hipMalloc(&w, 64 * 576 * 2); hipMalloc(&zero, 256); hipMemset(zero, 0, 256);
  hipMalloc(&stats, (M / 128 + 1) * 2 * 64 * 4); hipMalloc(&vec, 64 * 4 * 4);
  hipMalloc(&taps, 64 * 4);
  hipMemset(x, 0x3c, M * C * 2); hipMemset(bny, 0x3c, M * C * 2); hipMemset(w, 0x3b, 64 * 576 * 2); hipMemset(vec, 0, 1024);
  std::vector<int> ht;
  for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) ht.push_back((dy << 16) | (dx & 0xffff));
  hipMemcpy(taps, ht.data(), 36, hipMemcpyHostToDevice);
  ConvArgs a = {};
  a.x = x; a.w = w; a.y = y; a.zero = zero; a.taps = taps; a.stats = stats;
  a.IH = a.OH = a.OYn = H; a.IW = a.OW = a.OXn = W; a.Cin = a.Cout = C;
  a.osy = a.osx = a.isy = a.isx = 1; a.T = 9; a.M = (int)M; a.K = 576; a.dense_out = 1; a.ldo = C; a.K1 = 576;
  a.stream_out = 1;
#if (P3_DBG & 8)
  long long* dbg;
  hipMalloc(&dbg, 1024 * 4 * 5 * 8); hipMemset(dbg, 0, 1024 * 4 * 5 * 8);
  a.oscale = (const float*)dbg;
#endif
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    if (mode == 1) { a.stats = nullptr; a.addend = bny; }   // plain data gradient with a residual gradient added
    for (int i = 0; i < 3; ++i) if (int rc = conv_p3_launch(a, 0)) { printf("launch failed %d\n", rc); return 1; }
    hipEventRecord(e0, 0);
    const int n = 20;
    for (int i = 0; i < n; ++i) conv_p3_launch(a, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("P3_DBG=%d %s: %.1f us\n", P3_DBG, mode ? "dgrad+addend" : "fwd+stats", ms * 1000 / n);
#if (P3_DBG & 8)
    {
      std::vector<long long> h(1024 * 4 * 5);
      hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
      const int tiles = (int)((M / 128 + 511) / 512);
      for (int wg : {0, 1, 255, 300}) for (int wv : {0, 3}) {
        printf("  wg %d wave %d cycles per tile:", wg, wv);
        for (int i = 0; i < 5; ++i) printf(" %lld", h[(wg * 4 + wv) * 5 + i] / tiles);
        printf("   (top, taps, patch wait, barrier, epilogue)\n");
      }
    }
#endif
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("device error\n"); return 1; }
  return 0;
}
