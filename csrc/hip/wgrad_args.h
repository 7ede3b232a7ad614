// Argument block of the weight-gradient kernels (conv_wgrad.hip, conv_wp3.hip).
#pragma once
#include "common.h"

struct WgradArgs {
  const bf16* x;     // [B, IH, IW, Cin]
  const bf16* dy;    // [B, OH, OW, Cout] dense, M = B*OH*OW rows
  float* slab;       // [splits][Cout][K]
  const bf16* zero;
  const int* taps;   // (dy << 16) | (dx & 0xffff)
  int IH, IW, Cin;
  int OH, OW, Cout;
  int isy, isx;
  int T, M, K;
  unsigned cin_magic;
  int ktiles_per_split;
  int splits;
  // decomposition of one k-tile step (BKP pixels) into (images, rows, cols)
  int step_b, step_y, step_x;
  // floats between the slabs of consecutive splits: Cout * K, plus K when the launch also sums the columns of X
  // (``colsum``: sum_m X[pixel(m) + tap][ci], written behind the split's [Cout][K] block)
  long long slab_stride;
  int colsum;
};

#ifndef MAX_TAPS
#define MAX_TAPS 64
#endif
