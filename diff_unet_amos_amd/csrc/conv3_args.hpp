// Launch arguments shared by the conv3d_k3 kernel variants.
#pragma once
#include "common.hpp"

namespace dua {

struct Conv3Args {
  const void* x; const void* w; const float* bias; void* y;
  double* stats;
  InXform xf;
  int N, D, H, W;
  int Cin, Cin_stride, Cin_off;     // valid input channels, buffer stride, offset (elements)
  int Cout, Cout_stride, Cout_off;
  int nchunks, ntiles, tiles_h, tiles_w, cout_pad;
  int ksplit, units_per_split;      // split-K over (chunk, kd) units; partial tiles go to `part` in fp32
  float* part;                      // fp32 partial tiles [ks][n][voxel][cout_pad]: split-K, or the partial-sum form (ksplit 1)
  const float* init;                // fp32 [n][voxel][cout_pad] added to the accumulators at the start, or null
  int tap_ch;                       // single-channel tap form: packed index of that channel, else -1
};

}  // namespace dua
