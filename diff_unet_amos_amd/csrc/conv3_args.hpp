// Launch arguments shared by the conv3d_k3 kernel variants.
#pragma once
#include "common.hpp"

namespace dua {

struct Conv3Args {
  const void* x; const void* w; const float* bias; void* y;
  stat_t* stats;
  InXform xf;
  int N, D, H, W;
  int Cin, Cin_stride, Cin_off;     // valid input channels, buffer stride, offset (elements)
  int Cout, Cout_stride, Cout_off;
  int nchunks, ntiles, tiles_h, tiles_w, cout_pad;
  int ksplit, units_per_split;      // split-K over (chunk, kd) units; partial tiles go to `part` in fp32
  float* part;                      // split-K: fp32 partial tiles [ks][n][voxel][cout_pad], else null
  int tap_ch;                       // single-channel tap form: packed index of that channel, else -1
  int in_blk, out_blk;              // 16-channel-blocked input / output buffers (wide-tile form and its producers)
  // Training, data-gradient launches of the wide-tile form (dua_conv3d_k3_dgrad_reduce): y = dA of the layer whose raw output is
  // bw_raw (channels-last, bw_stride / bw_off) and whose forward statistics + affine parameters are bw_xf; the epilogue adds the
  // three sums of that layer's InstanceNorm backward (in_bwd_reduce_kernel: sum dA, sum dZ, sum dZ zhat) to bw_sums INSTEAD of
  // this launch's own (sum y, sum y^2).  bw_sums null: the ordinary forward epilogue.
  const void* bw_raw; int bw_stride, bw_off;
  InXform bw_xf;
  double* bw_sums;
};

// conv3d_wide.hip: the 8-accumulator form (8x8x8 tiles) for fp16 layers with >= 1024 tiles of 4x8x8; D, H, W multiples of 8,
// Cin a multiple of 16.
// persistent (policies 8 / 9, A/B only): workgroups walk tiles (conv3d_k3_wide_pt_kernel); false = one tile per workgroup (shipped)
int launch_conv3_wide(Conv3Args a, int D, hipStream_t s, bool persistent, int stagger = 0);
bool conv3_wide_takes_bwd_sums(const Conv3Args& a);     // the shipped one-tile-per-workgroup form, 64-wide output tiles

}  // namespace dua
