// Boundary layout conversions and weight packing.
//
// The reference keeps every tensor NCDHW fp32 (torch default); the kernels here keep
// activations channels-last so that the K dimension of the implicit GEMM (channels) is the
// contiguous one.  These kernels run once per call at the API boundary
// (Diffusion.forward, models/diffusion/diffusion.py:49-63) or once per weight update.
#include "common.hpp"
#include "../../include/dua_hip.h"

namespace dua {

// ---- conv3 weights: [Cout][Cin_src][27] fp32 -> [ct][chunk][kd][t9][kg][64][EPG] ----
// zero_cp: packed input channel whose slab weights are forced to zero (the single-channel tap form keeps that channel's
// weights in its own block), or -1.
template <typename T>
__global__ void pack_conv3_kernel(int Cout, int Cin_src, int nchunks, const float* __restrict__ w,
                                  const int* __restrict__ perm, T* __restrict__ out, long total, int zero_cp) {
  constexpr int EPG = Elem<T>::EPG;
  constexpr int CK = 4 * EPG;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long t = i;
    const int e = (int)(t % EPG); t /= EPG;
    const int co_l = (int)(t % 64); t /= 64;
    const int kg = (int)(t % 4); t /= 4;
    const int t9 = (int)(t % 9); t /= 9;
    const int kd = (int)(t % 3); t /= 3;
    const int ch = (int)(t % nchunks); const int ct = (int)(t / nchunks);
    const int co = ct * 64 + co_l;
    const int cp = ch * CK + kg * EPG + e;            // packed input channel
    const int ci = perm ? perm[cp] : cp;
    float v = 0.f;
    if (co < Cout && ci >= 0 && ci < Cin_src && cp != zero_cp) v = w[((long)co * Cin_src + ci) * 27 + kd * 9 + t9];
    out[i] = (T)v;
  }
}

// ---- single-channel tap block: [ct][4 groups of 8 taps][64 couts][8] = w[co][ci][tap], taps 27..31 zero ----
template <typename T>
__global__ void pack_conv3_tap_kernel(int Cout, int Cin_src, int ci, const float* __restrict__ w, T* __restrict__ out, long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int e = (int)(i % 8), co_l = (int)((i / 8) % 64), g = (int)((i / 512) % 4), ct = (int)(i / 2048);
    const int co = ct * 64 + co_l, tap = g * 8 + e;
    float v = 0.f;
    if (co < Cout && tap < 27 && ci >= 0 && ci < Cin_src) v = w[((long)co * Cin_src + ci) * 27 + tap];
    out[i] = (T)v;
  }
}

// ---- the same layout for the DATA-GRADIENT convolution: dx = conv3(dy, W') with W'[ci][co][tap] = W[co][ci][26 - tap]
// (spatially flipped, in/out channels swapped), read straight from the forward weights [Cout][Cin][27] ----
template <typename T>
__global__ void pack_conv3_dgrad_kernel(int Cout, int Cin, int nchunks, const float* __restrict__ w,
                                        T* __restrict__ out, long total) {
  constexpr int EPG = Elem<T>::EPG;
  constexpr int CK = 4 * EPG;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long t = i;
    const int e = (int)(t % EPG); t /= EPG;
    const int o_l = (int)(t % 64); t /= 64;
    const int kg = (int)(t % 4); t /= 4;
    const int t9 = (int)(t % 9); t /= 9;
    const int kd = (int)(t % 3); t /= 3;
    const int ch = (int)(t % nchunks); const int ot = (int)(t / nchunks);
    const int ci = ot * 64 + o_l;                     // output channel of the gradient conv = forward input channel
    const int co = ch * CK + kg * EPG + e;            // its input channel = forward output channel
    float v = 0.f;
    if (ci < Cin && co < Cout) v = w[((long)co * Cin + ci) * 27 + 26 - (kd * 9 + t9)];
    out[i] = (T)v;
  }
}

// ---- the two packings above for fp16 with the identity channel map, through LDS (the training step repacks every layer's weights
// twice per step).  One thread per OUTPUT element reads w with lanes 27 floats apart: every source line is touched by 27 different
// waves and the 512 x 512 layer took 38 us for 42 MB.  Here a workgroup owns 8 output channels x one 32-channel chunk (forward) or
// 32 x 8 (data gradient) for all 27 taps: the rows of w arrive as contiguous 16-byte pieces, pass through LDS as fp16 in w's own
// order, and leave as 16-byte pieces of the packed layout, 128 contiguous bytes per (tap, k-group).  Cin % 4 == 0.
__device__ __forceinline__ void pack_conv3_rows_body(int bid, f16* st, int Cout, int Cin_src, int nchunks, const float* __restrict__ w,
                                                     f16* __restrict__ out) {
  // st: [co 8][ci 32][27]
  const int tid = threadIdx.x, cg = bid & 7, ch = (bid >> 3) % nchunks, ct = (bid >> 3) / nchunks;
  const int ci0 = ch * 32;
  const int nv = Cin_src - ci0 < 0 ? 0 : (Cin_src - ci0 > 32 ? 32 : Cin_src - ci0), npc = nv * 27 / 4;   // valid pieces per row
  f32x4 v[7];
#pragma unroll
  for (int u = 0; u < 7; ++u) {
    const int piece = tid + 256 * u, r = piece / 216, q = piece - r * 216, co = ct * 64 + cg * 8 + r;
    const bool ok = piece < 1728 && co < Cout && q < npc;
    v[u] = *(const f32x4*)(w + (ok ? ((long)co * Cin_src + ci0) * 27 + q * 4 : 0));
  }
#pragma unroll
  for (int u = 0; u < 7; ++u) {
    const int piece = tid + 256 * u, r = piece / 216, q = piece - r * 216, co = ct * 64 + cg * 8 + r;
    const bool ok = co < Cout && q < npc;
    if (piece < 1728) {
      f16x4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = ok ? (f16)v[u][e] : (f16)0.f;
      *(f16x4*)(st + r * 864 + q * 4) = h;
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int p = tid + 256 * u;
    if (p < 864) {
      const int r = p & 7, kg = (p >> 3) & 3, tap = p >> 5;
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = st[r * 864 + (kg * 8 + e) * 27 + tap];
      *(f16x8*)(out + ((((long)ct * nchunks + ch) * 27 + tap) * 4 + kg) * 512 + (cg * 8 + r) * 8) = o;
    }
  }
}

__global__ __launch_bounds__(256) void pack_conv3_rows_kernel(int Cout, int Cin_src, int nchunks, const float* __restrict__ w,
                                                              f16* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) f16 st[6912];
  pack_conv3_rows_body(blockIdx.x, st, Cout, Cin_src, nchunks, w, out);
}

__device__ __forceinline__ void pack_conv3_dgrad_rows_body(int bid, f16* st, int Cout, int Cin, int nchunks, const float* __restrict__ w,
                                                           f16* __restrict__ out) {
  // st: [co 32][ci 8][27]
  const int tid = threadIdx.x, cig = bid & 7, ch = (bid >> 3) % nchunks, ot = (bid >> 3) / nchunks;
  const int co0 = ch * 32, ci0 = ot * 64 + cig * 8;
  const int nv = Cin - ci0 < 0 ? 0 : (Cin - ci0 > 8 ? 8 : Cin - ci0), npc = nv * 27 / 4;
  f32x4 v[7];
#pragma unroll
  for (int u = 0; u < 7; ++u) {
    const int piece = tid + 256 * u, r = piece / 54, q = piece - r * 54, co = co0 + r;
    const bool ok = piece < 1728 && co < Cout && q < npc;
    v[u] = *(const f32x4*)(w + (ok ? ((long)co * Cin + ci0) * 27 + q * 4 : 0));
  }
#pragma unroll
  for (int u = 0; u < 7; ++u) {
    const int piece = tid + 256 * u, r = piece / 54, q = piece - r * 54, co = co0 + r;
    const bool ok = co < Cout && q < npc;
    if (piece < 1728) {
      f16x4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = ok ? (f16)v[u][e] : (f16)0.f;
      *(f16x4*)(st + r * 216 + q * 4) = h;
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int p = tid + 256 * u;
    if (p < 864) {
      const int r = p & 7, kg = (p >> 3) & 3, tap = p >> 5;                    // r: input channel of the forward conv, tap: of the gradient conv
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = st[(kg * 8 + e) * 216 + r * 27 + 26 - tap];
      *(f16x8*)(out + ((((long)ot * nchunks + ch) * 27 + tap) * 4 + kg) * 512 + (cig * 8 + r) * 8) = o;
    }
  }
}

__global__ __launch_bounds__(256) void pack_conv3_dgrad_rows_kernel(int Cout, int Cin, int nchunks, const float* __restrict__ w,
                                                                    f16* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) f16 st[6912];
  pack_conv3_dgrad_rows_body(blockIdx.x, st, Cout, Cin, nchunks, w, out);
}

// Many layers in one launch (a training step repacks all 28 convolutions twice: 56 launches of 3-12 us each before): workgroup
// -> (layer, its block) through the first-block table, then the body of the per-layer kernel.
constexpr int PACK_MAX = 64;
struct PackBatch {
  int count;
  int first_block[PACK_MAX + 1];
  int kind[PACK_MAX], Cout[PACK_MAX], Cin[PACK_MAX], nchunks[PACK_MAX];
  const float* w[PACK_MAX];
  f16* out[PACK_MAX];
};
__global__ __launch_bounds__(256) void pack_conv3_batch_kernel(PackBatch b) {
  __shared__ __attribute__((aligned(16))) f16 st[6912];
  int lo = 0, hi = b.count;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (b.first_block[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
  }
  const int bid = blockIdx.x - b.first_block[lo];
  if (b.kind[lo] == 0) pack_conv3_rows_body(bid, st, b.Cout[lo], b.Cin[lo], b.nchunks[lo], b.w[lo], b.out[lo]);
  else pack_conv3_dgrad_rows_body(bid, st, b.Cout[lo], b.Cin[lo], b.nchunks[lo], b.w[lo], b.out[lo]);
}

// ---- deconv k2 s2 weights: [Cin][Cout][8] fp32 -> [tap][ct][chunk][kg][64][EPG] ----
template <typename T>
__global__ void pack_deconv_kernel(int Cin, int Cout, int nchunks, int nct, const float* __restrict__ w,
                                   T* __restrict__ out, long total) {
  constexpr int EPG = Elem<T>::EPG;
  constexpr int CK = 4 * EPG;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long t = i;
    const int e = (int)(t % EPG); t /= EPG;
    const int co_l = (int)(t % 64); t /= 64;
    const int kg = (int)(t % 4); t /= 4;
    const int ch = (int)(t % nchunks); t /= nchunks;
    const int ct = (int)(t % nct); const int tap = (int)(t / nct);
    const int co = ct * 64 + co_l, ci = ch * CK + kg * EPG + e;
    float v = 0.f;
    if (co < Cout && ci < Cin) v = w[((long)ci * Cout + co) * 8 + tap];
    out[i] = (T)v;
  }
}

// ---- NCDHW fp32 -> channels-last T, channel slice [c_off, c_off+C) of a buffer with stride Cs ----
// One thread per (voxel, channel): reads are strided across lanes by voxel (coalesced along W
// for fixed c when the inner loop walks channels), writes land in the voxel's channel run.
template <typename T>
__global__ void to_channels_last_kernel(const float* __restrict__ src, int C, long vox, T* __restrict__ dst, int Cs,
                                        int c_off, int Cfill) {
  const int n = blockIdx.y;
  for (long v = blockIdx.x * 256L + threadIdx.x; v < vox; v += (long)gridDim.x * 256) {
    T* o = dst + ((long)n * vox + v) * Cs + c_off;
    for (int c = 0; c < C; ++c) o[c] = (T)src[((long)n * C + c) * vox + v];
    for (int c = C; c < Cfill; ++c) o[c] = (T)0.f;
  }
}

// Whole rows at once: the channels of up to two NCDHW sources side by side (torch.cat((image, x), dim=1), denoiser.py:298), zero
// fill up to the row width, written in 16-byte pieces.  The per-channel form above stores one 2-byte element at a time -- and the
// second part of a concatenation starts at an odd channel: 240 us for the 16-channel x_t of a config-4 step (2 x 96^3 voxels)
// against ~40 for the bytes moved.
template <typename T, int PIECES>
__global__ __launch_bounds__(256) void to_channels_last_rows_kernel(const float* __restrict__ s0, int C0, const float* __restrict__ s1,
                                                                    int C1, long vox, T* __restrict__ dst) {
  constexpr int EPP = 16 / (int)sizeof(T), CS = PIECES * EPP;
  using Piece = typename Elem<T>::Frag;
  const int n = blockIdx.y;
  for (long v = blockIdx.x * 256L + threadIdx.x; v < vox; v += (long)gridDim.x * 256) {
    float f[CS];
#pragma unroll
    for (int c = 0; c < CS; ++c) {
      f[c] = 0.f;
      if (c < C0) f[c] = s0[((long)n * C0 + c) * vox + v];
      else if (c < C0 + C1) f[c] = s1[((long)n * C1 + (c - C0)) * vox + v];
    }
    Piece* o = (Piece*)(dst + ((long)n * vox + v) * CS);
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      Piece w;
#pragma unroll
      for (int e = 0; e < EPP; ++e) w[e] = (T)f[p * EPP + e];
      o[p] = w;
    }
  }
}

template <typename T>
__global__ void from_channels_last_kernel(const T* __restrict__ src, int Cs, int c_off, int C, long vox,
                                          float* __restrict__ dst) {
  const int n = blockIdx.y;
  for (long v = blockIdx.x * 256L + threadIdx.x; v < vox; v += (long)gridDim.x * 256) {
    const T* i = src + ((long)n * vox + v) * Cs + c_off;
    for (int c = 0; c < C; ++c) dst[((long)n * C + c) * vox + v] = (float)i[c];
  }
}

static inline unsigned nblocks(long total) {
  long b = (total + 255) / 256;
  return (unsigned)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

template <typename T, int PIECES>
static int launch_rows(int N, const float* s0, int C0, const float* s1, int C1, long voxels, void* dst, void* stream) {
  hipLaunchKernelGGL((to_channels_last_rows_kernel<T, PIECES>), dim3(nblocks(voxels), N), dim3(256), 0, (hipStream_t)stream,
                     s0, C0, s1, C1, voxels, (T*)dst);
  return (int)hipGetLastError();
}

}  // namespace dua

extern "C" {

static long pack_conv3_common(int dtype, int Cout, int Cin_src, int Cin_packed, int tap_channel, const float* w,
                              const int* in_perm, int tap_src, void* w_packed, void* stream);

long dua_pack_conv3_weights(int dtype, int Cout, int Cin_src, int Cin_packed, const float* w, const int* in_perm,
                            void* w_packed, void* stream) {
  return pack_conv3_common(dtype, Cout, Cin_src, Cin_packed, -1, w, in_perm, -1, w_packed, stream);
}

long dua_pack_conv3_weights_tap(int dtype, int Cout, int Cin_src, int Cin_packed, int tap_channel, int tap_src_channel,
                                const float* w, const int* in_perm, void* w_packed, void* stream) {
  if (dtype != DUA_F16 || Cin_packed > 32 || tap_channel < 0 || tap_channel >= Cin_packed || tap_src_channel < 0 ||
      tap_src_channel >= Cin_src)
    return DUA_ERR_ARG;
  return pack_conv3_common(dtype, Cout, Cin_src, Cin_packed, tap_channel, w, in_perm, tap_src_channel, w_packed, stream);
}

static long pack_conv3_common(int dtype, int Cout, int Cin_src, int Cin_packed, int tap_channel, const float* w,
                              const int* in_perm, int tap_src, void* w_packed, void* stream) {
  const int epg = dtype == DUA_F16 ? 8 : 4, ck = 4 * epg;
  if ((dtype != DUA_F16 && dtype != DUA_F32) || Cout <= 0 || Cin_src <= 0 || Cin_packed <= 0) return DUA_ERR_ARG;
  const int nchunks = (Cin_packed + ck - 1) / ck, nct = (Cout + 63) / 64;
  const long total = (long)nct * nchunks * 27 * 4 * 64 * epg;
  const long tap_total = tap_channel >= 0 ? (long)nct * 4 * 64 * 8 : 0;
  const long bytes = (total + tap_total) * (dtype == DUA_F16 ? 2 : 4);
  if (!w_packed) return bytes;
  if (!w) return DUA_ERR_ARG;
  if (tap_channel >= 0)
    hipLaunchKernelGGL(dua::pack_conv3_tap_kernel<dua::f16>, dim3(dua::nblocks(tap_total)), dim3(256), 0, (hipStream_t)stream,
                       Cout, Cin_src, tap_src, w, (dua::f16*)w_packed + total, tap_total);
  // in_perm must cover nchunks*ck entries when given
  if (dtype == DUA_F16 && !in_perm && tap_channel < 0 && Cin_src % 4 == 0 && ((size_t)w & 15) == 0)
    hipLaunchKernelGGL(dua::pack_conv3_rows_kernel, dim3(nct * nchunks * 8), dim3(256), 0, (hipStream_t)stream, Cout, Cin_src, nchunks,
                       w, (dua::f16*)w_packed);
  else if (dtype == DUA_F16)
    hipLaunchKernelGGL(dua::pack_conv3_kernel<dua::f16>, dim3(dua::nblocks(total)), dim3(256), 0, (hipStream_t)stream,
                       Cout, Cin_src, nchunks, w, in_perm, (dua::f16*)w_packed, total, tap_channel);
  else
    hipLaunchKernelGGL(dua::pack_conv3_kernel<float>, dim3(dua::nblocks(total)), dim3(256), 0, (hipStream_t)stream,
                       Cout, Cin_src, nchunks, w, in_perm, (float*)w_packed, total, tap_channel);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? bytes : -(long)e;
}

long dua_pack_conv3_weights_dgrad(int dtype, int Cout, int Cin, int Cout_packed, const float* w, void* w_packed,
                                  void* stream) {
  const int epg = dtype == DUA_F16 ? 8 : 4, ck = 4 * epg;
  if ((dtype != DUA_F16 && dtype != DUA_F32) || Cout <= 0 || Cin <= 0 || Cout_packed < Cout) return DUA_ERR_ARG;
  const int nchunks = (Cout_packed + ck - 1) / ck, not_ = (Cin + 63) / 64;
  const long total = (long)not_ * nchunks * 27 * 4 * 64 * epg;
  const long bytes = total * (dtype == DUA_F16 ? 2 : 4);
  if (!w_packed) return bytes;
  if (!w) return DUA_ERR_ARG;
  if (dtype == DUA_F16 && Cin % 4 == 0 && ((size_t)w & 15) == 0)
    hipLaunchKernelGGL(dua::pack_conv3_dgrad_rows_kernel, dim3(not_ * nchunks * 8), dim3(256), 0, (hipStream_t)stream, Cout, Cin, nchunks,
                       w, (dua::f16*)w_packed);
  else if (dtype == DUA_F16)
    hipLaunchKernelGGL(dua::pack_conv3_dgrad_kernel<dua::f16>, dim3(dua::nblocks(total)), dim3(256), 0, (hipStream_t)stream,
                       Cout, Cin, nchunks, w, (dua::f16*)w_packed, total);
  else
    hipLaunchKernelGGL(dua::pack_conv3_dgrad_kernel<float>, dim3(dua::nblocks(total)), dim3(256), 0, (hipStream_t)stream,
                       Cout, Cin, nchunks, w, (float*)w_packed, total);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? bytes : -(long)e;
}

int dua_pack_conv3_weights_batch(int dtype, int count, const dua_pack_item* items, void* stream) {
  if (dtype != DUA_F16 || count <= 0 || count > dua::PACK_MAX || !items) return DUA_ERR_ARG;
  dua::PackBatch b;
  b.count = count;
  long blocks = 0;
  for (int i = 0; i < count; ++i) {
    const dua_pack_item& it = items[i];
    if ((it.kind != 0 && it.kind != 1) || it.Cout <= 0 || it.Cin <= 0 || it.Cin % 4 || !it.w || !it.out ||
        (((size_t)it.w) & 15) || (((size_t)it.out) & 15))
      return DUA_ERR_ARG;
    if (it.packed < (it.kind == 0 ? it.Cin : it.Cout)) return DUA_ERR_ARG;
    const int nchunks = (it.packed + 31) / 32, tiles = ((it.kind == 0 ? it.Cout : it.Cin) + 63) / 64;
    b.first_block[i] = (int)blocks;
    b.kind[i] = it.kind; b.Cout[i] = it.Cout; b.Cin[i] = it.Cin; b.nchunks[i] = nchunks;
    b.w[i] = it.w; b.out[i] = (dua::f16*)it.out;
    blocks += (long)tiles * nchunks * 8;
    if (blocks > 0x7fffffffL) return DUA_ERR_ARG;
  }
  for (int i = count; i <= dua::PACK_MAX; ++i) b.first_block[i] = (int)blocks;
  hipLaunchKernelGGL(dua::pack_conv3_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, b);
  return (int)hipGetLastError();
}

long dua_pack_deconv_weights(int dtype, int Cin, int Cout, const float* w, void* w_packed, void* stream) {
  const int epg = dtype == DUA_F16 ? 8 : 4, ck = 4 * epg;
  if ((dtype != DUA_F16 && dtype != DUA_F32) || Cout <= 0 || Cin <= 0) return DUA_ERR_ARG;
  const int nchunks = (Cin + ck - 1) / ck, nct = (Cout + 63) / 64;
  const long total = 8L * nct * nchunks * 4 * 64 * epg;
  const long bytes = total * (dtype == DUA_F16 ? 2 : 4);
  if (!w_packed) return bytes;
  if (!w) return DUA_ERR_ARG;
  if (dtype == DUA_F16)
    hipLaunchKernelGGL(dua::pack_deconv_kernel<dua::f16>, dim3(dua::nblocks(total)), dim3(256), 0, (hipStream_t)stream,
                       Cin, Cout, nchunks, nct, w, (dua::f16*)w_packed, total);
  else
    hipLaunchKernelGGL(dua::pack_deconv_kernel<float>, dim3(dua::nblocks(total)), dim3(256), 0, (hipStream_t)stream,
                       Cin, Cout, nchunks, nct, w, (float*)w_packed, total);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? bytes : -(long)e;
}

int dua_to_channels_last(int dtype, int N, int C, long voxels, const float* src, void* dst, int Cstride, int C_off,
                         int C_fill, void* stream) {
  if (!src || !dst || N <= 0 || C <= 0 || voxels <= 0 || C_off + (C_fill > C ? C_fill : C) > Cstride) return DUA_ERR_ARG;
  dim3 grid(dua::nblocks(voxels), N);
  if (dtype == DUA_F16)
    hipLaunchKernelGGL(dua::to_channels_last_kernel<dua::f16>, grid, dim3(256), 0, (hipStream_t)stream, src, C, voxels,
                       (dua::f16*)dst, Cstride, C_off, C_fill);
  else if (dtype == DUA_F32)
    hipLaunchKernelGGL(dua::to_channels_last_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, src, C, voxels,
                       (float*)dst, Cstride, C_off, C_fill);
  else return DUA_ERR_ARG;
  return (int)hipGetLastError();
}

int dua_to_channels_last_rows(int dtype, int N, int C0, const float* src0, int C1, const float* src1, long voxels, void* dst,
                              int Cstride, void* stream) {
  const int esz = dtype == DUA_F16 ? 2 : 4;
  if ((dtype != DUA_F16 && dtype != DUA_F32) || !src0 || C0 <= 0 || C1 < 0 || (C1 > 0 && !src1) || !dst || N <= 0 || voxels <= 0 ||
      C0 + C1 > Cstride || (Cstride * esz) % 16 || Cstride * esz > 64 || (((size_t)dst) & 15))
    return DUA_ERR_ARG;
  const int pieces = Cstride * esz / 16;
  if (dtype == DUA_F16) {
    switch (pieces) {
      case 1: return dua::launch_rows<dua::f16, 1>(N, src0, C0, src1, C1, voxels, dst, stream);
      case 2: return dua::launch_rows<dua::f16, 2>(N, src0, C0, src1, C1, voxels, dst, stream);
      case 3: return dua::launch_rows<dua::f16, 3>(N, src0, C0, src1, C1, voxels, dst, stream);
      default: return dua::launch_rows<dua::f16, 4>(N, src0, C0, src1, C1, voxels, dst, stream);
    }
  }
  switch (pieces) {
    case 1: return dua::launch_rows<float, 1>(N, src0, C0, src1, C1, voxels, dst, stream);
    case 2: return dua::launch_rows<float, 2>(N, src0, C0, src1, C1, voxels, dst, stream);
    case 3: return dua::launch_rows<float, 3>(N, src0, C0, src1, C1, voxels, dst, stream);
    default: return dua::launch_rows<float, 4>(N, src0, C0, src1, C1, voxels, dst, stream);
  }
}

int dua_from_channels_last(int dtype, int N, int C, long voxels, const void* src, int Cstride, int C_off, float* dst,
                           void* stream) {
  if (!src || !dst || N <= 0 || C <= 0 || voxels <= 0 || C_off + C > Cstride) return DUA_ERR_ARG;
  dim3 grid(dua::nblocks(voxels), N);
  if (dtype == DUA_F16)
    hipLaunchKernelGGL(dua::from_channels_last_kernel<dua::f16>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const dua::f16*)src, Cstride, C_off, C, voxels, dst);
  else if (dtype == DUA_F32)
    hipLaunchKernelGGL(dua::from_channels_last_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const float*)src, Cstride, C_off, C, voxels, dst);
  else return DUA_ERR_ARG;
  return (int)hipGetLastError();
}

}  // extern "C"
