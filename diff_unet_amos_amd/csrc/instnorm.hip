// InstanceNorm3d(affine) statistics finalisation and the "materialise" pass.
//
// The convolution kernels leave, per batch item and 64-voxel slab, (sum, centred M2) of their
// raw output.  instnorm_finalize combines the slabs in fp64 (Chan's parallel variance) into
// the biased variance nn.InstanceNorm3d uses and emits scale = gamma*rstd, shift = beta -
// mean*scale, which consumers apply while staging their input (common.hpp InXform).
//
// materialize writes an activation that has several consumers: x_i = LeakyReLU(IN(raw)) +
// embeddings[i] (models/basic_unet/denoiser.py:300-304) into the channel slice of a concat
// buffer, and optionally its MaxPool3d(2) (denoiser.py:100,106) for the next level.
#include "common.hpp"
#include "../../include/dua_hip.h"

namespace dua {

// grid (ceil(C/8), N), block 256 = 32 row-lanes x 8 channels
__global__ __launch_bounds__(256) void instnorm_finalize_kernel(int C, int rows, int c_pad, const float2* partials,
                                                                const float* counts, const float* gamma,
                                                                const float* beta, float eps, float* scale,
                                                                float* shift) {
  __shared__ double sS[32][8], sQ[32][8], sN[32][8];
  const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl, n = blockIdx.y;
  double S = 0, Q = 0, Nn = 0;
  if (c < C) {
    const float2* p = partials + (long)n * rows * c_pad + c;
    for (int rr = rl; rr < rows; rr += 32) {
      const float k = counts[rr];
      if (k > 0.f) {
        const float2 v = p[(long)rr * c_pad];
        S += (double)v.x;
        Q += (double)v.y + (double)v.x * (double)v.x / (double)k;
        Nn += (double)k;
      }
    }
  }
  sS[rl][cl] = S; sQ[rl][cl] = Q; sN[rl][cl] = Nn;
  __syncthreads();
  if (rl == 0 && c < C) {
    for (int j = 1; j < 32; ++j) { S += sS[j][cl]; Q += sQ[j][cl]; Nn += sN[j][cl]; }
    const double mean = S / Nn;
    double var = (Q - S * mean) / Nn;
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma[c] * rstd;
    scale[n * C + c] = g;
    shift[n * C + c] = beta[c] - (float)mean * g;
  }
}

// One thread = one 16-byte channel group of one OUTPUT voxel (pooled: of one 2x2x2 block).
template <typename T, bool POOL>
__global__ __launch_bounds__(256) void materialize_kernel(const T* __restrict__ raw, int C, int raw_stride,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift, float slope,
                                                          const T* __restrict__ emb, int emb_stride, T* __restrict__ out,
                                                          int out_stride, int out_off, T* __restrict__ pooled,
                                                          int pool_stride, int D, int H, int W, long total) {
  using Frag = typename Elem<T>::Frag;
  constexpr int EPG = Elem<T>::EPG;
  const int gpc = C / EPG;
  const int n = blockIdx.y;
  const long vox_n = (long)D * H * W;
  for (long it = blockIdx.x * 256L + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
    const int cg = (int)(it % gpc);
    long v = it / gpc;
    float sc[EPG], sh[EPG];
#pragma unroll
    for (int e = 0; e < EPG; ++e) { sc[e] = scale[n * C + cg * EPG + e]; sh[e] = shift[n * C + cg * EPG + e]; }
    if constexpr (!POOL) {
      const long gv = n * vox_n + v;
      Frag x = *(const Frag*)(raw + gv * raw_stride + cg * EPG);
      Frag o;
      Frag ev;
      if (emb) ev = *(const Frag*)(emb + gv * emb_stride + cg * EPG);
#pragma unroll
      for (int e = 0; e < EPG; ++e) {
        float y = fmaf((float)x[e], sc[e], sh[e]);
        y = y > 0.f ? y : y * slope;
        if (emb) y += (float)ev[e];
        o[e] = (T)y;
      }
      *(Frag*)(out + gv * out_stride + out_off + cg * EPG) = o;
    } else {
      const int W2 = W >> 1, H2 = H >> 1;
      const int pw = (int)(v % W2); v /= W2;
      const int ph = (int)(v % H2); const int pd = (int)(v / H2);
      float mx[EPG];
#pragma unroll
      for (int e = 0; e < EPG; ++e) mx[e] = -INFINITY;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int d = 2 * pd + (k >> 2), h = 2 * ph + ((k >> 1) & 1), w = 2 * pw + (k & 1);
        const long gv = n * vox_n + ((long)d * H + h) * W + w;
        Frag x = *(const Frag*)(raw + gv * raw_stride + cg * EPG);
        Frag ev;
        if (emb) ev = *(const Frag*)(emb + gv * emb_stride + cg * EPG);
        Frag o;
#pragma unroll
        for (int e = 0; e < EPG; ++e) {
          float y = fmaf((float)x[e], sc[e], sh[e]);
          y = y > 0.f ? y : y * slope;
          if (emb) y += (float)ev[e];
          o[e] = (T)y;
          mx[e] = fmaxf(mx[e], (float)o[e]);
        }
        *(Frag*)(out + gv * out_stride + out_off + cg * EPG) = o;
      }
      Frag po;
#pragma unroll
      for (int e = 0; e < EPG; ++e) po[e] = (T)mx[e];
      const long pv = n * (vox_n >> 3) + ((long)pd * H2 + ph) * W2 + pw;
      *(Frag*)(pooled + pv * pool_stride + cg * EPG) = po;
    }
  }
}

template <typename T>
static int launch_materialize(const dua_materialize_desc* d, const void* raw, const float* scale, const float* shift,
                              const void* emb, void* out, void* pooled, hipStream_t s) {
  constexpr int EPG = Elem<T>::EPG;
  const long vox = (long)d->D * d->H * d->W;
  const bool pool = pooled != nullptr;
  const long total = (pool ? vox / 8 : vox) * (d->C / EPG);
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  dim3 grid((unsigned)blocks, d->N);
  if (pool)
    hipLaunchKernelGGL((materialize_kernel<T, true>), grid, dim3(256), 0, s, (const T*)raw, d->C, d->raw_stride, scale,
                       shift, d->slope, (const T*)emb, d->emb_stride, (T*)out, d->out_stride, d->out_off, (T*)pooled,
                       d->pool_stride, d->D, d->H, d->W, total);
  else
    hipLaunchKernelGGL((materialize_kernel<T, false>), grid, dim3(256), 0, s, (const T*)raw, d->C, d->raw_stride, scale,
                       shift, d->slope, (const T*)emb, d->emb_stride, (T*)out, d->out_stride, d->out_off, (T*)nullptr,
                       0, d->D, d->H, d->W, total);
  return (int)hipGetLastError();
}

}  // namespace dua

extern "C" {

int dua_instnorm_finalize(int N, int C, int rows, int c_pad, const float* partials, const float* counts,
                          const float* gamma, const float* beta, float eps, float* scale, float* shift, void* stream) {
  if (N <= 0 || C <= 0 || rows <= 0 || c_pad < C || !partials || !counts || !gamma || !beta || !scale || !shift)
    return DUA_ERR_ARG;
  dim3 grid((C + 7) / 8, N);
  hipLaunchKernelGGL(dua::instnorm_finalize_kernel, grid, dim3(256), 0, (hipStream_t)stream, C, rows, c_pad,
                     (const float2*)partials, counts, gamma, beta, eps, scale, shift);
  return (int)hipGetLastError();
}

int dua_materialize(const dua_materialize_desc* d, const void* raw, const float* scale, const float* shift,
                    const void* emb, void* out, void* pooled, void* stream) {
  if (!d || !raw || !scale || !shift || !out) return DUA_ERR_ARG;
  if (d->C % 8 || d->raw_stride % 8 || d->out_stride % 8 || d->out_off % 8 || (emb && d->emb_stride % 8)) return DUA_ERR_ARG;
  if (pooled && ((d->D | d->H | d->W) & 1 || d->pool_stride % 8)) return DUA_ERR_ARG;
  if (d->dtype == DUA_F16) return dua::launch_materialize<dua::f16>(d, raw, scale, shift, emb, out, pooled, (hipStream_t)stream);
  if (d->dtype == DUA_F32) return dua::launch_materialize<float>(d, raw, scale, shift, emb, out, pooled, (hipStream_t)stream);
  return DUA_ERR_ARG;
}

}  // extern "C"
