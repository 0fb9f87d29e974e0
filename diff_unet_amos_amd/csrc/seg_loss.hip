// Segmentation loss of the training step and its gradient, fused:  losses/loss.py:25-86 for any subset of the names
// "mse", "bce", "dice" (weights w_* = 0 or 1) under loss_combine "sum" / "mean" / "log" (the combine's derivative
// reaches the kernel through *gscale); the diffusion configs use all three with "sum", i.e.
//   L = mean((sigmoid(p) - y)^2) + mean(BCEWithLogits(p, y)) + mean_{n,c}(1 - (2 I + e) / (S + Y + e)),
//   I = sum_v s*y, S = sum_v s, Y = sum_v y, e = 1e-5   (MONAI DiceLoss(sigmoid=True) defaults)
// p: logits channels-last [N][V][C] (compute dtype), y: labels NCDHW fp32 [N][C][V] as the reference's loader hands them.
//   reduce: sums[N*C*4 + 2] (fp64, pre-zeroed) += (I, S, Y, -) per (n, c), then (sum of squared errors, sum of BCE terms)
//   grad  : dp = g * [ (w_mse 2 (s - y) s (1 - s) + w_bce (s - y)) / M  +  w_dice dice'_{n,c} * s (1 - s) ],  M = N*C*V,
//           dice'_{n,c} = -(2 y (D + e) - (2 I + e)) / (D + e)^2 / (N C),  D = S + Y;  g = *gscale (device scalar)
// Both are one streaming pass (HBM bound): a thread owns one voxel and walks the C channels; labels are read coalesced
// per channel, logits as one contiguous run per voxel.
#include "common.hpp"
#include "../../include/dua_hip.h"

namespace dua {

constexpr int LOSS_MAXC = 64;

template <typename T>
__global__ __launch_bounds__(256) void seg_loss_reduce_kernel(const T* __restrict__ p, int p_stride,
                                                              const float* __restrict__ y, int C, long V,
                                                              double* __restrict__ sums, int N) {
  __shared__ float red[4][3 * LOSS_MAXC + 2];
  const int n = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float mse = 0.f, bce = 0.f;
  // per-channel partials live in LDS (C can be 16..64; registers indexed dynamically would spill)
  for (int i = threadIdx.x; i < 4 * (3 * LOSS_MAXC + 2); i += 256) (&red[0][0])[i] = 0.f;
  __syncthreads();
  // Eight channels at a time when the logits rows allow 16-byte reads (fp16, C and the row stride multiples of 8): one load brings
  // a voxel's eight logits, the labels of each channel are read coalesced.  One channel at a time (below) touched every logits
  // line C times with 2-byte loads: 164 us for the 170 MB of the config-4 step.
  const bool vec8 = sizeof(T) == 2 && C % 8 == 0 && p_stride % 8 == 0 && ((size_t)p & 15) == 0;
  for (int c0 = 0; vec8 && c0 < C; c0 += 8) {
    float I[8], S[8], Y[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { I[e] = 0.f; S[e] = 0.f; Y[e] = 0.f; }
    for (long v = blockIdx.x * 256L + threadIdx.x; v < V; v += (long)gridDim.x * 256) {
      const f16x8 pv8 = *(const f16x8*)((const f16*)p + ((long)n * V + v) * p_stride + c0);
      float yv8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) yv8[e] = y[((long)n * C + c0 + e) * V + v];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float pv = (float)pv8[e], yv = yv8[e];
        const float s = 1.f / (1.f + __expf(-pv));
        I[e] += s * yv; S[e] += s; Y[e] += yv;
        const float d = s - yv;
        mse = fmaf(d, d, mse);
        bce += fmaxf(pv, 0.f) - pv * yv + log1pf(__expf(-fabsf(pv)));
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = I[e], b = S[e], cc = Y[e];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); cc += __shfl_xor(cc, o); }
      if (lane == 0) { red[wave][3 * (c0 + e)] = a; red[wave][3 * (c0 + e) + 1] = b; red[wave][3 * (c0 + e) + 2] = cc; }
    }
  }
  for (int c = 0; !vec8 && c < C; ++c) {
    float I = 0.f, S = 0.f, Y = 0.f;
    for (long v = blockIdx.x * 256L + threadIdx.x; v < V; v += (long)gridDim.x * 256) {
      const float pv = (float)p[((long)n * V + v) * p_stride + c];
      const float yv = y[((long)n * C + c) * V + v];
      const float s = 1.f / (1.f + __expf(-pv));
      I += s * yv; S += s; Y += yv;
      const float d = s - yv;
      mse = fmaf(d, d, mse);
      bce += fmaxf(pv, 0.f) - pv * yv + log1pf(__expf(-fabsf(pv)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { I += __shfl_xor(I, o); S += __shfl_xor(S, o); Y += __shfl_xor(Y, o); }
    if (lane == 0) { red[wave][3 * c] = I; red[wave][3 * c + 1] = S; red[wave][3 * c + 2] = Y; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mse += __shfl_xor(mse, o); bce += __shfl_xor(bce, o); }
  if (lane == 0) { red[wave][3 * LOSS_MAXC] = mse; red[wave][3 * LOSS_MAXC + 1] = bce; }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * C; i += 256) {
    const double t = (double)red[0][i] + (double)red[1][i] + (double)red[2][i] + (double)red[3][i];
    unsafeAtomicAdd(sums + ((long)n * C + i / 3) * 4 + i % 3, t);
  }
  if (threadIdx.x < 2) {
    const int i = 3 * LOSS_MAXC + threadIdx.x;
    unsafeAtomicAdd(sums + (long)N * C * 4 + threadIdx.x,
                    (double)red[0][i] + (double)red[1][i] + (double)red[2][i] + (double)red[3][i]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void seg_loss_grad_kernel(const T* __restrict__ p, int p_stride,
                                                            const float* __restrict__ y, int C, long V,
                                                            const double* __restrict__ sums, int N,
                                                            const float* __restrict__ gscale, float w_mse, float w_bce,
                                                            float w_dice, T* __restrict__ dp, int dp_stride) {
  __shared__ float k0[LOSS_MAXC], k1[LOSS_MAXC];     // dice' = k0 * y + k1
  const int n = blockIdx.y;
  const float g = gscale ? *gscale : 1.f;
  const float invM = 1.f / ((float)N * (float)C * (float)V), invNC = w_dice / ((float)N * (float)C);
  const float a_mse = 2.f * w_mse * invM, a_bce = w_bce * invM;
  if (threadIdx.x < C) {
    const double* q = sums + ((long)n * C + threadIdx.x) * 4;
    const double De = q[1] + q[2] + 1e-5, Ie = 2.0 * q[0] + 1e-5;
    k0[threadIdx.x] = (float)(-2.0 / De) * invNC;
    k1[threadIdx.x] = (float)(Ie / (De * De)) * invNC;
  }
  __syncthreads();
  for (long v = blockIdx.x * 256L + threadIdx.x; v < V; v += (long)gridDim.x * 256) {
    const T* pr = p + ((long)n * V + v) * p_stride;
    T* dr = dp + ((long)n * V + v) * dp_stride;
    for (int c = 0; c < C; ++c) {
      const float pv = (float)pr[c];
      const float yv = y[((long)n * C + c) * V + v];
      const float s = 1.f / (1.f + __expf(-pv));
      const float ds = s * (1.f - s), d = s - yv;
      dr[c] = (T)(g * (a_mse * d * ds + a_bce * d + (k0[c] * yv + k1[c]) * ds));
    }
  }
}

}  // namespace dua

extern "C" {

int dua_seg_loss_reduce(int dtype, int N, int C, long voxels, const void* logits, int logits_stride, const float* labels,
                        double* sums, void* stream) {
  if (!logits || !labels || !sums || N <= 0 || C <= 0 || C > dua::LOSS_MAXC || voxels <= 0 || logits_stride < C) return DUA_ERR_ARG;
  long b = (voxels + 255) / 256;
  dim3 grid((unsigned)(b > 512 ? 512 : b), N);
  if (dtype == DUA_F16)
    hipLaunchKernelGGL(dua::seg_loss_reduce_kernel<dua::f16>, grid, dim3(256), 0, (hipStream_t)stream, (const dua::f16*)logits,
                       logits_stride, labels, C, voxels, sums, N);
  else if (dtype == DUA_F32)
    hipLaunchKernelGGL(dua::seg_loss_reduce_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)logits,
                       logits_stride, labels, C, voxels, sums, N);
  else return DUA_ERR_ARG;
  return (int)hipGetLastError();
}

int dua_seg_loss_grad(int dtype, int N, int C, long voxels, const void* logits, int logits_stride, const float* labels,
                      const double* sums, const float* gscale, float w_mse, float w_bce, float w_dice, void* dlogits,
                      int dlogits_stride, void* stream) {
  if (!logits || !labels || !sums || !dlogits || N <= 0 || C <= 0 || C > dua::LOSS_MAXC || voxels <= 0 ||
      logits_stride < C || dlogits_stride < C) return DUA_ERR_ARG;
  long b = (voxels + 255) / 256;
  dim3 grid((unsigned)(b > 4096 ? 4096 : b), N);
  if (dtype == DUA_F16)
    hipLaunchKernelGGL(dua::seg_loss_grad_kernel<dua::f16>, grid, dim3(256), 0, (hipStream_t)stream, (const dua::f16*)logits,
                       logits_stride, labels, C, voxels, sums, N, gscale, w_mse, w_bce, w_dice, (dua::f16*)dlogits, dlogits_stride);
  else if (dtype == DUA_F32)
    hipLaunchKernelGGL(dua::seg_loss_grad_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)logits,
                       logits_stride, labels, C, voxels, sums, N, gscale, w_mse, w_bce, w_dice, (float*)dlogits, dlogits_stride);
  else return DUA_ERR_ARG;
  return (int)hipGetLastError();
}

}  // extern "C"
