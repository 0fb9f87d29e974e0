// nn.Linear on fp32 token rows for the PARITY plan of the Swin path (BASELINE config 5):
//   out[m][n] = sum_k A[m][k] * W[n][k] + bias[n]   (optionally followed by the exact GELU of MONAI's MLPBlock)
// = qkv / proj of WindowAttention (models/swin_unetr/attention.py:97-120), linear1 / linear2 of the MLP
// (transformer.py:433-435), PatchMerging.reduction (patch.py:89-92) and the 1x1x1 conv3 of a channel-changing UnetResBlock
// (blocks.py:311-314) when the plan computes in fp32.  The fp16 plan has its own fused token kernels (swin_gemm.hip,
// swin_gemm_wide.hip); this one exists so that the fp32 plan launches no library GEMM either: exact-fp32 MFMA (32x32x2),
// 64 x 64 output tiles by four waves, operands staged through LDS in 32-wide K chunks.  It is the parity mode's kernel: plain,
// bounds-checked on every side (any M, N; K a multiple of 4), not tuned.
#include "common.hpp"
#include "../../include/dua_hip.h"

namespace dua {

constexpr int GF_BM = 64, GF_BN = 64, GF_BK = 32, GF_LD = GF_BK + 4;     // rows of 36 floats: 16-byte aligned fragments

__global__ __launch_bounds__(256) void linear_f32_kernel(long M, int K, int N, const float* __restrict__ A, long lda,
                                                         const float* __restrict__ W, const float* __restrict__ bias,
                                                         float* __restrict__ out, long ldc, int gelu) {
  __shared__ __attribute__((aligned(16))) float As[GF_BM][GF_LD];
  __shared__ __attribute__((aligned(16))) float Ws[GF_BN][GF_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  const long m0 = (long)blockIdx.x * GF_BM;
  const int n0 = blockIdx.y * GF_BN;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < K; k0 += GF_BK) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {                      // 64 rows x 8 pieces of 16 bytes per operand tile: two per thread
      const int idx = tid + u * 256, row = idx >> 3, k = k0 + (idx & 7) * 4;
      f32x4 av = {0.f, 0.f, 0.f, 0.f}, wv = {0.f, 0.f, 0.f, 0.f};
      if (m0 + row < M && k < K) av = *(const f32x4*)(A + (m0 + row) * lda + k);
      if (n0 + row < N && k < K) wv = *(const f32x4*)(W + (long)(n0 + row) * K + k);
      *(f32x4*)&As[row][(idx & 7) * 4] = av;
      *(f32x4*)&Ws[row][(idx & 7) * 4] = wv;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GF_BK; kk += 8) {
      const f32x4 a = *(const f32x4*)&As[wm * 32 + r][kk + 4 * hh];
      const f32x4 b = *(const f32x4*)&Ws[wn * 32 + r][kk + 4 * hh];
      mma32(acc, a, b);
    }
    __syncthreads();
  }
  const int n = n0 + wn * 32 + r;
  if (n >= N) return;
  const float bv = bias ? bias[n] : 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const long m = m0 + wm * 32 + acc_row(i, hh);
    if (m < M) {
      float v = acc[i] + bv;
      if (gelu) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
      out[m * ldc + n] = v;
    }
  }
}

}  // namespace dua

extern "C" int dua_linear_f32(long M, int K, int N, const float* A, long lda, const float* W, const float* bias, float* out,
                              long ldc, int gelu, void* stream) {
  if (M <= 0 || K <= 0 || K % 4 || N <= 0 || !A || !W || !out || lda < K || lda % 4 || ldc < N ||
      ((((size_t)A) | ((size_t)W)) & 15))
    return DUA_ERR_ARG;
  const long bm = (M + dua::GF_BM - 1) / dua::GF_BM;
  if (bm > 0x7fffffffL) return DUA_ERR_ARG;
  hipLaunchKernelGGL(dua::linear_f32_kernel, dim3((unsigned)bm, (N + dua::GF_BN - 1) / dua::GF_BN), dim3(256), 0,
                     (hipStream_t)stream, M, K, N, A, lda, W, bias, out, ldc, gelu);
  return (int)hipGetLastError();
}
