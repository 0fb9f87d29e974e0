// Token GEMMs of the Swin stages with tall activations and small weights (BASELINE config 5):
//   out[token][n] = sum_k A[token][k] * W[n][k] (+ bias[n])        M = 1e4 .. 1e6 tokens, K <= 384, N <= 192 per launch
// = the nn.Linear layers of WindowAttention (qkv, proj: models/swin_unetr/attention.py:91-94,99,118), of the MLP
// (MONAI MLPBlock linear1 / linear2, transformer.py:376,434) and of PatchMerging (reduction, patch.py:89-92) at the two
// fine stages (48^3 x 48 and 24^3 x 96 tokens), and the 1x1x1 conv3 of channel-changing UnetResBlocks (blocks.py:286-296).
// These are HBM-bound (a few FLOP per byte); what a library GEMM cannot do is fuse what follows, so each launch carries
// one of the epilogues the reference applies next:
//   PLAIN     + bias
//   GELU      + bias, exact GELU                                      (MLPBlock act, between linear1 and linear2)
//   STATS     raw output + per-(sample, channel) sum / sum of squares (norm3 statistics of conv3)
//   RESIDUAL  x[token] += out + bias on the fp32 stream               (x + mlp(norm2(x)), transformer.py:477-480)
//   SCATTER   window token -> voxel (window_reverse, roll back, crop), x += out + bias, ln_out = norm2(x)
//                                                                    (transformer.py:417-434, 475-476)
//
// Mapping: the weights are the A operand of MFMA 32x32x16 (rows = output channels, resident in LDS for the whole launch:
// persistent workgroups), a wave's 32 tokens are the B operand, read straight from global memory (one 16-byte k-group per
// lane and k-step; a token row is consumed completely by its two lanes).  The accumulator then holds, per LANE = TOKEN,
// four consecutive output channels per register quad: epilogues are per-token register arithmetic (LayerNorm = in-lane sum +
// one exchange with lane ^ 32) and 8 / 16-byte stores.
#include "common.hpp"
#include "../../include/dua_hip.h"
#include "swin_geom.hpp"

namespace dua {

namespace tg {
constexpr int MAXNB_ALL = 6;    // 32-row blocks of output channels per launch (N <= 192)
constexpr int KSTEPS = 12;      // k-steps (16 channels) resident per pass (192 channels)
}  // namespace tg

struct TokLinArgs {
  const f16* A; int lda; long M; int K, N;
  const f16* W; const float* bias;
  int mode;
  f16* out; int ldc, out_off;
  float* x;
  double* stats; int c_pad;
  WinGeom g; const float* gamma; const float* beta; float eps; f16* ln_out;
  int row_bytes;                // LDS row stride of W
};

__device__ __forceinline__ float gelu_erf(float a) { return 0.5f * a * (1.f + erff(a * 0.70710678118654752f)); }

template <int MODE, int NB>
__global__ __launch_bounds__(256) void token_linear_kernel(TokLinArgs a) {
  using namespace tg;
  constexpr int MAXNB = NB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int K = a.K, N = a.N, RS = a.row_bytes, kg = K >> 3;
  for (int i = tid; i < NB * 32 * kg; i += 256) {
    const int n = i / kg, g8 = i - n * kg;
    f16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (f16)0.f;
    if (n < N) v = *(const f16x8*)(a.W + (long)n * K + g8 * 8);
    *(f16x8*)(smem + n * RS + g8 * 16) = v;
  }
  __syncthreads();
  const int sample = blockIdx.y;                                  // STATS: one grid row per sample
  const f16* A = a.A + (long)sample * a.M * a.lda;
  float ssum[MODE == DUA_TOKLIN_STATS ? 2 : 1][16], ssq[MODE == DUA_TOKLIN_STATS ? 2 : 1][16];                                  // STATS (N <= 64): per-lane partial sums over this workgroup's tiles
  if (MODE == DUA_TOKLIN_STATS) {
#pragma unroll
    for (int nb = 0; nb < (NB < 2 ? NB : 2); ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) { ssum[nb][i] = 0.f; ssq[nb][i] = 0.f; }
  }
  const long tiles = (a.M + 127) / 128;
  for (long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const long tok = tile * 128 + wave * 32 + r;
    const bool valid = tok < a.M;
    const f16* arow = A + (valid ? tok : a.M - 1) * a.lda;
    f32x16 acc[MAXNB];
#pragma unroll
    for (int nb = 0; nb < MAXNB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16 * KSTEPS) {
      f16x8 bf[KSTEPS];
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const int kk = k0 + 16 * ks + 8 * hh;
#pragma unroll
        for (int e = 0; e < 8; ++e) bf[ks][e] = (f16)0.f;
        if (kk < K) bf[ks] = *(const f16x8*)(arow + kk);
      }
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        if (k0 + 16 * ks < K) {
#pragma unroll
          for (int nb = 0; nb < MAXNB; ++nb)
            if (nb < NB) {
              const f16x8 af = *(const f16x8*)(smem + (nb * 32 + r) * RS + (k0 + 16 * ks + 8 * hh) * 2);
              acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf[ks], acc[nb], 0, 0, 0);
            }
        }
      }
    }
    // ---- epilogues: register quad j of block nb = channels nb*32 + 8j + 4hh + (0..3) of this lane's token ----
    if (MODE == DUA_TOKLIN_PLAIN || MODE == DUA_TOKLIN_GELU || MODE == DUA_TOKLIN_STATS) {
      f16* orow = a.out + ((long)sample * a.M + tok) * a.ldc + a.out_off;
#pragma unroll
      for (int nb = 0; nb < MAXNB; ++nb)
        if (nb < NB) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int c0 = nb * 32 + 8 * j + 4 * hh;
            if (c0 < N) {
              f16x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float v = acc[nb][4 * j + e] + (a.bias ? a.bias[c0 + e] : 0.f);
                if (MODE == DUA_TOKLIN_GELU) v = gelu_erf(v);
                o[e] = (f16)v;
                if (MODE == DUA_TOKLIN_STATS && valid && nb < 2) {
                  const float q = (float)o[e];                   // statistics of what the consumer will read
                  ssum[nb][4 * j + e] += q; ssq[nb][4 * j + e] = fmaf(q, q, ssq[nb][4 * j + e]);
                }
              }
              if (valid) *(f16x4*)(orow + c0) = o;
            }
          }
        }
    } else if (MODE == DUA_TOKLIN_RESIDUAL) {
      if (valid) {
        float* xr = a.x + tok * N;
#pragma unroll
        for (int nb = 0; nb < MAXNB; ++nb)
          if (nb < NB) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int c0 = nb * 32 + 8 * j + 4 * hh;
              if (c0 < N) {
                f32x4 xv = *(f32x4*)(xr + c0);
#pragma unroll
                for (int e = 0; e < 4; ++e) xv[e] += acc[nb][4 * j + e] + (a.bias ? a.bias[c0 + e] : 0.f);
                *(f32x4*)(xr + c0) = xv;
              }
            }
          }
      }
    } else {                                                       // DUA_TOKLIN_SCATTER
      const WinGeom& g = a.g;
      const long tk = valid ? tok : 0;
      const int t = (int)(tk % g.n), wi = (int)((tk / g.n) % g.nw), b = (int)(tk / ((long)g.n * g.nw));
      int d, h, w;
      const bool real = window_to_voxel(g, wi, t, d, h, w) && valid;
      const long dst = ((((long)b * g.D + d) * g.H + h) * g.W + w) * N;
      float s = 0.f;
#pragma unroll
      for (int nb = 0; nb < MAXNB; ++nb)
        if (nb < NB) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int c0 = nb * 32 + 8 * j + 4 * hh;
            if (c0 < N) {
              f32x4 xv = {0.f, 0.f, 0.f, 0.f};
              if (real) xv = *(const f32x4*)(a.x + dst + c0);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float v = xv[e] + acc[nb][4 * j + e] + (a.bias ? a.bias[c0 + e] : 0.f);
                acc[nb][4 * j + e] = v;
                s += v;
              }
              if (real) *(f32x4*)(a.x + dst + c0) = f32x4{acc[nb][4 * j], acc[nb][4 * j + 1], acc[nb][4 * j + 2], acc[nb][4 * j + 3]};
            }
          }
        }
      s += __shfl_xor(s, 32);
      const float mean = s / (float)N;
      float q = 0.f;
#pragma unroll
      for (int nb = 0; nb < MAXNB; ++nb)
        if (nb < NB) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (nb * 32 + 8 * j + 4 * hh < N) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { const float dl = acc[nb][4 * j + e] - mean; q = fmaf(dl, dl, q); }
            }
        }
      q += __shfl_xor(q, 32);
      const float rstd = rsqrtf(q / (float)N + a.eps);
      if (real) {
#pragma unroll
        for (int nb = 0; nb < MAXNB; ++nb)
          if (nb < NB) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int c0 = nb * 32 + 8 * j + 4 * hh;
              if (c0 < N) {
                f16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (f16)((acc[nb][4 * j + e] - mean) * rstd * a.gamma[c0 + e] + a.beta[c0 + e]);
                *(f16x4*)(a.ln_out + dst + c0) = o;
              }
            }
          }
      }
    }
  }
  if (MODE == DUA_TOKLIN_STATS) {
#pragma unroll
    for (int nb = 0; nb < (NB < 2 ? NB : 2); ++nb)
      if (nb < NB) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float s = ssum[nb][i], ss = ssq[nb][i];
#pragma unroll
          for (int o2 = 16; o2 > 0; o2 >>= 1) { s += __shfl_xor(s, o2); ss += __shfl_xor(ss, o2); }
          const int c = nb * 32 + 8 * (i >> 2) + 4 * hh + (i & 3);
          if (r == 0 && c < N) {
            double* p = a.stats + (((long)sample * STAT_REPLICAS + (blockIdx.x % STAT_REPLICAS)) * a.c_pad + c) * 2;
            unsafeAtomicAdd(p, (double)s); unsafeAtomicAdd(p + 1, (double)ss);
          }
        }
      }
  }
}

}  // namespace dua

extern "C" int dua_token_linear(const dua_token_linear_desc* d, void* stream) {
  using namespace dua;
  if (!d || !d->A || !d->W || d->M <= 0 || d->K <= 0 || d->K % 8 || d->K > 384 || d->N <= 0 || d->N % 8 || d->N > 32 * tg::MAXNB_ALL ||
      d->lda < d->K || d->lda % 8 || d->samples <= 0)
    return DUA_ERR_ARG;
  TokLinArgs a{};
  a.A = (const f16*)d->A; a.lda = d->lda; a.M = d->M; a.K = d->K; a.N = d->N; a.W = (const f16*)d->W; a.bias = d->bias;
  a.mode = d->mode; a.out = (f16*)d->out; a.ldc = d->ldc; a.out_off = d->out_off; a.x = d->x; a.stats = d->stats;
  a.c_pad = d->c_pad; a.gamma = d->gamma; a.beta = d->beta; a.eps = d->eps; a.ln_out = (f16*)d->ln_out;
  int samples = 1;
  switch (d->mode) {
    case DUA_TOKLIN_PLAIN: case DUA_TOKLIN_GELU:
      if (!d->out || d->ldc % 4 || d->out_off % 4 || d->ldc < d->out_off + d->N || d->samples != 1) return DUA_ERR_ARG;
      break;
    case DUA_TOKLIN_STATS:
      if (!d->out || !d->stats || d->N > 64 || d->c_pad < d->N || d->ldc % 4 || d->out_off % 4 || d->ldc < d->out_off + d->N) return DUA_ERR_ARG;
      samples = d->samples;
      break;
    case DUA_TOKLIN_RESIDUAL:
      if (!d->x || d->samples != 1) return DUA_ERR_ARG;
      break;
    case DUA_TOKLIN_SCATTER:
      if (!d->x || !d->ln_out || !d->gamma || !d->beta || !geom_ok(&d->geom) || d->geom.C != d->N || d->samples != 1) return DUA_ERR_ARG;
      a.g = make_geom(&d->geom);
      if ((long)a.g.B * a.g.nw * a.g.n != d->M) return DUA_ERR_ARG;
      break;
    default: return DUA_ERR_ARG;
  }
  const int units = d->K * 2 / 16;
  a.row_bytes = d->K * 2 + ((units & 1) ? 0 : 16);
  const int NB = (d->N + 31) / 32;
  const int lds = NB * 32 * a.row_bytes + 16;
  using Kern = void (*)(TokLinArgs);
#define ROW(M_) {token_linear_kernel<M_, 1>, token_linear_kernel<M_, 2>, token_linear_kernel<M_, 3>, token_linear_kernel<M_, 4>, \
                 token_linear_kernel<M_, 5>, token_linear_kernel<M_, 6>}
  static const Kern table[5][6] = {ROW(DUA_TOKLIN_PLAIN), ROW(DUA_TOKLIN_GELU), ROW(DUA_TOKLIN_STATS), ROW(DUA_TOKLIN_RESIDUAL),
                                   ROW(DUA_TOKLIN_SCATTER)};
#undef ROW
  const Kern kern = table[d->mode][NB - 1];
  if (lds > 64 * 1024) {
    static bool raised[5][6] = {};
    static int attr_dev = -1;
    int dev = 0;
    hipGetDevice(&dev);
    if (attr_dev != dev) { for (auto& row : raised) for (bool& b : row) b = false; attr_dev = dev; }
    if (!raised[d->mode][NB - 1]) {
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
        return DUA_ERR_ARG;
      raised[d->mode][NB - 1] = true;
    }
  }
  // Enough workgroups to fill every CU to its occupancy limit: a wave works through load -> MFMA -> epilogue of one tile
  // at a time, so the latency of its loads is hidden by the OTHER waves of the SIMD, not inside the wave.
  static int occ[5][6] = {};
  static int occ_lds[5][6] = {};
  int& oc = occ[d->mode][NB - 1];
  if (oc == 0 || occ_lds[d->mode][NB - 1] != lds) {
    int nblk = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)kern, 256, lds) != hipSuccess || nblk < 1) nblk = 1;
    oc = nblk > 8 ? 8 : nblk;
    occ_lds[d->mode][NB - 1] = lds;
  }
  const long tiles = (d->M + 127) / 128;
  long cap = 256L * oc / samples;
  if (cap < 1) cap = 1;
  dim3 grid((unsigned)(tiles < cap ? tiles : cap), samples);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
