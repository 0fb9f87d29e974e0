// The 1x1x1 output convolution of the denoiser (final_conv, models/basic_unet/denoiser.py:282,311) for the TRAINING
// step -- forward on a materialised activation and the whole backward in one pass:
//   forward : logits[v][k] = b[k] + sum_c u[v][c] * W[k][c]
//   backward: du[v][c] = sum_k dlog[v][k] * W[k][c];  dW[k][c] += sum_v dlog[v][k] * u[v][c];  db[k] += sum_v dlog[v][k]
// (the sampling path has its own fused tail, sampler.hip).  K = classes <= 16, C <= 64: 2*K*C FLOP per voxel against
// (C + K) elements of traffic -- HBM bound, so plain VALU FMAs with the weights in registers; torch runs these as
// tall-skinny library GEMMs (K = 1.8 M voxels deep for dW) that took 1.6 + 0.2 ms of a 25 ms step.
#include "common.hpp"
#include "../../include/dua_hip.h"

namespace dua {

constexpr int HEAD_K = 16, HEAD_C = 64, HEAD_TV = 64;     // class / channel capacity, voxels per tile

// thread = (k = tid & 15, vq = tid >> 4): 16 voxels per sub-pass, W[k][:] in registers
template <typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ u, int u_stride, int C,
                                                       const float* __restrict__ W, const float* __restrict__ b, int K,
                                                       T* __restrict__ out, int out_stride, long total) {
  using Frag = typename Elem<T>::Frag;
  constexpr int EPG = Elem<T>::EPG;
  constexpr int RS = HEAD_C * (int)sizeof(T) + 16;          // padded tile row: the 4 voxels of a wave hit different banks
  __shared__ __attribute__((aligned(16))) char tile[HEAD_TV * RS];
  const int k = threadIdx.x & 15, vq = threadIdx.x >> 4;
  float w[HEAD_C];
#pragma unroll
  for (int c = 0; c < HEAD_C; ++c) w[c] = (k < K && c < C) ? W[k * C + c] : 0.f;
  const float bk = k < K ? b[k] : 0.f;
  const int gpc = C / EPG;
  for (long t0 = (long)blockIdx.x * HEAD_TV; t0 < total; t0 += (long)gridDim.x * HEAD_TV) {
    __syncthreads();
    for (int it = threadIdx.x; it < HEAD_TV * gpc; it += 256) {
      const int vl = it / gpc, g = it % gpc;
      Frag f;
      if (t0 + vl < total) f = *(const Frag*)(u + (t0 + vl) * u_stride + g * EPG);
      else
#pragma unroll
        for (int e = 0; e < EPG; ++e) f[e] = (T)0.f;
      *(Frag*)(tile + vl * RS + g * 16) = f;
    }
    __syncthreads();
#pragma unroll
    for (int sub = 0; sub < HEAD_TV / 16; ++sub) {
      const int vl = sub * 16 + vq;
      float acc = bk;
#pragma unroll
      for (int g = 0; g < HEAD_C / EPG; ++g) {
        if (g < gpc) {
          const Frag f = *(const Frag*)(tile + vl * RS + g * 16);
#pragma unroll
          for (int e = 0; e < EPG; ++e) acc = fmaf((float)f[e], w[g * EPG + e], acc);
        }
      }
      if (k < K && t0 + vl < total) out[(t0 + vl) * out_stride + k] = (T)acc;
    }
  }
}

// thread = (c = tid & 63, vq = tid >> 6): 4 voxels per sub-pass; W[:][c] and the dW[:][c] partials in registers
template <typename T>
__global__ __launch_bounds__(256) void head_bwd_kernel(const T* __restrict__ dlog, int dl_stride, int K,
                                                       const T* __restrict__ u, int u_stride, int C,
                                                       const float* __restrict__ W, T* __restrict__ du, int du_stride,
                                                       float* __restrict__ dW, float* __restrict__ db,
                                                       float* __restrict__ part, long total) {
  __shared__ float dl[HEAD_TV][HEAD_K];
  __shared__ float red[4][HEAD_K + 1][HEAD_C];
  const int c = threadIdx.x & 63, vq = threadIdx.x >> 6;
  float w[HEAD_K], dwp[HEAD_K], dbp = 0.f;
#pragma unroll
  for (int k = 0; k < HEAD_K; ++k) { w[k] = (k < K && c < C) ? W[k * C + c] : 0.f; dwp[k] = 0.f; }
  // this thread's share of a tile's dlogits (4 of the 64 x 16 values), loaded one tile ahead of its use
  constexpr int NDL = HEAD_TV * HEAD_K / 256;
  T pdl[NDL];
  auto load_dl = [&](long t0) {
#pragma unroll
    for (int j = 0; j < NDL; ++j) {
      const int it = threadIdx.x + 256 * j, vl = it / HEAD_K, k = it % HEAD_K;
      long v = t0 + vl;
      v = v < total ? v : total - 1;
      pdl[j] = dlog[v * dl_stride + (k < K ? k : K - 1)];
    }
  };
  const long tstep = (long)gridDim.x * HEAD_TV;
  long t0 = (long)blockIdx.x * HEAD_TV;
  if (t0 < total) load_dl(t0);
  for (; t0 < total; t0 += tstep) {
    // all 16 activations this thread needs from the tile are requested up front: one HBM latency per tile, not per voxel.
    // Branch-free (clamped addresses, select afterwards): a load under a lane-dependent branch sits in its own basic
    // block and hipcc waits for it there -- 16 serialised HBM round trips per tile
    T ur[HEAD_TV / 4];
    const int cl = c < C ? c : C - 1;
#pragma unroll
    for (int sub = 0; sub < HEAD_TV / 4; ++sub) {
      long v = t0 + sub * 4 + vq;
      v = v < total ? v : total - 1;
      ur[sub] = u[v * u_stride + cl];
    }
    __syncthreads();                      // the previous tile's dl reads are done
#pragma unroll
    for (int j = 0; j < NDL; ++j) {
      const int it = threadIdx.x + 256 * j, vl = it / HEAD_K, k = it % HEAD_K;
      dl[vl][k] = (k < K && t0 + vl < total) ? (float)pdl[j] : 0.f;
    }
    __syncthreads();
    if (t0 + tstep < total) load_dl(t0 + tstep);
    float uv[HEAD_TV / 4];
#pragma unroll
    for (int sub = 0; sub < HEAD_TV / 4; ++sub) uv[sub] = (c < C && t0 + sub * 4 + vq < total) ? (float)ur[sub] : 0.f;
#pragma unroll
    for (int sub = 0; sub < HEAD_TV / 4; ++sub) {
      const int vl = sub * 4 + vq;
      const long v = t0 + vl;
      float g = 0.f;
#pragma unroll
      for (int k = 0; k < HEAD_K; ++k) {
        const float d = dl[vl][k];
        g = fmaf(d, w[k], g);
        dwp[k] = fmaf(d, uv[sub], dwp[k]);
      }
      if (c < HEAD_K) dbp += dl[vl][c];
      if (c < C && v < total) du[v * du_stride + c] = (T)g;
    }
  }
#pragma unroll
  for (int k = 0; k < HEAD_K; ++k) red[vq][k][c] = dwp[k];
  red[vq][HEAD_K][c] = dbp;
  __syncthreads();
  for (int i = threadIdx.x; i < (HEAD_K + 1) * HEAD_C; i += 256) {
    const int k = i / HEAD_C, cc = i % HEAD_C;
    const float s = (red[0][k][cc] + red[1][k][cc]) + (red[2][k][cc] + red[3][k][cc]);
    if (part) part[(long)blockIdx.x * (HEAD_K + 1) * HEAD_C + i] = s;       // summed by head_reduce_kernel
    else if (k < K && cc < C) unsafeAtomicAdd(dW + k * C + cc, s);          // ~1000 blocks on ~1000 addresses: slow
    else if (k == HEAD_K && cc < K) unsafeAtomicAdd(db + cc, s);
  }
}

// fp16 forward on the matrix cores (C a multiple of 32, K <= 16): classes as MFMA rows, voxels as columns -- the operand scheme
// of the sampler's fused tail (sampler.hip): raw rows are the B operand straight from global memory, a lane ends up with four
// consecutive classes of one voxel and stores them as 8 bytes (16 voxels x 32 B contiguous per instruction).
template <int KS>
__global__ __launch_bounds__(256) void head_fwd_mfma_kernel(const f16* __restrict__ u, int u_stride, int C,
                                                            const float* __restrict__ W, const float* __restrict__ b, int K,
                                                            f16* __restrict__ out, int out_stride, long total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int vl = lane & 15, kq = lane >> 4;
  f16x8 aw[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e) aw[ks][e] = (f16)(vl < K ? W[vl * C + 32 * ks + 8 * kq + e] : 0.f);
  typedef float f32x4a __attribute__((ext_vector_type(4)));
  f32x4a bias;
#pragma unroll
  for (int j = 0; j < 4; ++j) bias[j] = 4 * kq + j < K ? b[4 * kq + j] : 0.f;
  const long ntile = (total + 255) / 256;
  for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const long wbase = (tile * 4 + wave) * 64;
    f16x8 fr[4][KS];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const long v = wbase + 16 * mb + vl;
      const long vc = v < total ? v : total - 1;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) fr[mb][ks] = *(const f16x8*)(u + vc * u_stride + 32 * ks + 8 * kq);
    }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      f32x4a acc = bias;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(aw[ks], fr[mb][ks], acc, 0, 0, 0);
      const long v = wbase + 16 * mb + vl;
      if (v >= total) continue;
      f16* o = out + v * out_stride + 4 * kq;
      if (4 * kq + 4 <= K && out_stride % 4 == 0) *(f16x4*)o = f16x4{(f16)acc[0], (f16)acc[1], (f16)acc[2], (f16)acc[3]};
      else
#pragma unroll
        for (int j = 0; j < 4; ++j) if (4 * kq + j < K) o[j] = (f16)acc[j];
    }
  }
}

// fp16, C = 64: the backward on the matrix cores.  Per wave and 32 voxels:
//   du^T[c][v] = W^T[c][k] . dlog^T[k][v]      MFMA 32x32x16 (k = the 16 classes), dlog rows straight from global as B
//   dW[k][c]  += dlog^T[k][v] . u[v][c]        MFMA 16x16x32 (k = 32 voxels): both operands want 8 consecutive VOXELS of one
//                                              class / channel per lane, i.e. the transposes of the channels-last tiles --
//                                              built once per tile in LDS (2-byte scatter writes, 16-byte operand reads)
//   db[k]     += dlog^T[k][v] . 1
// The VALU form above spends 32 FMAs and 16 LDS reads per (voxel, channel) and ran at 7x its HBM time.
__global__ __launch_bounds__(256) void head_bwd_mfma_kernel(const f16* __restrict__ dlog, int dl_stride, int K,
                                                            const f16* __restrict__ u, int u_stride,
                                                            const float* __restrict__ W, f16* __restrict__ du, int du_stride,
                                                            float* __restrict__ part, long total) {
  constexpr int C = HEAD_C;
  constexpr int UT = 32 * 2 + 16;                 // bytes per row of the transposed tiles (32 voxels + pad)
  constexpr int OS = C * 2 + 16;                  // bytes per row of the du staging tile
  __shared__ __attribute__((aligned(16))) char smem[4 * (C * UT + HEAD_K * UT + 32 * OS)];
  __shared__ float red[4][(HEAD_K + 1) * HEAD_C];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;         // 32x32x16 roles
  const int l16 = lane & 15, kq = lane >> 4;       // 16x16x32 roles
  char* uT = smem + wave * (C * UT + HEAD_K * UT + 32 * OS);       // [64 channels][32 voxels]
  char* dT = uT + C * UT;                                           // [16 classes][32 voxels]
  char* ot = dT + HEAD_K * UT;                                      // [32 voxels][64 channels] du staging
  f16x8 aw[2];                                     // W^T fragments: row = channel nb*32 + r, k = classes 8hh..8hh+7
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int e = 0; e < 8; ++e) aw[nb][e] = (f16)((8 * hh + e) < K ? W[(8 * hh + e) * C + nb * 32 + r] : 0.f);
  f16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (f16)1.f;
  typedef float f32x4a __attribute__((ext_vector_type(4)));
  f32x4a dw[4], dbv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int b = 0; b < 4; ++b) dw[b] = f32x4a{0.f, 0.f, 0.f, 0.f};
  const long ntile = (total + 127) / 128;          // 128 voxels per workgroup tile, 32 per wave
  for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const long v0 = tile * 128 + wave * 32;
    const long v = v0 + r;
    const bool ok = v < total;
    const long vc = ok ? v : (total - 1);
    // dlog row half of this lane's voxel (B operand of the du product) and the wave's u tile (4 x 16 B per lane)
    f16x8 dl;
#pragma unroll
    for (int e = 0; e < 8; ++e) dl[e] = (f16)0.f;
    if (ok) {
      if (K == HEAD_K) dl = *(const f16x8*)(dlog + vc * dl_stride + 8 * hh);
      else
#pragma unroll
        for (int e = 0; e < 8; ++e) if (8 * hh + e < K) dl[e] = dlog[vc * dl_stride + 8 * hh + e];
    }
    f16x8 uf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {                   // chunk j*64 + lane of the 32 x 8 chunks: voxel = chunk / 8, group = chunk % 8
      const int ch = j * 64 + lane, vl = ch >> 3, g = ch & 7;
      const long vv = v0 + vl;
#pragma unroll
      for (int e = 0; e < 8; ++e) uf[j][e] = (f16)0.f;
      if (vv < total) uf[j] = *(const f16x8*)(u + vv * u_stride + g * 8);
    }
    // transposes into LDS
#pragma unroll
    for (int e = 0; e < 8; ++e) *(f16*)(dT + (8 * hh + e) * UT + r * 2) = dl[e];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ch = j * 64 + lane, vl = ch >> 3, g = ch & 7;
#pragma unroll
      for (int e = 0; e < 8; ++e) *(f16*)(uT + (g * 8 + e) * UT + vl * 2) = uf[j][e];
    }
    __builtin_amdgcn_wave_barrier();
    // du^T = W^T . dlog^T
    f32x16 acc[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
      acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw[nb], dl, acc[nb], 0, 0, 0);
    }
    // dW += dlog^T . u, db += dlog^T . 1   (k = the wave's 32 voxels)
    const f16x8 ad = *(const f16x8*)(dT + l16 * UT + kq * 16);             // A: row = class l16, voxels 8kq..8kq+7
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const f16x8 bu = *(const f16x8*)(uT + (b * 16 + l16) * UT + kq * 16);   // B: col = channel b*16 + l16
      dw[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ad, bu, dw[b], 0, 0, 0);
    }
    dbv = __builtin_amdgcn_mfma_f32_16x16x32_f16(ad, ones, dbv, 0, 0, 0);
    // du: accumulator (lane = voxel r, quads of 4 channels) -> staging rows -> 128-byte rows to global
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c0 = nb * 32 + 8 * j + 4 * hh;
        *(f16x4*)(ot + r * OS + c0 * 2) = f16x4{(f16)acc[nb][4 * j], (f16)acc[nb][4 * j + 1], (f16)acc[nb][4 * j + 2], (f16)acc[nb][4 * j + 3]};
      }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ch = j * 64 + lane, vl = ch >> 3, g = ch & 7;
      if (v0 + vl < total) *(f16x8*)(du + (v0 + vl) * du_stride + g * 8) = *(const f16x8*)(ot + vl * OS + g * 16);
    }
    __builtin_amdgcn_wave_barrier();
  }
  // accumulators -> per-wave partials -> per-workgroup partials (summed by head_reduce_kernel)
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[wave][(4 * kq + j) * HEAD_C + b * 16 + l16] = dw[b][j];
  if (l16 == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) red[wave][HEAD_K * HEAD_C + 4 * kq + j] = dbv[j];
  }
  __syncthreads();
  for (int i = tid; i < (HEAD_K + 1) * HEAD_C; i += 256) {
    const bool live = i < HEAD_K * HEAD_C || i < HEAD_K * HEAD_C + HEAD_K;
    const float sm_ = live ? (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]) : 0.f;
    part[(long)blockIdx.x * (HEAD_K + 1) * HEAD_C + i] = sm_;
  }
}

// dW[k][c] += sum over blocks of part[block][k][c]; db[k] += sum of part[block][HEAD_K][k]
__global__ __launch_bounds__(256) void head_reduce_kernel(const float* __restrict__ part, int nblocks, int K, int C,
                                                          float* __restrict__ dW, float* __restrict__ db) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= (HEAD_K + 1) * HEAD_C) return;
  const int k = i / HEAD_C, cc = i % HEAD_C;
  const bool is_w = k < K && cc < C, is_b = k == HEAD_K && cc < K;
  if (!is_w && !is_b) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const int per = (nblocks + gridDim.y - 1) / gridDim.y;      // blockIdx.y takes one slice of the partials
  int b = blockIdx.y * per;
  nblocks = min(nblocks, b + per);
  for (; b + 3 < nblocks; b += 4) {
    s0 += part[(long)b * (HEAD_K + 1) * HEAD_C + i]; s1 += part[(long)(b + 1) * (HEAD_K + 1) * HEAD_C + i];
    s2 += part[(long)(b + 2) * (HEAD_K + 1) * HEAD_C + i]; s3 += part[(long)(b + 3) * (HEAD_K + 1) * HEAD_C + i];
  }
  for (; b < nblocks; ++b) s0 += part[(long)b * (HEAD_K + 1) * HEAD_C + i];
  const float s = (s0 + s1) + (s2 + s3);
  if (is_w) unsafeAtomicAdd(dW + k * C + cc, s); else unsafeAtomicAdd(db + cc, s);     // gridDim.y adds per address
}

static inline unsigned head_blocks(long total) {
  long b = (total + HEAD_TV - 1) / HEAD_TV;
  return (unsigned)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}

}  // namespace dua

extern "C" {

int dua_head_fwd(int dtype, long voxels, int C, int K, const void* u, int u_stride, const float* W, const float* b,
                 void* logits, int logits_stride, void* stream) {
  if (!u || !W || !b || !logits || voxels <= 0 || C <= 0 || C > dua::HEAD_C || C % 8 || K <= 0 || K > dua::HEAD_K ||
      u_stride % 8 || u_stride < C || logits_stride < K) return DUA_ERR_ARG;
  dim3 grid(dua::head_blocks(voxels));
  if (dtype == DUA_F16 && (C == 32 || C == 64) && u_stride % 8 == 0) {
    const long t256 = (voxels + 255) / 256;
    dim3 g2((unsigned)(t256 < 1024 ? t256 : 1024));
    if (C == 32) hipLaunchKernelGGL(dua::head_fwd_mfma_kernel<1>, g2, dim3(256), 0, (hipStream_t)stream, (const dua::f16*)u, u_stride, C, W, b, K, (dua::f16*)logits, logits_stride, voxels);
    else hipLaunchKernelGGL(dua::head_fwd_mfma_kernel<2>, g2, dim3(256), 0, (hipStream_t)stream, (const dua::f16*)u, u_stride, C, W, b, K, (dua::f16*)logits, logits_stride, voxels);
    return (int)hipGetLastError();
  }
  if (dtype == DUA_F16)
    hipLaunchKernelGGL(dua::head_fwd_kernel<dua::f16>, grid, dim3(256), 0, (hipStream_t)stream, (const dua::f16*)u, u_stride,
                       C, W, b, K, (dua::f16*)logits, logits_stride, voxels);
  else if (dtype == DUA_F32)
    hipLaunchKernelGGL(dua::head_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)u, u_stride, C, W,
                       b, K, (float*)logits, logits_stride, voxels);
  else return DUA_ERR_ARG;
  return (int)hipGetLastError();
}

long dua_head_bwd_workspace(long voxels) {
  if (voxels <= 0) return DUA_ERR_ARG;
  const long tiles = (voxels + dua::HEAD_TV - 1) / dua::HEAD_TV;
  return (tiles > 1024 ? 1024 : tiles) * (dua::HEAD_K + 1) * dua::HEAD_C * 4;
}

int dua_head_bwd(int dtype, long voxels, int C, int K, const void* dlogits, int dlogits_stride, const void* u,
                 int u_stride, const float* W, void* du, int du_stride, float* dW, float* db, void* workspace,
                 long workspace_bytes, void* stream) {
  if (!dlogits || !u || !W || !du || !dW || !db || voxels <= 0 || C <= 0 || C > dua::HEAD_C || K <= 0 ||
      K > dua::HEAD_K || dlogits_stride < K || u_stride < C || du_stride < C) return DUA_ERR_ARG;
  const long tiles = (voxels + dua::HEAD_TV - 1) / dua::HEAD_TV;
  dim3 grid((unsigned)(tiles > 1024 ? 1024 : tiles));
  float* part = (workspace && workspace_bytes >= dua_head_bwd_workspace(voxels)) ? (float*)workspace : nullptr;
  if (dtype == DUA_F16 && C == dua::HEAD_C && part && u_stride % 8 == 0 && du_stride % 8 == 0 && (K < dua::HEAD_K || dlogits_stride % 8 == 0)) {
    // matrix-core form; its grid must not exceed what the workspace was sized for
    const long t128 = (voxels + 127) / 128;
    dim3 g2((unsigned)(t128 < (long)grid.x ? t128 : (long)grid.x));
    hipLaunchKernelGGL(dua::head_bwd_mfma_kernel, g2, dim3(256), 0, (hipStream_t)stream, (const dua::f16*)dlogits, dlogits_stride, K,
                       (const dua::f16*)u, u_stride, W, (dua::f16*)du, du_stride, part, voxels);
    hipLaunchKernelGGL(dua::head_reduce_kernel, dim3(((dua::HEAD_K + 1) * dua::HEAD_C + 255) / 256, 16), dim3(256), 0,
                       (hipStream_t)stream, part, (int)g2.x, K, C, dW, db);
    return (int)hipGetLastError();
  }
  if (dtype == DUA_F16)
    hipLaunchKernelGGL(dua::head_bwd_kernel<dua::f16>, grid, dim3(256), 0, (hipStream_t)stream, (const dua::f16*)dlogits,
                       dlogits_stride, K, (const dua::f16*)u, u_stride, C, W, (dua::f16*)du, du_stride, dW, db, part, voxels);
  else if (dtype == DUA_F32)
    hipLaunchKernelGGL(dua::head_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dlogits,
                       dlogits_stride, K, (const float*)u, u_stride, C, W, (float*)du, du_stride, dW, db, part, voxels);
  else return DUA_ERR_ARG;
  if (part)
    hipLaunchKernelGGL(dua::head_reduce_kernel, dim3(((dua::HEAD_K + 1) * dua::HEAD_C + 255) / 256, 16), dim3(256), 0,
                       (hipStream_t)stream, part, (int)grid.x, K, C, dW, db);
  return (int)hipGetLastError();
}

}  // extern "C"
