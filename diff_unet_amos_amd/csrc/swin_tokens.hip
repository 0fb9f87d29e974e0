// Token-stream kernels of the Swin transformer inside the diff_swin_unetr variant (BASELINE config 5).
//
// The residual token stream x of a stage lives in HBM as fp32 [B][D][H][W][C] (the reference's "b d h w c" layout,
// models/swin_unetr/transformer.py:93-109); everything that feeds a GEMM or a convolution is written in the compute
// dtype T.  Every kernel here is one streaming pass (HBM bound), a group of G lanes owning one token so that LayerNorm
// is a register reduction:
//
//  window_gather_norm       SwinTransformerBlock.forward_part1 up to the attention (transformer.py:378-417):
//                           norm1 -> zero-pad to a window multiple -> roll(-shift) -> window_partition, optionally after
//                           folding the previous block's MLP output into the stream (x += y, transformer.py:477-480).
//  window_scatter_add_norm  the way back (transformer.py:417-431, 475-476): window_reverse -> roll(+shift) -> crop ->
//                           x = shortcut + attn, and norm2(x) for the MLP (transformer.py:433-434).
//  stage_out                the adds between stages (transformer.py:277-312): x = y + t_proj(swish(t)) per sample,
//                           out = layer_norm(x) without affine (proj_out, :253-268) + the encoder's feature map
//                           (swin_unetr/denoiser.py:367-368), written as a channels-last slice for the convolutions.
//  patch_embed              PatchEmbed's Conv3d(k = s = 2) + bias (+ t_proj, + proj_out) for stage 0.
//  instnorm_stats           per-(n, c) sum / sum of squares of a channels-last tensor (statistics of the 1x1x1 conv3 of
//                           UnetResBlock, blocks.py:286-296, whose GEMM is a library call).
//  gelu                     exact (erf) GELU in place between the two MLP GEMMs (MONAI MLPBlock, act "GELU").
#include "common.hpp"
#include "../../include/dua_hip.h"
#include "swin_geom.hpp"

namespace dua {

// fp16 PatchEmbed on the matrix cores (swin_gemm.hip)
int launch_patch_embed_mfma(int B, int D, int H, int W, int Cs, int Cp, const void* in, const float* wk, const float* bias,
                            const float* tadd, int tadd_stride, float eps, const void* emb, float* x, void* out, int out_stride,
                            int out_off, hipStream_t stream);

template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// LayerNorm of one token spread over G lanes, CPL channels per lane (channel = i * G + lane-in-group).
template <int G, int CPL>
__device__ __forceinline__ void group_layernorm(float (&v)[CPL], int C, float eps, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) s += v[i];
  mean = group_sum<G>(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) { const float d = v[i] - mean; q = fmaf(d, d, q); }
  rstd = rsqrtf(group_sum<G>(q) / (float)C + eps);
}

// 16-byte accesses: a lane owns NP pieces of four consecutive channels (piece = i * G + lane-in-group), G = C / (4 NP) lanes per
// token.  (One channel per lane and request -- 12 bytes per lane in flight -- ran at 1.6 TB/s at 48^3 x 48: 20 us against 8.5.)
template <typename T, int G, int NP>
__global__ __launch_bounds__(256) void window_gather_norm_vec_kernel(WinGeom g, float* __restrict__ x, const T* __restrict__ y,
                                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                     float eps, T* __restrict__ out) {
  typedef T TV4 __attribute__((ext_vector_type(4)));
  const int j = threadIdx.x % G;
  const long tok = blockIdx.x * (long)(256 / G) + threadIdx.x / G;
  const long total = (long)g.B * g.nw * g.n;
  if (tok >= total) return;
  const int t = (int)(tok % g.n), wi = (int)((tok / g.n) % g.nw), b = (int)(tok / ((long)g.n * g.nw));
  int d, h, w;
  const bool real = window_to_voxel(g, wi, t, d, h, w);
  T* o = out + tok * g.C;
  if (!real) {                                              // F.pad after norm1: padded tokens are zeros
#pragma unroll
    for (int i = 0; i < NP; ++i) *(TV4*)(o + (i * G + j) * 4) = TV4{(T)0.f, (T)0.f, (T)0.f, (T)0.f};
    return;
  }
  const long src = ((((long)b * g.D + d) * g.H + h) * g.W + w) * g.C;
  float v[4 * NP];
  f32x4 gm[NP], bt[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const f32x4 xv = *(const f32x4*)(x + src + (i * G + j) * 4);
    gm[i] = *(const f32x4*)(gamma + (i * G + j) * 4);
    bt[i] = *(const f32x4*)(beta + (i * G + j) * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[4 * i + e] = xv[e];
  }
  if (y) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const TV4 yv = *(const TV4*)(y + src + (i * G + j) * 4);
      f32x4 xv;
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[4 * i + e] += (float)yv[e]; xv[e] = v[4 * i + e]; }
      *(f32x4*)(x + src + (i * G + j) * 4) = xv;
    }
  }
  float mean, rstd;
  group_layernorm<G, 4 * NP>(v, g.C, eps, mean, rstd);
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    TV4 ov;
#pragma unroll
    for (int e = 0; e < 4; ++e) ov[e] = (T)((v[4 * i + e] - mean) * rstd * gm[i][e] + bt[i][e]);
    *(TV4*)(o + (i * G + j) * 4) = ov;
  }
}

template <typename T, int G, int NP>
__global__ __launch_bounds__(256) void window_scatter_add_norm_kernel(WinGeom g, float* __restrict__ x, const T* __restrict__ yw,
                                                                      const float* __restrict__ gamma,
                                                                      const float* __restrict__ beta, float eps,
                                                                      T* __restrict__ out) {
  typedef T TV4 __attribute__((ext_vector_type(4)));
  const int j = threadIdx.x % G;
  const long tok = blockIdx.x * (long)(256 / G) + threadIdx.x / G;
  const long total = (long)g.B * g.D * g.H * g.W;
  if (tok >= total) return;
  const int w = (int)(tok % g.W), h = (int)((tok / g.W) % g.H), d = (int)((tok / ((long)g.W * g.H)) % g.D);
  const int b = (int)(tok / ((long)g.W * g.H * g.D));
  const int ds = (d - g.sd + g.Dp) % g.Dp, hs = (h - g.sh + g.Hp) % g.Hp, ws = (w - g.sw + g.Wp) % g.Wp;
  const int wi = ((ds / g.wd) * g.nwh + hs / g.wh) * g.nww + ws / g.ww;
  const int t = ((ds % g.wd) * g.wh + hs % g.wh) * g.ww + ws % g.ww;
  const long src = (((long)b * g.nw + wi) * g.n + t) * g.C, dst = tok * g.C;
  float v[4 * NP];
  f32x4 gm[NP], bt[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int c = (i * G + j) * 4;
    f32x4 xv = *(const f32x4*)(x + dst + c);
    const TV4 yv = *(const TV4*)(yw + src + c);
    gm[i] = *(const f32x4*)(gamma + c);
    bt[i] = *(const f32x4*)(beta + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) { xv[e] += (float)yv[e]; v[4 * i + e] = xv[e]; }
    *(f32x4*)(x + dst + c) = xv;
  }
  float mean, rstd;
  group_layernorm<G, 4 * NP>(v, g.C, eps, mean, rstd);
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    TV4 ov;
#pragma unroll
    for (int e = 0; e < 4; ++e) ov[e] = (T)((v[4 * i + e] - mean) * rstd * gm[i][e] + bt[i][e]);
    *(TV4*)(out + dst + (i * G + j) * 4) = ov;
  }
}

template <typename T, int G, int NP>
__global__ __launch_bounds__(256) void stage_out_kernel(long tokens, long per_sample, int C, const T* __restrict__ y,
                                                        const float* __restrict__ tadd, int tadd_stride, float eps,
                                                        const T* __restrict__ emb, float* __restrict__ x, T* __restrict__ out,
                                                        int out_stride, int out_off) {
  typedef T TV4 __attribute__((ext_vector_type(4)));
  const int j = threadIdx.x % G;
  const long tok = blockIdx.x * (long)(256 / G) + threadIdx.x / G;
  if (tok >= tokens) return;
  const long b = tok / per_sample;
  float v[4 * NP];
  TV4 ev[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int c = (i * G + j) * 4;
    const TV4 yv = *(const TV4*)(y + tok * C + c);
    f32x4 ta = {0.f, 0.f, 0.f, 0.f};
    if (tadd) ta = *(const f32x4*)(tadd + b * tadd_stride + c);
    if (emb) ev[i] = *(const TV4*)(emb + tok * C + c);
    f32x4 xv;
#pragma unroll
    for (int e = 0; e < 4; ++e) { xv[e] = (float)yv[e] + ta[e]; v[4 * i + e] = xv[e]; }
    if (x) *(f32x4*)(x + tok * C + c) = xv;
  }
  float mean, rstd;
  group_layernorm<G, 4 * NP>(v, C, eps, mean, rstd);
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    TV4 ov;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float r = (v[4 * i + e] - mean) * rstd;
      if (emb) r += (float)ev[i][e];
      ov[e] = (T)r;
    }
    *(TV4*)(out + tok * out_stride + out_off + (i * G + j) * 4) = ov;
  }
}

// One thread per output token: E accumulators, the 2x2x2 x Cp inputs streamed through registers, weights broadcast from LDS.
template <typename T, int E>
__global__ __launch_bounds__(256) void patch_embed_kernel(int B, int D, int H, int W, int Cs, int Cp, const T* __restrict__ in,
                                                          const float* __restrict__ wk, const float* __restrict__ bias,
                                                          const float* __restrict__ tadd, int tadd_stride, float eps,
                                                          const T* __restrict__ emb, float* __restrict__ x,
                                                          T* __restrict__ out, int out_stride, int out_off) {
  extern __shared__ float wl[];                     // [8 * Cp][E]
  const int K = 8 * Cp;
  for (int i = threadIdx.x; i < K * E; i += 256) wl[i] = wk[i];
  __syncthreads();
  const int D2 = D / 2, H2 = H / 2, W2 = W / 2;
  const long total = (long)B * D2 * H2 * W2;
  const long tok = blockIdx.x * 256L + threadIdx.x;
  if (tok >= total) return;
  const int w2 = (int)(tok % W2), h2 = (int)((tok / W2) % H2), d2 = (int)((tok / ((long)W2 * H2)) % D2);
  const int b = (int)(tok / ((long)W2 * H2 * D2));
  float acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) acc[e] = bias[e] + (tadd ? tadd[(long)b * tadd_stride + e] : 0.f);
  constexpr int EPG = Elem<T>::EPG;
  using Frag = typename Elem<T>::Frag;
  for (int tap = 0; tap < 8; ++tap) {
    const int d = 2 * d2 + (tap >> 2), h = 2 * h2 + ((tap >> 1) & 1), w = 2 * w2 + (tap & 1);
    const T* p = in + ((((long)b * D + d) * H + h) * W + w) * Cs;
    for (int c0 = 0; c0 < Cp; c0 += EPG) {
      const Frag f = *(const Frag*)(p + c0);
#pragma unroll
      for (int q = 0; q < EPG; ++q) {
        const float a = (float)f[q];
        const float* wr = wl + (tap * Cp + c0 + q) * E;
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = fmaf(a, wr[e], acc[e]);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) s += acc[e];
  const float mean = s / (float)E;
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) { const float dlt = acc[e] - mean; q = fmaf(dlt, dlt, q); }
  const float rstd = rsqrtf(q / (float)E + eps);
  if (x) {
#pragma unroll
    for (int e = 0; e < E; e += 4) *(f32x4*)(x + tok * E + e) = f32x4{acc[e], acc[e + 1], acc[e + 2], acc[e + 3]};
  }
  T* o = out + tok * out_stride + out_off;
#pragma unroll
  for (int e0 = 0; e0 < E; e0 += EPG) {
    Frag f, m;
    if (emb) m = *(const Frag*)(emb + tok * E + e0);
#pragma unroll
    for (int q2 = 0; q2 < EPG; ++q2) {
      float r = (acc[e0 + q2] - mean) * rstd;
      if (emb) r += (float)m[q2];
      f[q2] = (T)r;
    }
    *(Frag*)(o + e0) = f;
  }
}

// Sum and sum of squares per (n, c) of a channels-last tensor: thread = (voxel lane, k-group); block partials are reduced
// in LDS and land with one pair of atomics per (block, channel) in replica blockIdx.x % 8 of the statistics buffer.
template <typename T>
__global__ __launch_bounds__(256) void instnorm_stats_kernel(const T* __restrict__ x, int x_stride, int x_off, long vox, int C,
                                                             stat_t* __restrict__ stats, int c_pad) {
  constexpr int EPG = Elem<T>::EPG;
  using Frag = typename Elem<T>::Frag;
  // fp32 (parity) tensors: the thread's partial sums in double, like the convolution epilogues (a channel whose mean is many
  // standard deviations loses digits of its variance to fp32 partial sums of x^2: a raw tensor of 100 +- 0.3 normalised 2e-3 ..
  // 8e-3 off before, 5e-5 -- the consumer's own fp32 rounding -- after)
  using part_t = typename std::conditional<sizeof(T) == 4, double, float>::type;
  extern __shared__ __attribute__((aligned(8))) char red_raw[];
  part_t* red = (part_t*)red_raw;                   // [vl][C][2]
  const int groups = C / EPG, vlanes = 256 / groups;
  const int n = blockIdx.y, g = threadIdx.x % groups, vl = threadIdx.x / groups;
  part_t s[EPG], q[EPG];
#pragma unroll
  for (int e = 0; e < EPG; ++e) { s[e] = 0; q[e] = 0; }
  if (vl < vlanes) {
    for (long v = blockIdx.x * (long)vlanes + vl; v < vox; v += (long)gridDim.x * vlanes) {
      const Frag f = *(const Frag*)(x + ((long)n * vox + v) * x_stride + x_off + g * EPG);
#pragma unroll
      for (int e = 0; e < EPG; ++e) {
        const float a = (float)f[e];
        s[e] += a;
        if constexpr (sizeof(T) == 4) q[e] += (double)a * (double)a; else q[e] = fmaf(a, a, q[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < EPG; ++e) {
      red[((long)vl * C + g * EPG + e) * 2] = s[e];
      red[((long)vl * C + g * EPG + e) * 2 + 1] = q[e];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    double S = 0, Q = 0;
    for (int l = 0; l < vlanes; ++l) { S += red[((long)l * C + c) * 2]; Q += red[((long)l * C + c) * 2 + 1]; }
    stats_add(stats, n, c_pad, blockIdx.x % STAT_REPLICAS, c, S, Q);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gelu_kernel(T* __restrict__ x, long groups) {
  constexpr int EPG = Elem<T>::EPG;
  using Frag = typename Elem<T>::Frag;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < groups; i += (long)gridDim.x * 256) {
    Frag f = *(Frag*)(x + i * EPG);
#pragma unroll
    for (int e = 0; e < EPG; ++e) {
      const float a = (float)f[e];
      f[e] = (T)gelu_erf(a);
    }
    *(Frag*)(x + i * EPG) = f;
  }
}

}  // namespace dua

// C -> lanes per token / channels per lane: 48 -> 16 x 3, 96 -> 32 x 3, 192 -> 64 x 3, 384 -> 64 x 6, 768 -> 64 x 12
// lanes per token x pieces of four channels per lane (C = 4 G NP)
#define DUA_TOKEN_DISPATCH(C_, T_, CALL)                 \
  switch (C_) {                                          \
    case 48:  { CALL(T_, 4, 3); break; }                 \
    case 96:  { CALL(T_, 8, 3); break; }                 \
    case 192: { CALL(T_, 16, 3); break; }                \
    case 384: { CALL(T_, 32, 3); break; }                \
    case 768: { CALL(T_, 64, 3); break; }                \
    default: return DUA_ERR_ARG;                         \
  }

extern "C" {

int dua_window_gather_norm(int dtype, const dua_window_geom* geom, float* x, const void* y, const float* gamma,
                           const float* beta, float eps, void* out, void* stream) {
  using namespace dua;
  if (!geom_ok(geom) || !x || !gamma || !beta || !out) return DUA_ERR_ARG;
  const WinGeom g = make_geom(geom);
  const long total = (long)g.B * g.nw * g.n;
#define CALL(T_, G_, NP_)                                                                                                   \
  hipLaunchKernelGGL((window_gather_norm_vec_kernel<T_, G_, NP_>), dim3((unsigned)((total + 256 / G_ - 1) / (256 / G_))),    \
                     dim3(256), 0, (hipStream_t)stream, g, x, (const T_*)y, gamma, beta, eps, (T_*)out)
  if (dtype == DUA_F16) { DUA_TOKEN_DISPATCH(g.C, f16, CALL) }
  else if (dtype == DUA_F32) { DUA_TOKEN_DISPATCH(g.C, float, CALL) }
  else return DUA_ERR_ARG;
#undef CALL
  return (int)hipGetLastError();
}

int dua_window_scatter_add_norm(int dtype, const dua_window_geom* geom, float* x, const void* yw, const float* gamma,
                                const float* beta, float eps, void* out, void* stream) {
  using namespace dua;
  if (!geom_ok(geom) || !x || !yw || !gamma || !beta || !out) return DUA_ERR_ARG;
  const WinGeom g = make_geom(geom);
  const long total = (long)g.B * g.D * g.H * g.W;
#define CALL(T_, G_, CPL_)                                                                                                  \
  hipLaunchKernelGGL((window_scatter_add_norm_kernel<T_, G_, CPL_>), dim3((unsigned)((total + 256 / G_ - 1) / (256 / G_))),  \
                     dim3(256), 0, (hipStream_t)stream, g, x, (const T_*)yw, gamma, beta, eps, (T_*)out)
  if (dtype == DUA_F16) { DUA_TOKEN_DISPATCH(g.C, f16, CALL) }
  else if (dtype == DUA_F32) { DUA_TOKEN_DISPATCH(g.C, float, CALL) }
  else return DUA_ERR_ARG;
#undef CALL
  return (int)hipGetLastError();
}

int dua_stage_out(int dtype, int B, long tokens_per_sample, int C, const void* y, const float* tadd, int tadd_stride,
                  float eps, const void* emb, float* x, void* out, int out_stride, int out_off, void* stream) {
  using namespace dua;
  if (B <= 0 || tokens_per_sample <= 0 || !y || !out || out_stride < out_off + C || (tadd && tadd_stride < C)) return DUA_ERR_ARG;
  // 16-byte accesses: four-channel pieces must be aligned in every operand
  if (out_stride % 4 || out_off % 4 || (tadd && (tadd_stride % 4 || ((size_t)tadd & 15))) || ((size_t)y & 7) || ((size_t)out & 7) ||
      (x && ((size_t)x & 15)) || (emb && ((size_t)emb & 7)))
    return DUA_ERR_ARG;
  const long total = (long)B * tokens_per_sample;
#define CALL(T_, G_, CPL_)                                                                                                 \
  hipLaunchKernelGGL((stage_out_kernel<T_, G_, CPL_>), dim3((unsigned)((total + 256 / G_ - 1) / (256 / G_))), dim3(256), 0, \
                     (hipStream_t)stream, total, tokens_per_sample, C, (const T_*)y, tadd, tadd_stride, eps,                \
                     (const T_*)emb, x, (T_*)out, out_stride, out_off)
  if (dtype == DUA_F16) { DUA_TOKEN_DISPATCH(C, f16, CALL) }
  else if (dtype == DUA_F32) { DUA_TOKEN_DISPATCH(C, float, CALL) }
  else return DUA_ERR_ARG;
#undef CALL
  return (int)hipGetLastError();
}

int dua_patch_embed(int dtype, int B, int D, int H, int W, int Cin_stride, int Cin_packed, int E, const void* in,
                    const float* w_packed, const float* bias, const float* tadd, int tadd_stride, float eps, const void* emb,
                    float* x, void* out, int out_stride, int out_off, void* stream) {
  using namespace dua;
  if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || D % 2 || H % 2 || W % 2 || E != 48 || Cin_packed <= 0 || Cin_packed % 8 ||
      Cin_packed > 32 || Cin_stride < Cin_packed || Cin_stride % 8 || !in || !w_packed || !bias || !out ||
      out_stride < out_off + E || out_stride % 8 || out_off % 8 || (tadd && tadd_stride < E))
    return DUA_ERR_ARG;
  if (dtype == DUA_F16 && (((size_t)bias & 15) || ((size_t)w_packed & 15) || (tadd && (tadd_stride % 4 || ((size_t)tadd & 15))) ||
                           (x && ((size_t)x & 15)) || (emb && ((size_t)emb & 7))))
    return DUA_ERR_ARG;                                  // the MFMA form moves four-channel pieces
  const long total = (long)B * (D / 2) * (H / 2) * (W / 2);
  const size_t lds = (size_t)8 * Cin_packed * E * sizeof(float);
  dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == DUA_F16)            // MFMA form; the fp32 parity mode keeps the VALU kernel (an fmaf chain per output)
    return launch_patch_embed_mfma(B, D, H, W, Cin_stride, Cin_packed, in, w_packed, bias, tadd, tadd_stride, eps, emb, x, out,
                                   out_stride, out_off, (hipStream_t)stream);
  else if (dtype == DUA_F32)
    hipLaunchKernelGGL((patch_embed_kernel<float, 48>), grid, dim3(256), lds, (hipStream_t)stream, B, D, H, W, Cin_stride,
                       Cin_packed, (const float*)in, w_packed, bias, tadd, tadd_stride, eps, (const float*)emb, x, (float*)out,
                       out_stride, out_off);
  else return DUA_ERR_ARG;
  return (int)hipGetLastError();
}

int dua_instnorm_stats(int dtype, int N, long voxels, int C, const void* x, int x_stride, int x_off, dua_stat_word* stats,
                       int c_pad, void* stream) {
  using namespace dua;
  if (N <= 0 || voxels <= 0 || C <= 0 || C % 8 || C > 2048 || !x || !stats || c_pad < C || x_stride % 8 || x_off % 8 ||
      x_stride < x_off + C)
    return DUA_ERR_ARG;
  const int epg = dtype == DUA_F16 ? 8 : 4;
  const int groups = C / epg;
  if (groups > 256) return DUA_ERR_ARG;
  const int vlanes = 256 / groups;
  long b = (voxels + (long)vlanes * 8 - 1) / ((long)vlanes * 8);
  dim3 grid((unsigned)(b > 1024 ? 1024 : (b < 1 ? 1 : b)), N);
  const size_t lds = (size_t)vlanes * C * 2 * (dtype == DUA_F32 ? sizeof(double) : sizeof(float));
  if (dtype == DUA_F16)
    hipLaunchKernelGGL(instnorm_stats_kernel<f16>, grid, dim3(256), lds, (hipStream_t)stream, (const f16*)x, x_stride, x_off,
                       voxels, C, stats, c_pad);
  else if (dtype == DUA_F32)
    hipLaunchKernelGGL(instnorm_stats_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, (const float*)x, x_stride,
                       x_off, voxels, C, stats, c_pad);
  else return DUA_ERR_ARG;
  return (int)hipGetLastError();
}

int dua_gelu(int dtype, long elems, void* x, void* stream) {
  using namespace dua;
  if (elems <= 0 || elems % 8 || !x) return DUA_ERR_ARG;
  const long groups = elems / (dtype == DUA_F16 ? 8 : 4);
  long b = (groups + 255) / 256;
  dim3 grid((unsigned)(b > 8192 ? 8192 : b));
  if (dtype == DUA_F16) hipLaunchKernelGGL(gelu_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, (f16*)x, groups);
  else if (dtype == DUA_F32) hipLaunchKernelGGL(gelu_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (float*)x, groups);
  else return DUA_ERR_ARG;
  return (int)hipGetLastError();
}

}  // extern "C"
