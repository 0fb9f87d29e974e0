// Memory-bound pieces of the diff_swin_unetr variant (BASELINE config 5).
//
//  patch_merge_norm : PatchMerging.forward up to (not including) the reduction Linear
//                     (models/swin_unetr/patch.py:44-61 and the legacy 3-D gather :70-91): zero-pad odd extents, gather
//                     the 2x2x2 neighbourhood of every second voxel into an 8C vector, LayerNorm(8C).  The LEGACY
//                     gather reads corner (0,1,0) twice and (0,0,1) twice (x5 == x2, x6 == x3) and never reads
//                     (1,1,0) / (0,1,1); it is what the reference's default PatchMerging computes, so it is reproduced.
//  residual_norm_act: the tail of UnetResBlock.forward (models/swin_unetr/blocks.py:308-316):
//                     out = LeakyReLU( InstanceNorm(conv2 raw) + residual ), residual = the block input, or
//                     InstanceNorm(conv3 raw) when the block changes the channel count (conv3 = 1x1x1); then the two adds
//                     of SwinUNETRDenoiser.forward on a block's output (swin_unetr/denoiser.py:370-399): + embeddings[k]
//                     for the encoder blocks, + reverse_attention(skip) = skip * (1 - sigmoid(skip)) for the decoders.
// Both are one streaming pass over their tensors (HBM bound); tokens / voxels are channels-last, as the reference's
// Swin tensors already are ([b, d, h, w, c]).
#include "common.hpp"
#include "../../include/dua_hip.h"

namespace dua {

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// one wave per output token; lanes stride over the 8C gathered elements (two passes: moments, then normalise + store)
// x is the fp32 token stream of the stage; y (optional, T) is the last block's MLP output still to be added to it
// (transformer.py:477-480) -- the sum is formed on the fly, the stream itself is dead after the merge.
// WPT waves per output token (1: a workgroup holds 4 tokens; 4: the whole workgroup works on one token -- the coarse stages
// have 216 and 27 output tokens of 1536 / 3072 elements, where one wave per token leaves 27 waves walking 48 dependent loads
// each: 59-88 us for a few hundred KB).  A lane owns elements l, l + 64 WPT, ... of the 8C gathered vector, held in registers
// between the moments and the normalisation; corner and channel of an element advance incrementally (no division), and
// every load is issued unconditionally on a clamped address (a load under a lane-dependent branch sits in its own basic
// block and is waited for there).  NPER > 0 fixes the values per lane at compile time (the Swin-UNETR widths: 6 or 12): with
// the run-time count every one of the 48 / WPT slots keeps its value, address and flag in registers (~200 VGPRs, two waves
// per SIMD -- 59 us for the 32 MB of the 48^3-token merge).
template <typename T, int WPT, int NPER = 0>
__global__ __launch_bounds__(256) void patch_merge_norm_kernel(const float* __restrict__ x, const T* __restrict__ y, int B, int D,
                                                               int H, int W, int C, int legacy, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, T* __restrict__ out) {
  constexpr int MAXV = NPER > 0 ? NPER : 48 / WPT;       // 8C / (64 WPT) values per lane, C <= 384
  constexpr int STRIDE = 64 * WPT;
  __shared__ float part[4][2];
  const int D2 = (D + 1) / 2, H2 = (H + 1) / 2, W2 = (W + 1) / 2;
  const long ntok = (long)B * D2 * H2 * W2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long tok = WPT == 1 ? blockIdx.x * 4L + wave : (long)blockIdx.x;
  const bool live = tok < ntok;
  const long tk = live ? tok : 0;
  const int w2 = (int)(tk % W2), h2 = (int)((tk / W2) % H2), d2 = (int)((tk / ((long)W2 * H2)) % D2), b = (int)(tk / ((long)W2 * H2 * D2));
  // corner k of the gathered vector -> (di, dj, dk); V2: itertools.product order; legacy: patch.py:82-89
  const int leg[8] = {0, 4, 2, 1, 5, 2, 1, 7};          // bit 2 = d offset, bit 1 = h offset, bit 0 = w offset
  const int E = 8 * C, nper = NPER > 0 ? NPER : E / STRIDE;
  const int e0 = WPT == 1 ? lane : threadIdx.x;
  float v[MAXV];
  long idx[MAXV];
  bool in[MAXV];
  int k = 0, c = e0;
  while (c >= C) { c -= C; ++k; }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    in[i] = false; idx[i] = 0;
    if (i < nper) {
      const int code = legacy ? leg[k & 7] : (k & 7);
      const int d = 2 * d2 + (code >> 2), h = 2 * h2 + ((code >> 1) & 1), w = 2 * w2 + (code & 1);
      in[i] = d < D && h < H && w < W;                   // F.pad(..., value 0) of odd extents
      idx[i] = in[i] ? ((((long)b * D + d) * H + h) * W + w) * C + c : 0;
      c += STRIDE;
      while (c >= C) { c -= C; ++k; }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) v[i] = i < nper ? x[idx[i]] : 0.f;
  if (y) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (i < nper) v[i] += (float)y[idx[i]];
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) { v[i] = in[i] ? v[i] : 0.f; s += v[i]; }
  s = wave_sum64(s);
  if (WPT > 1) {
    if (lane == 0) part[wave][0] = s;
    __syncthreads();
    s = (part[0][0] + part[1][0]) + (part[2][0] + part[3][0]);
  }
  const float mean = s / (float)E;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (i < nper) { const float dl = v[i] - mean; q = fmaf(dl, dl, q); }
  q = wave_sum64(q);
  if (WPT > 1) {
    if (lane == 0) part[wave][1] = q;
    __syncthreads();
    q = (part[0][1] + part[1][1]) + (part[2][1] + part[3][1]);
  }
  const float rstd = rsqrtf(q / (float)E + eps);
  if (!live) return;
  T* o = out + tok * E;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (i < nper) { const int e = e0 + STRIDE * i; o[e] = (T)((v[i] - mean) * rstd * gamma[e] + beta[e]); }
}

template <typename T>
__global__ __launch_bounds__(256) void residual_norm_act_kernel(const T* __restrict__ raw, int raw_stride, InXform xf,
                                                                const T* __restrict__ res, int res_stride, InXform rf,
                                                                int has_rf, long vox, int C, T* __restrict__ out,
                                                                int out_stride, int out_off, float slope,
                                                                const T* __restrict__ post, int post_stride, int post_off,
                                                                const T* __restrict__ ra, int ra_stride, int ra_off) {
  extern __shared__ float tbl[];      // scale, shift of raw; scale, shift of the residual
  const int n = blockIdx.y;
  float* sc = tbl; float* sh = tbl + C; float* rsc = tbl + 2 * C; float* rsh = tbl + 3 * C; float* dump = tbl + 4 * C;
  xform_preamble(xf, n, C, sc, sh, dump);
  if (has_rf) xform_preamble(rf, n, C, rsc, rsh, dump);
  __syncthreads();
  constexpr int EPG = Elem<T>::EPG;
  using Frag = typename Elem<T>::Frag;
  const int groups = C / EPG;
  const long items = vox * groups;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < items; i += (long)gridDim.x * 256) {
    const long v = i / groups;
    const int g = (int)(i - v * groups), c0 = g * EPG;
    const Frag a = *(const Frag*)(raw + ((long)n * vox + v) * raw_stride + c0);
    const Frag r = *(const Frag*)(res + ((long)n * vox + v) * res_stride + c0);
    Frag o, pa, rs;
    if (post) pa = *(const Frag*)(post + ((long)n * vox + v) * post_stride + post_off + c0);
    if (ra) rs = *(const Frag*)(ra + ((long)n * vox + v) * ra_stride + ra_off + c0);
#pragma unroll
    for (int e = 0; e < EPG; ++e) {
      float y = fmaf((float)a[e], sc[c0 + e], sh[c0 + e]);
      const float rr = has_rf ? fmaf((float)r[e], rsc[c0 + e], rsh[c0 + e]) : (float)r[e];
      y += rr;
      y = y > 0.f ? y : y * slope;
      if (post) y += (float)pa[e];                                       // + embeddings[k]   (denoiser.py:370-383)
      if (ra) { const float s = (float)rs[e]; y += s * (1.f - 1.f / (1.f + __expf(-s))); }   // + reverse_attention(skip) (:405-408)
      o[e] = (T)y;
    }
    *(Frag*)(out + ((long)n * vox + v) * out_stride + out_off + c0) = o;
  }
}

}  // namespace dua

extern "C" {

int dua_patch_merge_norm(int dtype, int B, int D, int H, int W, int C, int legacy, const float* x, const void* y,
                         const float* gamma, const float* beta, float eps, void* out, void* stream) {
  if (!x || !gamma || !beta || !out || B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || C > 384) return DUA_ERR_ARG;
  const long ntok = (long)B * ((D + 1) / 2) * ((H + 1) / 2) * ((W + 1) / 2);
  const bool wide = ntok < 2048 && (8 * C) % 256 == 0;        // few tokens: a whole workgroup per token
  dim3 grid((unsigned)(wide ? ntok : (ntok + 3) / 4));
  const int nper = 8 * C / (wide ? 256 : 64);            // values per lane
#define DUA_PM(T_, WPT_, NPER_)                                                                                           \
  hipLaunchKernelGGL((dua::patch_merge_norm_kernel<T_, WPT_, NPER_>), grid, dim3(256), 0, (hipStream_t)stream, x, (const T_*)y, B, \
                     D, H, W, C, legacy, gamma, beta, eps, (T_*)out)
#define DUA_PM_T(T_)                                                                                                      \
  if (wide) { if (nper == 6) DUA_PM(T_, 4, 6); else if (nper == 12) DUA_PM(T_, 4, 12); else if (nper == 3) DUA_PM(T_, 4, 3);  \
              else DUA_PM(T_, 4, 0); }                                                                                    \
  else { if (nper == 6) DUA_PM(T_, 1, 6); else if (nper == 12) DUA_PM(T_, 1, 12); else DUA_PM(T_, 1, 0); }
  if (dtype == DUA_F16) { DUA_PM_T(dua::f16) }
  else if (dtype == DUA_F32) { DUA_PM_T(float) }
  else return DUA_ERR_ARG;
#undef DUA_PM_T
#undef DUA_PM
  return (int)hipGetLastError();
}

int dua_residual_norm_act(int dtype, int N, long voxels, int C, const void* raw, int raw_stride, const dua_in_norm* in,
                          const void* res, int res_stride, const dua_in_norm* res_in, void* out, int out_stride, int out_off,
                          float slope, const void* post_add, int post_stride, int post_off, const void* ra_src,
                          int ra_stride, int ra_off, int background, void* stream) {
  if (!raw || !in || !in->stats || !res || !out || N <= 0 || voxels <= 0 || C <= 0 || C % 8 || raw_stride % 8 ||
      res_stride % 8 || out_stride % 8 || out_off % 8 || C > 2048)
    return DUA_ERR_ARG;
  if (res_in && !res_in->stats) return DUA_ERR_ARG;
  if (post_add && (post_stride % 8 || post_off % 8 || post_stride < post_off + C)) return DUA_ERR_ARG;
  if (ra_src && (ra_stride % 8 || ra_off % 8 || ra_stride < ra_off + C)) return DUA_ERR_ARG;
  const dua::InXform xf = dua::make_xform(in, C), rf = dua::make_xform(res_in, C);
  long b = (voxels * (C / 8) + 255) / 256;
  const long cap = background ? 256 : 4096;       // background: one workgroup per CU (the loop strides over the grid)
  dim3 grid((unsigned)(b > cap ? cap : b), N);
  const size_t lds = (size_t)5 * C * sizeof(float);
  if (dtype == DUA_F16)
    hipLaunchKernelGGL(dua::residual_norm_act_kernel<dua::f16>, grid, dim3(256), lds, (hipStream_t)stream, (const dua::f16*)raw,
                       raw_stride, xf, (const dua::f16*)res, res_stride, rf, res_in ? 1 : 0, voxels, C, (dua::f16*)out,
                       out_stride, out_off, slope, (const dua::f16*)post_add, post_stride, post_off, (const dua::f16*)ra_src,
                       ra_stride, ra_off);
  else if (dtype == DUA_F32)
    hipLaunchKernelGGL(dua::residual_norm_act_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, (const float*)raw,
                       raw_stride, xf, (const float*)res, res_stride, rf, res_in ? 1 : 0, voxels, C, (float*)out, out_stride,
                       out_off, slope, (const float*)post_add, post_stride, post_off, (const float*)ra_src, ra_stride, ra_off);
  else return DUA_ERR_ARG;
  return (int)hipGetLastError();
}

}  // extern "C"
