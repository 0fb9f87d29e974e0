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
// One wave per output token: lane l owns elements l, l + 64, ... of the 8C gathered vector (8C is a multiple of 64), held in
// registers between the moments and the normalisation; corner and channel of an element advance incrementally (no division).
template <typename T>
__global__ __launch_bounds__(256) void patch_merge_norm_kernel(const float* __restrict__ x, const T* __restrict__ y, int B, int D,
                                                               int H, int W, int C, int legacy, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, T* __restrict__ out) {
  constexpr int MAXV = 48;                               // 8C / 64 values per lane, C <= 384
  const int D2 = (D + 1) / 2, H2 = (H + 1) / 2, W2 = (W + 1) / 2;
  const long ntok = (long)B * D2 * H2 * W2;
  const int lane = threadIdx.x & 63;
  const long tok = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (tok >= ntok) return;
  const int w2 = (int)(tok % W2), h2 = (int)((tok / W2) % H2), d2 = (int)((tok / ((long)W2 * H2)) % D2), b = (int)(tok / ((long)W2 * H2 * D2));
  // corner k of the gathered vector -> (di, dj, dk); V2: itertools.product order; legacy: patch.py:82-89
  const int leg[8] = {0, 4, 2, 1, 5, 2, 1, 7};          // bit 2 = d offset, bit 1 = h offset, bit 0 = w offset
  const int E = 8 * C, nper = E >> 6;
  float v[MAXV];
  float s = 0.f;
  int k = 0, c = lane;
  while (c >= C) { c -= C; ++k; }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    v[i] = 0.f;
    if (i < nper) {
      const int code = legacy ? leg[k] : k;
      const int d = 2 * d2 + (code >> 2), h = 2 * h2 + ((code >> 1) & 1), w = 2 * w2 + (code & 1);
      if (d < D && h < H && w < W) {                     // F.pad(..., value 0) of odd extents
        const long idx = ((((long)b * D + d) * H + h) * W + w) * C + c;
        v[i] = y ? x[idx] + (float)y[idx] : x[idx];
      }
      s += v[i];
      c += 64;
      while (c >= C) { c -= C; ++k; }
    }
  }
  s = wave_sum64(s);
  const float mean = s / (float)E;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (i < nper) { const float dl = v[i] - mean; q = fmaf(dl, dl, q); }
  q = wave_sum64(q);
  const float rstd = rsqrtf(q / (float)E + eps);
  T* o = out + tok * E;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (i < nper) { const int e = lane + 64 * i; o[e] = (T)((v[i] - mean) * rstd * gamma[e] + beta[e]); }
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
  dim3 grid((unsigned)((ntok + 3) / 4));
  if (dtype == DUA_F16)
    hipLaunchKernelGGL(dua::patch_merge_norm_kernel<dua::f16>, grid, dim3(256), 0, (hipStream_t)stream, x, (const dua::f16*)y, B, D,
                       H, W, C, legacy, gamma, beta, eps, (dua::f16*)out);
  else if (dtype == DUA_F32)
    hipLaunchKernelGGL(dua::patch_merge_norm_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, x, (const float*)y, B, D, H, W,
                       C, legacy, gamma, beta, eps, (float*)out);
  else return DUA_ERR_ARG;
  return (int)hipGetLastError();
}

int dua_residual_norm_act(int dtype, int N, long voxels, int C, const void* raw, int raw_stride, const dua_in_norm* in,
                          const void* res, int res_stride, const dua_in_norm* res_in, void* out, int out_stride, int out_off,
                          float slope, const void* post_add, int post_stride, int post_off, const void* ra_src,
                          int ra_stride, int ra_off, void* stream) {
  if (!raw || !in || !in->stats || !res || !out || N <= 0 || voxels <= 0 || C <= 0 || C % 8 || raw_stride % 8 ||
      res_stride % 8 || out_stride % 8 || out_off % 8 || C > 2048)
    return DUA_ERR_ARG;
  if (res_in && !res_in->stats) return DUA_ERR_ARG;
  if (post_add && (post_stride % 8 || post_off % 8 || post_stride < post_off + C)) return DUA_ERR_ARG;
  if (ra_src && (ra_stride % 8 || ra_off % 8 || ra_stride < ra_off + C)) return DUA_ERR_ARG;
  const dua::InXform xf = dua::make_xform(in, C), rf = dua::make_xform(res_in, C);
  long b = (voxels * (C / 8) + 255) / 256;
  dim3 grid((unsigned)(b > 4096 ? 4096 : b), N);
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
