// InstanceNorm3d(affine) statistics finalisation and the "materialise" pass.
//
// The convolution kernels leave per-(n, c) fp64 sums (x, x^2) of their raw output (8 replica rows).
// Every consumer turns them into the biased variance nn.InstanceNorm3d uses and applies scale =
// gamma*rstd, shift = beta - mean*scale while staging its input (common.hpp InXform/xform_preamble);
// instnorm_finalize writes the same scale/shift out for inspection.
//
// materialize writes an activation that has several consumers: x_i = LeakyReLU(IN(raw)) [+ add[n, c]] +
// embeddings[i] (models/basic_unet/denoiser.py:300-304) into the channel slice of a concat
// buffer, and optionally its MaxPool3d(2) (denoiser.py:100,106) for the next level.
#include "common.hpp"
#include "../../include/dua_hip.h"

namespace dua {

__global__ void instnorm_finalize_kernel(InXform xf, int C, float* scale, float* shift) {
  extern __shared__ float sm[];
  const int n = blockIdx.x;
  xform_preamble(xf, n, C, sm, sm + C, sm + 2 * C);
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) { scale[n * C + c] = sm[c]; shift[n * C + c] = sm[C + c]; }
}

// One thread = one 16-byte channel group of one OUTPUT voxel (pooled: of one 2x2x2 block).
template <typename T, bool POOL, bool EMB>
__global__ __launch_bounds__(256) void materialize_kernel(const T* __restrict__ raw, int C, int raw_stride,
                                                          InXform xf,
                                                          const T* __restrict__ emb, int emb_stride, T* __restrict__ out,
                                                          int out_stride, int out_off, T* __restrict__ pooled,
                                                          int pool_stride, int D, int H, int W, long total, int out_blk) {
  using Frag = typename Elem<T>::Frag;
  constexpr int EPG = Elem<T>::EPG;
  extern __shared__ float sm[];
  const int gpc = C / EPG;
  const int n = blockIdx.y;
  const float slope = xf.slope;
  xform_preamble(xf, n, C, sm, sm + C, sm + 2 * C);
  __syncthreads();
  const long vox_n = (long)D * H * W;
  // The grid stride is a multiple of 256: when the channel-group count divides 256 a thread keeps ONE channel group for all
  // its iterations and its per-channel constants stay in registers.  (Capping the grid at 1024 workgroups to pay the
  // preamble less often was measured: 61.9 -> 65.1 us on the 96^3 x 64 level -- the pass already streams at 5.5 TB/s.)
  const bool fixed = 256 % gpc == 0;
  float sc[EPG], sh[EPG], ad[EPG];
  if (fixed) {
    const int cg = threadIdx.x % gpc;
#pragma unroll
    for (int e = 0; e < EPG; ++e) { sc[e] = sm[cg * EPG + e]; sh[e] = sm[C + cg * EPG + e]; ad[e] = sm[2 * C + cg * EPG + e]; }
  }
  for (long it = blockIdx.x * 256L + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
    const int cg = (int)(it % gpc);
    long v = it / gpc;
    if (!fixed) {
#pragma unroll
      for (int e = 0; e < EPG; ++e) { sc[e] = sm[cg * EPG + e]; sh[e] = sm[C + cg * EPG + e]; ad[e] = sm[2 * C + cg * EPG + e]; }
    }
    if constexpr (!POOL) {
      const long gv = n * vox_n + v;
      Frag x = *(const Frag*)(raw + gv * raw_stride + cg * EPG);
      Frag o;
      Frag ev;
      if constexpr (EMB) ev = *(const Frag*)(emb + gv * emb_stride + cg * EPG);
#pragma unroll
      for (int e = 0; e < EPG; ++e) {
        float y = fmaf((float)x[e], sc[e], sh[e]);
        y = (y > 0.f ? y : y * slope) + ad[e];
        if constexpr (EMB) y += (float)ev[e];
        o[e] = (T)y;
      }
      *(Frag*)(out + n * vox_n * out_stride + chan_off(out_blk, v, out_off + cg * EPG, out_stride, vox_n)) = o;
    } else {
      const int W2 = W >> 1, H2 = H >> 1;
      const int pw = (int)(v % W2); v /= W2;
      const int ph = (int)(v % H2); const int pd = (int)(v / H2);
      float mx[EPG];
#pragma unroll
      for (int e = 0; e < EPG; ++e) mx[e] = -INFINITY;
      // all eight voxels of the block are requested before the first is used (with the embedding test inside the loop hipcc
      // kept the voxels in order, load - wait - store eight times over: the small levels were eight dependent round trips)
      long gvk[8];
      Frag xk[8], evk[EMB ? 8 : 1];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int d = 2 * pd + (k >> 2), h = 2 * ph + ((k >> 1) & 1), w = 2 * pw + (k & 1);
        gvk[k] = n * vox_n + ((long)d * H + h) * W + w;
        xk[k] = *(const Frag*)(raw + gvk[k] * raw_stride + cg * EPG);
      }
      if constexpr (EMB) {
#pragma unroll
        for (int k = 0; k < 8; ++k) evk[k] = *(const Frag*)(emb + gvk[k] * emb_stride + cg * EPG);
      }
      __builtin_amdgcn_sched_barrier(0);       // keep every request in front of the first use
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        Frag o;
#pragma unroll
        for (int e = 0; e < EPG; ++e) {
          float y = fmaf((float)xk[k][e], sc[e], sh[e]);
          y = (y > 0.f ? y : y * slope) + ad[e];
          if constexpr (EMB) y += (float)evk[k][e];
          o[e] = (T)y;
          mx[e] = fmaxf(mx[e], (float)o[e]);
        }
        *(Frag*)(out + n * vox_n * out_stride + chan_off(out_blk, gvk[k] - n * vox_n, out_off + cg * EPG, out_stride, vox_n)) = o;
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
static int launch_materialize(const dua_materialize_desc* d, const void* raw, const dua_in_norm* in,
                              const void* emb, void* out, void* pooled, hipStream_t s) {
  constexpr int EPG = Elem<T>::EPG;
  const long vox = (long)d->D * d->H * d->W;
  const bool pool = pooled != nullptr;
  const long total = (pool ? vox / 8 : vox) * (d->C / EPG);
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  dim3 grid((unsigned)blocks, d->N);
  const InXform xf = make_xform(in, d->C);
  const size_t lds = 3 * sizeof(float) * d->C;
  auto go = [&](auto kern) {
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, (const T*)raw, d->C, d->raw_stride, xf, (const T*)emb, d->emb_stride,
                       (T*)out, d->out_stride, d->out_off, (T*)pooled, pool ? d->pool_stride : 0, d->D, d->H, d->W, total, d->out_blocked ? 1 : 0);
  };
  if (pool) { if (emb) go(materialize_kernel<T, true, true>); else go(materialize_kernel<T, true, false>); }
  else { if (emb) go(materialize_kernel<T, false, true>); else go(materialize_kernel<T, false, false>); }
  return (int)hipGetLastError();
}

}  // namespace dua

extern "C" {

int dua_instnorm_finalize(int N, int C, const dua_in_norm* in, float* scale, float* shift, void* stream) {
  if (N <= 0 || C <= 0 || !in || !in->stats || !in->gamma || !in->beta || in->c_pad < C || !scale || !shift)
    return DUA_ERR_ARG;
  hipLaunchKernelGGL(dua::instnorm_finalize_kernel, dim3(N), dim3(256), 3 * sizeof(float) * C, (hipStream_t)stream,
                     dua::make_xform(in, C), C, scale, shift);
  return (int)hipGetLastError();
}

int dua_materialize(const dua_materialize_desc* d, const void* raw, const dua_in_norm* in,
                    const void* emb, void* out, void* pooled, void* stream) {
  if (!d || !raw || !in || !in->stats || !in->gamma || !in->beta || in->c_pad < d->C || !out) return DUA_ERR_ARG;
  if (d->C % 8 || d->raw_stride % 8 || d->out_stride % 8 || d->out_off % 8 || (emb && d->emb_stride % 8)) return DUA_ERR_ARG;
  if (pooled && ((d->D | d->H | d->W) & 1 || d->pool_stride % 8)) return DUA_ERR_ARG;
  if (d->out_blocked && (d->dtype != DUA_F16 || d->out_stride % 16 || d->out_off % 16 || d->C % 16)) return DUA_ERR_ARG;
  if (d->dtype == DUA_F16) return dua::launch_materialize<dua::f16>(d, raw, in, emb, out, pooled, (hipStream_t)stream);
  if (d->dtype == DUA_F32) return dua::launch_materialize<float>(d, raw, in, emb, out, pooled, (hipStream_t)stream);
  return DUA_ERR_ARG;
}

}  // extern "C"
