// Backward of  a = LeakyReLU(InstanceNorm3d_affine(y)) [+ add[n, c]] [+ emb]  (MONAI ADN "NDA" + the timestep-embedding
// bias of TwoConv.forward, models/basic_unet/denoiser.py:63-67,206-207) for the training step (train.py:258-268).
//
// Forward kept the raw convolution output y and its per-(n, c) sums, never the normalised tensor, so both passes
// recompute z = (y - mean) * rstd * gamma + beta from them:
//   reduce : S0 = sum dA,  S1 = sum dZ,  S2 = sum dZ * zhat      with dZ = dA * (z > 0 ? 1 : slope), zhat = (y - mean) * rstd
//            -> d add[n, c] = S0,  d beta[c] = sum_n S1,  d gamma[c] = sum_n S2
//   apply  : dY = gamma * rstd * (dZ - S1 / V - zhat * S2 / V)
// Both are one streaming pass over dA and y (HBM bound): thread = one 16-byte channel group, fixed per thread, walking
// voxels; per-block LDS reduction, then fp64 atomics into one of 8 replica rows (as the forward statistics).
#include "common.hpp"
#include "../../include/dua_hip.h"

namespace dua {

// mean / rstd / gamma / beta of channels [0, C) into LDS (16 channels per wave and pass, see stats_read_wave16)
__device__ __forceinline__ void norm_preamble(const InXform& xf, int n, int C, float* mu, float* rs, float* ga, float* be) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int c0 = wave * 16; c0 < C; c0 += nw * 16) {
    const int c = c0 + (lane & 15);
    const bool ok = c < C;
    const int cc = ok ? c : C - 1;
    const float gam = xf.gamma[cc], bet = xf.beta[cc];
    double S, Q;
    stats_read_wave16(xf.stats, n, xf.c_pad, cc, S, Q);
    const double mean = S * xf.inv_count;
    double var = Q * xf.inv_count - mean * mean;
    var = var > 0 ? var : 0;
    if (ok && lane < 16) {
      mu[c] = (float)mean;
      rs[c] = (float)(1.0 / sqrt(var + (double)xf.eps));
      ga[c] = gam;
      be[c] = bet;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void in_bwd_reduce_kernel(const T* __restrict__ dA, int da_stride, int da_off,
                                                            const T* __restrict__ raw, int raw_stride, int raw_off,
                                                            InXform xf, int C, long vox, double* __restrict__ sums) {
  using Frag = typename Elem<T>::Frag;
  constexpr int EPG = Elem<T>::EPG;
  extern __shared__ float sm[];
  float* mu = sm; float* rs = sm + C; float* ga = sm + 2 * C; float* be = sm + 3 * C;
  float* red = sm + 4 * C;                    // [256][3 * EPG]
  const int gpc = C / EPG, n = blockIdx.y;
  norm_preamble(xf, n, C, mu, rs, ga, be);
  __syncthreads();
  const int tpg = 256 / gpc;                  // threads sharing one channel group; threads beyond tpg * gpc idle
  const int cg = threadIdx.x % gpc, vl = threadIdx.x / gpc;
  float m[EPG], r[EPG], g[EPG], b[EPG], s0[EPG], s1[EPG], s2[EPG];
#pragma unroll
  for (int e = 0; e < EPG; ++e) {
    m[e] = mu[cg * EPG + e]; r[e] = rs[cg * EPG + e]; g[e] = ga[cg * EPG + e]; b[e] = be[cg * EPG + e];
    s0[e] = s1[e] = s2[e] = 0.f;
  }
  const float slope = xf.slope;
  for (long v = (long)blockIdx.x * tpg + vl; v < vox && vl < tpg; v += (long)gridDim.x * tpg) {
    const long gv = n * vox + v;
    const Frag da = *(const Frag*)(dA + gv * da_stride + da_off + cg * EPG);
    const Frag y = *(const Frag*)(raw + gv * raw_stride + raw_off + cg * EPG);
#pragma unroll
    for (int e = 0; e < EPG; ++e) {
      const float zh = ((float)y[e] - m[e]) * r[e];
      const float z = fmaf(zh, g[e], b[e]);
      const float d = (float)da[e];
      const float dz = z > 0.f ? d : d * slope;
      s0[e] += d; s1[e] += dz; s2[e] = fmaf(dz, zh, s2[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < EPG; ++e) {
    red[threadIdx.x * 3 * EPG + e] = s0[e];
    red[threadIdx.x * 3 * EPG + EPG + e] = s1[e];
    red[threadIdx.x * 3 * EPG + 2 * EPG + e] = s2[e];
  }
  __syncthreads();
  // thread t < 3 * C: (which, channel) -> sum over the tpg threads of that channel group
  for (int t = threadIdx.x; t < 3 * C; t += 256) {
    const int which = t / C, c = t % C, cgc = c / EPG, e = c % EPG;
    double acc = 0;
    for (int j = 0; j < tpg; ++j) acc += (double)red[(j * gpc + cgc) * 3 * EPG + which * EPG + e];
    unsafeAtomicAdd(sums + (((long)n * STAT_REPLICAS + (blockIdx.x & (STAT_REPLICAS - 1))) * xf.c_pad + c) * 4 + which, acc);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void in_bwd_apply_kernel(const T* __restrict__ dA, int da_stride, int da_off,
                                                           const T* __restrict__ raw, int raw_stride, int raw_off,
                                                           InXform xf, const double* __restrict__ sums, int C, long vox,
                                                           T* __restrict__ out, int out_stride, int out_off,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           float* __restrict__ dadd) {
  using Frag = typename Elem<T>::Frag;
  constexpr int EPG = Elem<T>::EPG;
  extern __shared__ float sm[];
  float* mu = sm; float* rs = sm + C; float* ga = sm + 2 * C; float* be = sm + 3 * C;
  float* k1 = sm + 4 * C; float* k2 = sm + 5 * C;       // S1 / V, S2 / V
  const int gpc = C / EPG, n = blockIdx.y;
  norm_preamble(xf, n, C, mu, rs, ga, be);
  for (int c = threadIdx.x; c < C; c += 256) {
    double a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
    for (int r = 0; r < STAT_REPLICAS; ++r) {
      const double* p = sums + (((long)n * STAT_REPLICAS + r) * xf.c_pad + c) * 4;
      a0 += p[0]; a1 += p[1]; a2 += p[2];
    }
    k1[c] = (float)(a1 * xf.inv_count);
    k2[c] = (float)(a2 * xf.inv_count);
    // parameter gradients of this layer, by the first block of every sample (consecutive lanes, consecutive channels):
    // d add[n][c] = sum dA, d beta[c] = sum_n sum dZ, d gamma[c] = sum_n sum dZ * zhat (buffers zeroed by the caller)
    if (blockIdx.x == 0) {
      if (dadd) dadd[(long)n * C + c] = (float)a0;
      if (dbeta) unsafeAtomicAdd(dbeta + c, (float)a1);
      if (dgamma) unsafeAtomicAdd(dgamma + c, (float)a2);
    }
  }
  __syncthreads();
  const float slope = xf.slope;
  const long total = vox * gpc;
  if (256 % gpc == 0) {
    // the grid stride is a multiple of gpc, so a thread keeps ONE channel group: its six per-channel constants live in
    // registers (read from LDS inside the loop they made this HBM-bound pass LDS-bound: 48 ds_reads per 16-byte group,
    // SQ_LDS_IDX_ACTIVE ~ 70 % of the kernel's cycles, half of them bank conflicts)
    const int cg = threadIdx.x % gpc;
    float m[EPG], r[EPG], g[EPG], b[EPG], q1[EPG], q2[EPG];
#pragma unroll
    for (int e = 0; e < EPG; ++e) {
      const int c = cg * EPG + e;
      m[e] = mu[c]; r[e] = rs[c]; g[e] = ga[c]; b[e] = be[c]; q1[e] = k1[c]; q2[e] = k2[c];
    }
    for (long it = blockIdx.x * 256L + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
      const long gv = n * vox + it / gpc;
      const Frag da = *(const Frag*)(dA + gv * da_stride + da_off + cg * EPG);
      const Frag y = *(const Frag*)(raw + gv * raw_stride + raw_off + cg * EPG);
      Frag o;
#pragma unroll
      for (int e = 0; e < EPG; ++e) {
        const float zh = ((float)y[e] - m[e]) * r[e];
        const float z = fmaf(zh, g[e], b[e]);
        const float d = (float)da[e];
        const float dz = z > 0.f ? d : d * slope;
        o[e] = (T)(g[e] * r[e] * (dz - q1[e] - zh * q2[e]));
      }
      *(Frag*)(out + gv * out_stride + out_off + cg * EPG) = o;
    }
    return;
  }
  for (long it = blockIdx.x * 256L + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
    const int cg = (int)(it % gpc);
    const long gv = n * vox + it / gpc;
    const Frag da = *(const Frag*)(dA + gv * da_stride + da_off + cg * EPG);
    const Frag y = *(const Frag*)(raw + gv * raw_stride + raw_off + cg * EPG);
    Frag o;
#pragma unroll
    for (int e = 0; e < EPG; ++e) {
      const int c = cg * EPG + e;
      const float zh = ((float)y[e] - mu[c]) * rs[c];
      const float z = fmaf(zh, ga[c], be[c]);
      const float d = (float)da[e];
      const float dz = z > 0.f ? d : d * slope;
      o[e] = (T)(ga[c] * rs[c] * (dz - k1[c] - zh * k2[c]));
    }
    *(Frag*)(out + gv * out_stride + out_off + cg * EPG) = o;
  }
}

template <typename T>
static int launch_bwd(const dua_norm_bwd_desc* d, const void* dA, const void* raw, const dua_in_norm* in, double* sums,
                      void* out, hipStream_t s, float* dgamma = nullptr, float* dbeta = nullptr, float* dadd = nullptr) {
  constexpr int EPG = Elem<T>::EPG;
  const InXform xf = make_xform(in, d->C);
  const int gpc = d->C / EPG;
  if (gpc > 256) return DUA_ERR_ARG;
  if (!out) {
    const int tpg = 256 / gpc;
    long blocks = (d->voxels + (long)tpg * 8 - 1) / ((long)tpg * 8);      // >= 8 voxels per thread
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    const size_t lds = sizeof(float) * (4 * d->C + 256 * 3 * EPG);
    hipLaunchKernelGGL(in_bwd_reduce_kernel<T>, dim3((unsigned)blocks, d->N), dim3(256), lds, s, (const T*)dA, d->da_stride,
                       d->da_off, (const T*)raw, d->raw_stride, d->raw_off, xf, d->C, d->voxels, sums);
  } else {
    long blocks = (d->voxels * gpc + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(in_bwd_apply_kernel<T>, dim3((unsigned)blocks, d->N), dim3(256), sizeof(float) * 6 * d->C, s,
                       (const T*)dA, d->da_stride, d->da_off, (const T*)raw, d->raw_stride, d->raw_off, xf,
                       (const double*)sums, d->C, d->voxels, (T*)out, d->out_stride, d->out_off, dgamma, dbeta, dadd);
  }
  return (int)hipGetLastError();
}

static bool bwd_args_ok(const dua_norm_bwd_desc* d, const void* dA, const void* raw, const dua_in_norm* in, const void* sums) {
  if (!d || !dA || !raw || !in || !in->stats || !in->gamma || !in->beta || !sums) return false;
  if (d->N <= 0 || d->voxels <= 0 || d->C <= 0 || d->C % 8 || d->C > 1024 || in->c_pad < d->C) return false;
  if (d->da_stride % 8 || d->da_off % 8 || d->raw_stride % 8 || d->raw_off % 8) return false;
  return d->dtype == DUA_F16 || d->dtype == DUA_F32;
}

}  // namespace dua

extern "C" {

int dua_instnorm_bwd_reduce(const dua_norm_bwd_desc* d, const void* dA, const void* raw, const dua_in_norm* in,
                            double* sums, void* stream) {
  if (!dua::bwd_args_ok(d, dA, raw, in, sums)) return DUA_ERR_ARG;
  return d->dtype == DUA_F16 ? dua::launch_bwd<dua::f16>(d, dA, raw, in, sums, nullptr, (hipStream_t)stream)
                             : dua::launch_bwd<float>(d, dA, raw, in, sums, nullptr, (hipStream_t)stream);
}

int dua_instnorm_bwd_apply(const dua_norm_bwd_desc* d, const void* dA, const void* raw, const dua_in_norm* in,
                           const double* sums, void* dY, float* dgamma, float* dbeta, float* dadd, void* stream) {
  if (!dua::bwd_args_ok(d, dA, raw, in, sums) || !dY || d->out_stride % 8 || d->out_off % 8) return DUA_ERR_ARG;
  return d->dtype == DUA_F16
             ? dua::launch_bwd<dua::f16>(d, dA, raw, in, (double*)sums, dY, (hipStream_t)stream, dgamma, dbeta, dadd)
             : dua::launch_bwd<float>(d, dA, raw, in, (double*)sums, dY, (hipStream_t)stream, dgamma, dbeta, dadd);
}

}  // extern "C"

// ---- MaxPool3d(2) backward, fused with the sum of the two gradient paths of x_l -----------------------------------
// x_l feeds the skip half of the decoder's concat AND the pooling of the next level (denoiser.py:100,106,190):
//   out[v] = dA[v] + (v is the arg-max of its 2x2x2 window ? dP[window] : 0)
// The arg-max is recomputed from the stored activation, first maximum in (d, h, w) scan order as torch's max_pool3d.
namespace dua {

template <typename T>
__global__ __launch_bounds__(256) void maxpool2_bwd_add_kernel(const T* __restrict__ act, int act_stride, int act_off,
                                                               const T* __restrict__ dA, int da_stride, int da_off,
                                                               const T* __restrict__ dP, int dp_stride,
                                                               T* __restrict__ out, int out_stride, int C, int D, int H,
                                                               int W, long total) {
  using Frag = typename Elem<T>::Frag;
  constexpr int EPG = Elem<T>::EPG;
  const int gpc = C / EPG, n = blockIdx.y;
  const int W2 = W >> 1, H2 = H >> 1;
  const long vox_n = (long)D * H * W;
  for (long it = blockIdx.x * 256L + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
    const int cg = (int)(it % gpc);
    long v = it / gpc;
    const int pw = (int)(v % W2); v /= W2;
    const int ph = (int)(v % H2); const int pd = (int)(v / H2);
    const long pv = n * (vox_n >> 3) + ((long)pd * H2 + ph) * W2 + pw;
    const Frag g = *(const Frag*)(dP + pv * dp_stride + cg * EPG);
    Frag a[8];
    long gv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int d = 2 * pd + (k >> 2), h = 2 * ph + ((k >> 1) & 1), w = 2 * pw + (k & 1);
      gv[k] = n * vox_n + ((long)d * H + h) * W + w;
      a[k] = *(const Frag*)(act + gv[k] * act_stride + act_off + cg * EPG);
    }
    int arg[EPG];
#pragma unroll
    for (int e = 0; e < EPG; ++e) {
      float mx = (float)a[0][e];
      arg[e] = 0;
#pragma unroll
      for (int k = 1; k < 8; ++k)
        if ((float)a[k][e] > mx) { mx = (float)a[k][e]; arg[e] = k; }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      Frag o;
      if (dA) o = *(const Frag*)(dA + gv[k] * da_stride + da_off + cg * EPG);
#pragma unroll
      for (int e = 0; e < EPG; ++e) {
        const float base = dA ? (float)o[e] : 0.f;
        o[e] = (T)(base + (arg[e] == k ? (float)g[e] : 0.f));
      }
      *(Frag*)(out + gv[k] * out_stride + cg * EPG) = o;
    }
  }
}

}  // namespace dua

extern "C" int dua_maxpool2_bwd_add(int dtype, int N, int D, int H, int W, int C, const void* act, int act_stride,
                                    int act_off, const void* dA, int da_stride, int da_off, const void* dP, int dp_stride,
                                    void* out, int out_stride, void* stream) {
  if (!act || !dP || !out || N <= 0 || C <= 0 || C % 8 || (D | H | W) & 1 || D <= 0 || H <= 0 || W <= 0) return DUA_ERR_ARG;
  if (act_stride % 8 || act_off % 8 || dp_stride % 8 || out_stride % 8 || (dA && (da_stride % 8 || da_off % 8))) return DUA_ERR_ARG;
  const int epg = dtype == DUA_F16 ? 8 : 4;
  const long total = (long)(D / 2) * (H / 2) * (W / 2) * (C / epg);
  long b = (total + 255) / 256;
  dim3 grid((unsigned)(b > 8192 ? 8192 : b), N);
  if (dtype == DUA_F16)
    hipLaunchKernelGGL(dua::maxpool2_bwd_add_kernel<dua::f16>, grid, dim3(256), 0, (hipStream_t)stream, (const dua::f16*)act,
                       act_stride, act_off, (const dua::f16*)dA, da_stride, da_off, (const dua::f16*)dP, dp_stride,
                       (dua::f16*)out, out_stride, C, D, H, W, total);
  else if (dtype == DUA_F32)
    hipLaunchKernelGGL(dua::maxpool2_bwd_add_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)act,
                       act_stride, act_off, (const float*)dA, da_stride, da_off, (const float*)dP, dp_stride, (float*)out,
                       out_stride, C, D, H, W, total);
  else return DUA_ERR_ARG;
  return (int)hipGetLastError();
}
