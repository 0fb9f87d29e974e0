// Gaussian-diffusion elementwise arithmetic and the fused denoiser tail.
//
// Reference arithmetic (guided_diffusion/gaussian_diffusion.py, GD below):
//   q_sample        GD:187-205   x_t = sqrt(acp[t]) * x0 + sqrt(1-acp[t]) * eps
//   p_mean_variance GD:292-313   x0^ = clamp(model_out, -1, 1); mean = c1[t]*x0^ + c2[t]*x_t
//   p_sample        GD:430-438   x_{t-1} = mean + 1[t!=0] * exp(0.5*logvar[t]) * eps
//   ddim_sample     GD:566-584   e = (sr[t]*x_t - x0^) / srm1[t];
//                                x_{t-1} = x0^*sqrt(acp_prev) + sqrt(1-acp_prev-s^2)*e + 1[t!=0]*s*eps
//   sum of x0^      models/diffusion/diffusion.py:94-98
// The host turns the float64 schedule tables into per-sample fp32 coefficient rows exactly as
// _extract_into_tensor does (GD:904-917, cast to fp32 after the lookup):
//   DDPM row: {c1, c2, 1[t!=0]*exp(0.5*logvar)}        DDIM row: {sr, srm1, sqrt(acp_prev),
//                                                       sqrt(1-acp_prev-s^2), 1[t!=0]*s}
//
// final_conv_sampler fuses, per voxel: InstanceNorm+LeakyReLU of the last decoder block,
// the 1x1x1 final_conv (models/basic_unet/denoiser.py:282,311), the sampler update above, the
// running sum of x0^, and the write of x_{t-1} into the next step's denoiser input buffer
// (channels-last, the torch.cat([image, x]) of denoiser.py:298 in place).
#include "common.hpp"
#include "../../include/dua_hip.h"

namespace dua {

// ---- Philox4x32-10 + Box-Muller (production-mode noise; parity tests inject eps instead) ----
// 32 x 32 -> 64-bit product in ONE quarter-rate instruction (hipcc emits v_mul_hi_u32 + v_mul_lo_u32 for the C form: 40
// quarter-rate multiplies per Philox call, ~20 us of VALU time per step in the tail kernel)
__device__ __forceinline__ void mul_wide(uint32_t k, uint32_t x, uint32_t& hi, uint32_t& lo) {
  unsigned long long p, carry;
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p), "=s"(carry) : "s"(k), "v"(x));
  hi = (uint32_t)(p >> 32); lo = (uint32_t)p;
}
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    uint32_t h0, l0, h1, l1;
    mul_wide(0xD2511F53u, c[0], h0, l0);
    mul_wide(0xCD9E8D57u, c[2], h1, l1);
    const uint32_t n0 = h1 ^ c[1] ^ k0, n2 = h0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = l1; c[2] = n2; c[3] = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
  const float u1 = ((float)a + 0.5f) * 2.3283064365386963e-10f;   // (0,1)
  const float u2 = ((float)b + 0.5f) * 2.3283064365386963e-10f;
  const float rad = sqrtf(-2.f * __logf(u1));
  float s, c;
  __sincosf(6.283185307179586f * u2, &s, &c);
  z0 = rad * c; z1 = rad * s;
}

__device__ __forceinline__ float sampler_update(int mode, const float* k, float out, float xt, float eps, float& xs) {
  xs = fminf(fmaxf(out, -1.f), 1.f);
  if (mode == DUA_MODE_DDPM) {
    const float mean = k[0] * xs + k[1] * xt;
    return mean + k[2] * eps;
  }
  const float e = (k[0] * xt - xs) / k[1];
  const float mean = xs * k[2] + k[3] * e;
  return mean + k[4] * eps;
}

// ---- generic NCDHW elementwise kernels (any model callable between them) ----
__global__ void q_sample_kernel(long per, long total, const float* __restrict__ x0, const float* __restrict__ eps,
                                const float* __restrict__ coef, float* __restrict__ out) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int n = (int)(i / per);
    out[i] = coef[2 * n] * x0[i] + coef[2 * n + 1] * eps[i];
  }
}

__global__ void sampler_step_kernel(int mode, long per, long total, const float* __restrict__ model_out,
                                    const float* __restrict__ x, const float* __restrict__ eps,
                                    const float* __restrict__ coef, float* __restrict__ x_out,
                                    float* __restrict__ xstart_out, float* __restrict__ xstart_sum) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int n = (int)(i / per);
    float xs;
    const float v = sampler_update(mode, coef + 8 * n, model_out[i], x[i], eps[i], xs);
    x_out[i] = v;
    if (xstart_out) xstart_out[i] = xs;
    if (xstart_sum) xstart_sum[i] += xs;
  }
}

// ---- fused tail: one thread = one voxel ----
struct TailArgs {
  const void* raw; InXform xf; const float* wf; const float* bf;
  const float* coef; float* x_state; const float* noise; const int* step_word; void* xin;
  float* xsum; float* logits; float* xstart;
  long vox; int K, raw_stride, C, xin_stride, mode; unsigned seed_lo, seed_hi;
  const unsigned long long* seed_dev;
  // residual form (MFMA kernel only): act = LeakyReLU(xf(raw) + rf(res)) + reverse_attention(ra), real channels [0, kvalid)
  const void* res; InXform rf; int res_stride; const void* ra; int ra_stride, ra_off, kvalid;
};

__device__ __forceinline__ void philox_key(const TailArgs& a, uint32_t& k0, uint32_t& k1) {
  k0 = a.seed_lo; k1 = a.seed_hi;
  if (a.seed_dev) { const unsigned long long s = *a.seed_dev; k0 = (uint32_t)s; k1 = (uint32_t)(s >> 32); }
}

template <typename T, int CX>
__global__ __launch_bounds__(256) void final_conv_sampler_kernel(TailArgs a) {
  using Frag = typename Elem<T>::Frag;
  constexpr int EPG = Elem<T>::EPG;
  extern __shared__ __attribute__((aligned(16))) float wl[];   // [K/4][CX][4] weights, then scale[K], shift[K]
  const int n = blockIdx.y;
  const int K = a.K;
  float* sc = wl + CX * K;
  float* sh = sc + K;
  for (int i = threadIdx.x; i < CX * K; i += 256) {
    const int e = i & 3, c = (i >> 2) % CX, kq = i / (4 * CX);
    wl[i] = c < a.C ? a.wf[c * K + kq * 4 + e] : 0.f;
  }
  xform_preamble(a.xf, n, K, sc, sh, sh + K);
  __syncthreads();
  const long v = blockIdx.x * 256L + threadIdx.x;
  if (v >= a.vox) return;
  float acc[CX];
#pragma unroll
  for (int c = 0; c < CX; ++c) acc[c] = c < a.C ? a.bf[c] : 0.f;
  const T* rp = (const T*)a.raw + ((long)n * a.vox + v) * a.raw_stride;
  for (int kg = 0; kg < K / EPG; ++kg) {
    const Frag f = *(const Frag*)(rp + kg * EPG);
    float y[EPG];
#pragma unroll
    for (int e = 0; e < EPG; ++e) {
      float t = fmaf((float)f[e], sc[kg * EPG + e], sh[kg * EPG + e]);
      t = t > 0.f ? t : t * a.xf.slope;
      y[e] = t;
    }
#pragma unroll
    for (int q = 0; q < EPG / 4; ++q) {
      const f32x4* wq = (const f32x4*)(wl + ((kg * EPG) / 4 + q) * CX * 4);
#pragma unroll
      for (int c = 0; c < CX; ++c) {
        const f32x4 w4 = wq[c];
        acc[c] = fmaf(y[4 * q + 0], w4[0], acc[c]);
        acc[c] = fmaf(y[4 * q + 1], w4[1], acc[c]);
        acc[c] = fmaf(y[4 * q + 2], w4[2], acc[c]);
        acc[c] = fmaf(y[4 * q + 3], w4[3], acc[c]);
      }
    }
  }
  const long gv = (long)n * a.vox + v;
  if (a.logits) {
    for (int c = 0; c < a.C; ++c) a.logits[((long)n * a.C + c) * a.vox + v] = acc[c];
  }
  if (a.mode == DUA_MODE_LOGITS) return;

  float eps[CX];
  if (a.noise) {
#pragma unroll
    for (int c = 0; c < CX; ++c) eps[c] = c < a.C ? a.noise[((long)n * a.C + c) * a.vox + v] : 0.f;
  } else {
    const uint32_t step = a.step_word ? (uint32_t)a.step_word[0] : 0u;
    uint32_t key0, key1;
    philox_key(a, key0, key1);
#pragma unroll
    for (int q = 0; q < CX / 4; ++q) {
      uint32_t ctr[4] = {(uint32_t)gv, (uint32_t)(gv >> 32), step, (uint32_t)q};
      philox4x32_10(ctr, key0, key1);
      box_muller(ctr[0], ctr[1], eps[4 * q], eps[4 * q + 1]);
      box_muller(ctr[2], ctr[3], eps[4 * q + 2], eps[4 * q + 3]);
    }
  }
  float k[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) k[i] = a.coef[8 * n + i];
  float* xs_p = a.x_state + gv * CX;
  float xn[CX], x0[CX];
#pragma unroll
  for (int q = 0; q < CX / 4; ++q) {
    const f32x4 xt = *(const f32x4*)(xs_p + 4 * q);
#pragma unroll
    for (int e = 0; e < 4; ++e) xn[4 * q + e] = sampler_update(a.mode, k, acc[4 * q + e], xt[e], eps[4 * q + e], x0[4 * q + e]);
  }
#pragma unroll
  for (int q = 0; q < CX / 4; ++q) {
    f32x4 o = {xn[4 * q], xn[4 * q + 1], xn[4 * q + 2], xn[4 * q + 3]};
    *(f32x4*)(xs_p + 4 * q) = o;
  }
  if (a.xsum) {
    float* sp = a.xsum + gv * CX;
#pragma unroll
    for (int q = 0; q < CX / 4; ++q) {
      f32x4 s4 = *(f32x4*)(sp + 4 * q);
      s4[0] += x0[4 * q]; s4[1] += x0[4 * q + 1]; s4[2] += x0[4 * q + 2]; s4[3] += x0[4 * q + 3];
      *(f32x4*)(sp + 4 * q) = s4;
    }
  }
  if (a.xstart) {
    for (int c = 0; c < a.C; ++c) a.xstart[((long)n * a.C + c) * a.vox + v] = x0[c];
  }
  if (a.xin) {
    T* xp = (T*)a.xin + gv * a.xin_stride;
#pragma unroll
    for (int g = 0; g < CX / EPG; ++g) {
      Frag o;
#pragma unroll
      for (int e = 0; e < EPG; ++e) {
        const int c = g * EPG + e;
        o[e] = (T)xn[c];
      }
      // channels >= C of the slice hold the conditioning image / zero padding: leave them alone
      if ((g + 1) * EPG <= a.C) *(Frag*)(xp + g * EPG) = o;
      else
        for (int e = 0; e < EPG; ++e)
          if (g * EPG + e < a.C) xp[g * EPG + e] = o[e];
    }
  }
}

// ---- fused tail on the matrix cores (fp16, C <= 16, K a multiple of 32) ----------------------------------
// The 1x1x1 final_conv is a [C x K] x [K x voxels] GEMM: per wave 64 voxels = 4 blocks of 16, MFMA 16x16x32 with the
// CLASSES as rows (A operand = the weights, in registers) and the voxels as columns (B operand = raw fragments loaded
// straight from the tensor, one 16-byte k-group per lane, normalised in registers).  The accumulator then holds, per lane,
// four consecutive classes of ONE voxel: the sampler state moves as 16-byte loads / stores (64 contiguous bytes per voxel,
// 1 KB per instruction -- the voxels-as-rows form moved it as 4-byte accesses and ran at 3.5 TB/s), the next input as 8-byte
// stores, and one Philox4x32 call yields the lane's 4 normals with the same (voxel, class quad) counter as the VALU form
// below, which stays for fp32 parity mode and odd shapes: both forms draw the same noise field.
typedef float f32x4a __attribute__((ext_vector_type(4)));
// RES: the residual form -- the activation is assembled from the last UnetResBlock's two branches and the reverse-attention
// term instead of being read back from a materialised tensor (three 16-byte loads per fragment, two voxel blocks in flight).
// EXTRA: the launch also writes logits / the running sum of x0^ / x0^ itself or reads injected noise (parity tests, config 3) -- a separate
// instantiation, so that the plain sampling step does not hold their values in registers across the next tile's requests.
template <int KS, bool RES = false, bool EXTRA = true>
__global__ __launch_bounds__(256, (RES || KS > 2) ? 2 : 3) void final_conv_sampler_mfma_kernel(TailArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wl[];   // scale[K], shift[K], add[K] (+ the same of the residual branch)
  const int n = blockIdx.y, K = a.K;
  float* sc_l = wl; float* sh_l = wl + K;
  float* rsc_l = wl + 3 * K; float* rsh_l = wl + 4 * K;
  xform_preamble(a.xf, n, a.kvalid, sc_l, sh_l, sh_l + K);
  if (RES) xform_preamble(a.rf, n, a.kvalid, rsc_l, rsh_l, rsh_l + K);
  for (int k = a.kvalid + threadIdx.x; k < K; k += 256) {      // padding channels contribute nothing
    sc_l[k] = 0.f; sh_l[k] = 0.f;
    if (RES) { rsc_l[k] = 0.f; rsh_l[k] = 0.f; }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int vl = lane & 15, kq = lane >> 4;            // B column (voxel) / A row (class) index; k-group and class quad
  float sc[KS][8], sh[KS][8], rsc[RES ? KS : 1][8], rsh[RES ? KS : 1][8];
  // The head runs at (almost) fp32 precision on fp16 MFMAs: weights and activations are split into an fp16 value and the
  // fp16 image of its rounding error, and hi*hi + lo*hi + hi*lo are accumulated (lo*lo is below fp32 round-off).  The logits
  // are the one output that nothing averages afterwards, and the head's rounding was a seventh of the whole network's logit
  // error (tools/precision_sites.py); three MFMAs per k-step instead of one are free in an HBM-bound kernel.
  f16x8 aw[KS], awl[KS];
  bool kok[KS];                                        // this lane's 8 channels of k-step ks are real channels
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    kok[ks] = 32 * ks + 8 * kq < a.kvalid;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 32 * ks + 8 * kq + e;
      sc[ks][e] = sc_l[k]; sh[ks][e] = sh_l[k];
      if (RES) { rsc[ks][e] = rsc_l[k]; rsh[ks][e] = rsh_l[k]; }
      const float wv = vl < a.C ? a.wf[vl * K + k] : 0.f;            // A: row = class vl
      aw[ks][e] = (f16)wv;
      awl[ks][e] = (f16)(wv - (float)aw[ks][e]);
    }
  }
  float bias[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bias[j] = 4 * kq + j < a.C ? a.bf[4 * kq + j] : 0.f;
  float k8[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) k8[i] = a.coef ? a.coef[8 * n + i] : 0.f;
  const uint32_t step = a.step_word ? (uint32_t)a.step_word[0] : 0u;
  uint32_t key0, key1;
  philox_key(a, key0, key1);
  const f16* raw = (const f16*)a.raw + (long)n * a.vox * a.raw_stride;
  const f16* res = RES ? (const f16*)a.res + (long)n * a.vox * a.res_stride : nullptr;
  const f16* rav = RES && a.ra ? (const f16*)a.ra + (long)n * a.vox * a.ra_stride + a.ra_off : nullptr;
  constexpr int MBS = RES ? 2 : 4;                     // voxel blocks whose loads are in flight together
  f16x8 zero8;
#pragma unroll
  for (int e = 0; e < 8; ++e) zero8[e] = (f16)0.f;
  const bool sampling = a.mode != DUA_MODE_LOGITS;
  // A workgroup walks several 256-voxel tiles: the preamble above (per-channel statistics -> scale / shift in double
  // precision, weights to registers) is a few microseconds of dependent loads, too much to pay per 77 KB of traffic.
  const long ntiles = (a.vox + 255) / 256;
  if constexpr (!RES && KS <= 2 && !EXTRA) {
    // The plain sampling step (no logits / x0^ outputs, in-kernel noise), software-pipelined over the tiles of a workgroup: the NEXT tile's operands are requested after this tile's arithmetic and
    // BEFORE this tile's stores are issued.  vmcnt counts loads and stores in issue order, so loads requested behind a tile's
    // stores would not count as landed until those stores were acknowledged by memory: every tile paid a store round trip on
    // top of its load round trip.  With twelve waves per CU in flight it is worth 3-4 us of the 73 us launch; the forms with
    // extra outputs (parity tests, config 3's running sum of x0^) keep the plain loop below: pipelined they need 2 waves fewer per SIMD.
    f16x8 fr[4][KS];
    f32x4 xt[4];
    float ez[EXTRA ? 4 : 1][4];                              // injected noise (parity tests); the plain step draws it in place
    auto request = [&](long tile) {
      const long wb = (tile * 4L + wave) * 64;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        const long v = wb + 16 * mb + vl;
        const long vc = v < a.vox ? v : 0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) fr[mb][ks] = kok[ks] ? *(const f16x8*)(raw + vc * a.raw_stride + 32 * ks + 8 * kq) : zero8;
        xt[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (sampling) xt[mb] = *(const f32x4*)(a.x_state + ((long)n * a.vox + vc) * 16 + 4 * kq);
        if constexpr (EXTRA) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            ez[mb][j] = (sampling && a.noise && 4 * kq + j < a.C) ? a.noise[((long)n * a.C + 4 * kq + j) * a.vox + vc] : 0.f;
        }
      }
    };
    if ((long)blockIdx.x < ntiles) request(blockIdx.x);
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const long wbase = (tile * 4L + wave) * 64;
      f32x4a lg[EXTRA ? 4 : 1];                              // logits of classes 4 kq + j of voxel wbase + 16 mb + vl
      f32x4 xn[4], x0[EXTRA ? 4 : 1];
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        f32x4a acc = {bias[0], bias[1], bias[2], bias[3]};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          f16x8 y, yl;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float t = fmaf((float)fr[mb][ks][e], sc[ks][e], sh[ks][e]);
            t = t > 0.f ? t : t * a.xf.slope;
            y[e] = (f16)t;
            yl[e] = (f16)(t - (float)y[e]);
          }
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(awl[ks], y, acc, 0, 0, 0);    // small terms first
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(aw[ks], yl, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(aw[ks], y, acc, 0, 0, 0);     // columns of voxels >= vox are never stored
        }
        if constexpr (EXTRA) lg[mb] = acc;
        if (sampling) {
          const long gv = (long)n * a.vox + wbase + 16 * mb + vl;
          float eps[4];
          if (!EXTRA || !a.noise) {                              // same counter as the VALU form: (voxel, step, class quad)
            uint32_t ctr[4] = {(uint32_t)gv, (uint32_t)(gv >> 32), step, (uint32_t)kq};
            philox4x32_10(ctr, key0, key1);
            box_muller(ctr[0], ctr[1], eps[0], eps[1]);
            box_muller(ctr[2], ctr[3], eps[2], eps[3]);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) eps[j] = ez[EXTRA ? mb : 0][j];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float x0j;
            xn[mb][j] = sampler_update(a.mode, k8, acc[j], xt[mb][j], eps[j], x0j);
            if constexpr (EXTRA) x0[mb][j] = x0j;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (tile + gridDim.x < ntiles) request(tile + gridDim.x);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        const long v = wbase + 16 * mb + vl;
        if (v >= a.vox) continue;
        const long gv = (long)n * a.vox + v;
        if constexpr (EXTRA) {
          if (a.logits) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (4 * kq + j < a.C) a.logits[((long)n * a.C + 4 * kq + j) * a.vox + v] = lg[mb][j];
          }
        }
        if (!sampling) continue;
        *(f32x4*)(a.x_state + gv * 16 + 4 * kq) = xn[mb];
        if constexpr (EXTRA) {
          if (a.xsum) {
            f32x4 s4 = *(const f32x4*)(a.xsum + gv * 16 + 4 * kq);
#pragma unroll
            for (int j = 0; j < 4; ++j) s4[j] += x0[mb][j];
            *(f32x4*)(a.xsum + gv * 16 + 4 * kq) = s4;
          }
          if (a.xstart) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (4 * kq + j < a.C) a.xstart[((long)n * a.C + 4 * kq + j) * a.vox + v] = x0[mb][j];
          }
        }
        if (a.xin) {
          f16* xp = (f16*)a.xin + gv * a.xin_stride + 4 * kq;
          if (4 * kq + 4 <= a.C) {
            *(f16x4*)xp = f16x4{(f16)xn[mb][0], (f16)xn[mb][1], (f16)xn[mb][2], (f16)xn[mb][3]};
          } else {                     // channels >= C of the slice hold the conditioning image / zero padding: leave them alone
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (4 * kq + j < a.C) xp[j] = (f16)xn[mb][j];
          }
        }
      }
    }
    return;
  }
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  const long wbase = (tile * 4L + wave) * 64;
  // Everything this wave needs from memory is requested up front (raw fragments of all four voxel blocks, the
  // sampler state, injected noise): one memory round trip per wave instead of one per block.
#pragma unroll
  for (int mb0 = 0; mb0 < 4; mb0 += MBS) {
  f16x8 fr[MBS][KS], fr2[RES ? MBS : 1][KS], fr3[RES ? MBS : 1][KS];
  f32x4 xt[MBS];
  float ez[MBS][4];
#pragma unroll
  for (int mb = 0; mb < MBS; ++mb) {
    const long v = wbase + 16 * (mb0 + mb) + vl;
    const long vc = v < a.vox ? v : 0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      fr[mb][ks] = kok[ks] ? *(const f16x8*)(raw + vc * a.raw_stride + 32 * ks + 8 * kq) : zero8;
      if (RES) {
        fr2[mb][ks] = kok[ks] ? *(const f16x8*)(res + vc * a.res_stride + 32 * ks + 8 * kq) : zero8;
        fr3[mb][ks] = kok[ks] && rav ? *(const f16x8*)(rav + vc * a.ra_stride + 32 * ks + 8 * kq) : zero8;
      }
    }
    xt[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (sampling) xt[mb] = *(const f32x4*)(a.x_state + ((long)n * a.vox + vc) * 16 + 4 * kq);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      ez[mb][j] = (sampling && a.noise && 4 * kq + j < a.C) ? a.noise[((long)n * a.C + 4 * kq + j) * a.vox + vc] : 0.f;
  }
#pragma unroll
  for (int mb = 0; mb < MBS; ++mb) {
    f32x4a acc = {bias[0], bias[1], bias[2], bias[3]};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      f16x8 y, yl;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = fmaf((float)fr[mb][ks][e], sc[ks][e], sh[ks][e]);
        if (RES) t += fmaf((float)fr2[mb][ks][e], rsc[ks][e], rsh[ks][e]);      // same order as residual_norm_act_kernel
        t = t > 0.f ? t : t * a.xf.slope;
        if (RES) { const float s = (float)fr3[mb][ks][e]; if (rav) t += s * (1.f - 1.f / (1.f + __expf(-s))); }
        y[e] = (f16)t;
        yl[e] = (f16)(t - (float)y[e]);
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(awl[ks], y, acc, 0, 0, 0);    // small terms first
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(aw[ks], yl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(aw[ks], y, acc, 0, 0, 0);     // columns of voxels >= vox are never stored
    }
    // lane now holds the logits of classes 4 kq + j of voxel wbase + 16 (mb0 + mb) + vl
    const long v = wbase + 16 * (mb0 + mb) + vl;
    if (v >= a.vox) continue;
    const long gv = (long)n * a.vox + v;
    if (a.logits) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * kq + j < a.C) a.logits[((long)n * a.C + 4 * kq + j) * a.vox + v] = acc[j];
    }
    if (!sampling) continue;
    if (!a.noise) {                                          // same counter as the VALU form: (voxel, step, class quad)
      uint32_t ctr[4] = {(uint32_t)gv, (uint32_t)(gv >> 32), step, (uint32_t)kq};
      philox4x32_10(ctr, key0, key1);
      box_muller(ctr[0], ctr[1], ez[mb][0], ez[mb][1]);
      box_muller(ctr[2], ctr[3], ez[mb][2], ez[mb][3]);
    }
    f32x4 xn, x0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x0j;
      xn[j] = sampler_update(a.mode, k8, acc[j], xt[mb][j], ez[mb][j], x0j);
      x0[j] = x0j;
    }
    *(f32x4*)(a.x_state + gv * 16 + 4 * kq) = xn;
    if (a.xsum) {
      f32x4 s4 = *(const f32x4*)(a.xsum + gv * 16 + 4 * kq);
#pragma unroll
      for (int j = 0; j < 4; ++j) s4[j] += x0[j];
      *(f32x4*)(a.xsum + gv * 16 + 4 * kq) = s4;
    }
    if (a.xstart) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * kq + j < a.C) a.xstart[((long)n * a.C + 4 * kq + j) * a.vox + v] = x0[j];
    }
    if (a.xin) {
      f16* xp = (f16*)a.xin + gv * a.xin_stride + 4 * kq;
      if (4 * kq + 4 <= a.C) {
        *(f16x4*)xp = f16x4{(f16)xn[0], (f16)xn[1], (f16)xn[2], (f16)xn[3]};
      } else {                       // channels >= C of the slice hold the conditioning image / zero padding: leave them alone
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (4 * kq + j < a.C) xp[j] = (f16)xn[j];
      }
    }
  }
  }
  }
}

template <typename T, int CX>
static int launch_tail(const dua_tail_desc* d, TailArgs& a, hipStream_t s) {
  const size_t lds = (size_t)(CX * d->K + 3 * d->K) * sizeof(float);
  dim3 grid((unsigned)((d->voxels + 255) / 256), d->N);
  hipLaunchKernelGGL((final_conv_sampler_kernel<T, CX>), grid, dim3(256), lds, s, a);
  return (int)hipGetLastError();
}

template <typename T>
static int dispatch_tail(const dua_tail_desc* d, TailArgs& a, hipStream_t s) {
  switch (d->CX) {
    case 8: return launch_tail<T, 8>(d, a, s);
    case 16: return launch_tail<T, 16>(d, a, s);
    case 24: return launch_tail<T, 24>(d, a, s);
    case 32: return launch_tail<T, 32>(d, a, s);
  }
  return DUA_ERR_ARG;
}

// Grid of the persistent MFMA tail: exactly the workgroups that are resident together (wgs_per_cu x CUs over the N samples),
// each walking its share of the 256-voxel tiles.  A fixed 1024 left 256 workgroups for a second round that ran on a third of the chip.
static inline unsigned tail_grid(long ntiles, int N, int wgs_per_cu) {
  (void)ensure_prepared();
  const int cus = device_cus() > 0 ? device_cus() : 256;
  long g = (long)wgs_per_cu * cus / (N > 0 ? N : 1);
  if (g < 1) g = 1;
  return (unsigned)(ntiles < g ? ntiles : g);
}

static inline unsigned nblk(long total) {
  long b = (total + 255) / 256;
  return (unsigned)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

}  // namespace dua

extern "C" {

int dua_q_sample(int N, long per_sample, const float* x0, const float* eps, const float* coef, float* out, void* stream) {
  if (N <= 0 || per_sample <= 0 || !x0 || !eps || !coef || !out) return DUA_ERR_ARG;
  const long total = N * per_sample;
  hipLaunchKernelGGL(dua::q_sample_kernel, dim3(dua::nblk(total)), dim3(256), 0, (hipStream_t)stream, per_sample, total,
                     x0, eps, coef, out);
  return (int)hipGetLastError();
}

int dua_sampler_step(int mode, int N, long per_sample, const float* model_out, const float* x, const float* eps,
                     const float* coef, float* x_out, float* xstart_out, float* xstart_sum, void* stream) {
  if ((mode != DUA_MODE_DDPM && mode != DUA_MODE_DDIM) || N <= 0 || per_sample <= 0 || !model_out || !x || !eps ||
      !coef || !x_out)
    return DUA_ERR_ARG;
  const long total = N * per_sample;
  hipLaunchKernelGGL(dua::sampler_step_kernel, dim3(dua::nblk(total)), dim3(256), 0, (hipStream_t)stream, mode,
                     per_sample, total, model_out, x, eps, coef, x_out, xstart_out, xstart_sum);
  return (int)hipGetLastError();
}

static int tail_entry(const dua_tail_desc* d, const void* raw, const dua_in_norm* in, const dua_tail_residual* r,
                      const float* wf, const float* bf, const float* coef, float* x_state, const float* noise,
                      const int* step_word, void* xin, float* xstart_sum, float* logits, float* xstart, void* stream) {
  if (!d || !raw || !wf || !bf) return DUA_ERR_ARG;
  const bool identity = !in || !in->stats;        // raw is an already materialised activation (Swin-UNETR's decoder1 output)
  const int real = r ? r->channels : d->K;        // channels actually present in raw
  if (!identity && (!in->gamma || !in->beta || in->c_pad < real)) return DUA_ERR_ARG;
  if (d->K % 8 || d->raw_stride % 8 || real > d->raw_stride || d->C <= 0 || d->C > d->CX || d->K > 512) return DUA_ERR_ARG;
  if (r) {
    if (identity || d->dtype != DUA_F16 || d->CX != 16 || (d->K != 32 && d->K != 64) || real <= 0 || real % 8 || real > d->K ||
        !r->res || r->res_stride % 8 || real > r->res_stride || !r->res_norm.stats || !r->res_norm.gamma || !r->res_norm.beta ||
        r->res_norm.c_pad < real)
      return DUA_ERR_ARG;
    if (r->ra_src && (r->ra_stride % 8 || r->ra_off % 8 || r->ra_off + real > r->ra_stride)) return DUA_ERR_ARG;
  }
  if (d->mode == DUA_MODE_LOGITS) { if (!logits) return DUA_ERR_ARG; }
  else if (d->mode == DUA_MODE_DDPM || d->mode == DUA_MODE_DDIM) { if (!coef || !x_state) return DUA_ERR_ARG; }
  else return DUA_ERR_ARG;
  if (xin && d->xin_stride % 8) return DUA_ERR_ARG;
  dua::TailArgs a;
  a.raw = raw; a.xf = dua::make_xform(identity ? nullptr : in, d->K);
  if (identity) a.xf.slope = 1.f;
  a.wf = wf; a.bf = bf; a.coef = coef; a.x_state = x_state;
  a.noise = noise; a.step_word = step_word; a.xin = xin; a.xsum = xstart_sum; a.logits = logits; a.xstart = xstart;
  a.vox = d->voxels; a.K = d->K; a.raw_stride = d->raw_stride; a.C = d->C; a.xin_stride = d->xin_stride;
  a.mode = d->mode;
  a.seed_lo = (unsigned)(d->seed & 0xffffffffull); a.seed_hi = (unsigned)(d->seed >> 32);
  a.seed_dev = d->seed_dev;
  a.res = nullptr; a.ra = nullptr; a.kvalid = d->K; a.res_stride = a.ra_stride = a.ra_off = 0; a.rf = dua::InXform{};
  if (r) {
    a.res = r->res; a.res_stride = r->res_stride; a.rf = dua::make_xform(&r->res_norm, real);
    a.ra = r->ra_src; a.ra_stride = r->ra_stride; a.ra_off = r->ra_off; a.kvalid = real;
    const long ntiles = (d->voxels + 255) / 256;
    dim3 grid(dua::tail_grid(ntiles, d->N, 2), d->N);
    const size_t lds = (size_t)6 * d->K * sizeof(float);
    if (d->K == 32) hipLaunchKernelGGL((dua::final_conv_sampler_mfma_kernel<1, true>), grid, dim3(256), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((dua::final_conv_sampler_mfma_kernel<2, true>), grid, dim3(256), lds, (hipStream_t)stream, a);
    return (int)hipGetLastError();
  }
  if (d->dtype == DUA_F16 && d->CX == 16 && d->K % 32 == 0 && (d->K == 32 || d->K == 64 || d->K == 128)) {
    const long ntiles = (d->voxels + 255) / 256;
    dim3 grid(dua::tail_grid(ntiles, d->N, 3), d->N);
    const size_t lds = (size_t)6 * d->K * sizeof(float);
    const bool extra = a.logits || a.xsum || a.xstart || a.noise || a.mode == DUA_MODE_LOGITS;
    if (d->K == 64 && !extra) hipLaunchKernelGGL((dua::final_conv_sampler_mfma_kernel<2, false, false>), grid, dim3(256), lds, (hipStream_t)stream, a);
    else if (d->K == 32) hipLaunchKernelGGL(dua::final_conv_sampler_mfma_kernel<1>, grid, dim3(256), lds, (hipStream_t)stream, a);
    else if (d->K == 64) hipLaunchKernelGGL(dua::final_conv_sampler_mfma_kernel<2>, grid, dim3(256), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(dua::final_conv_sampler_mfma_kernel<4>, grid, dim3(256), lds, (hipStream_t)stream, a);
    return (int)hipGetLastError();
  }
  if (d->dtype == DUA_F16) return dua::dispatch_tail<dua::f16>(d, a, (hipStream_t)stream);
  if (d->dtype == DUA_F32) return dua::dispatch_tail<float>(d, a, (hipStream_t)stream);
  return DUA_ERR_ARG;
}

int dua_final_conv_sampler(const dua_tail_desc* d, const void* raw, const dua_in_norm* in,
                           const float* wf, const float* bf, const float* coef, float* x_state, const float* noise,
                           const int* step_word, void* xin, float* xstart_sum, float* logits, float* xstart,
                           void* stream) {
  return tail_entry(d, raw, in, nullptr, wf, bf, coef, x_state, noise, step_word, xin, xstart_sum, logits, xstart, stream);
}

int dua_final_conv_sampler_res(const dua_tail_desc* d, const void* raw, const dua_in_norm* in, const dua_tail_residual* r,
                               const float* wf, const float* bf, const float* coef, float* x_state, const float* noise,
                               const int* step_word, void* xin, float* xstart_sum, float* logits, float* xstart,
                               void* stream) {
  if (!r) return DUA_ERR_ARG;
  return tail_entry(d, raw, in, r, wf, bf, coef, x_state, noise, step_word, xin, xstart_sum, logits, xstart, stream);
}

}  // extern "C"
