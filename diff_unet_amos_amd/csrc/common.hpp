// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of the Diff-UNet hot path.
//
// Data layout convention used by every kernel in this directory:
//   activations  : channels-last  [N][D][H][W][Cs]   (Cs = channel stride of the buffer; a
//                  kernel addresses channels [c_off, c_off + C) of it, so concat buffers
//                  are written in place by their producers -- reference torch.cat at
//                  models/basic_unet/denoiser.py:190,298 becomes "no kernel")
//   element type : f16 (production; MFMA 32x32x16, fp32 accumulate) or f32 (parity mode;
//                  MFMA 32x32x2, bit-for-bit an fmaf chain)
//   "k-group"    : 16 bytes of consecutive channels (8 x f16 or 4 x f32) -- the unit a lane
//                  loads for one MFMA operand fragment.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dua {
// Host-side launch state of the library (prepare.hip).  hipFuncSetAttribute acts on the current device only and is not a
// call to make for the first time inside a stream capture or from two threads at once (torch runs backward() on a thread of
// its own): every translation unit REGISTERS the kernels whose dynamic-LDS limit it needs raised (a namespace-scope
// LdsAttrs object, filled in during static initialisation), and ensure_prepared() raises all of them, for the whole
// library, the first time ANY launcher runs on a device -- under a mutex, published through an atomic flag per device.
// dua_prepare() is the same call as a C entry point: plans and trainers run it at construction, before any capture.
struct LdsAttr { const void* fn; int bytes; };
void register_lds_attrs(const LdsAttr* list, int n);
int ensure_prepared();          // 0, DUA_ERR_ARG (no current device) or the hipError_t of a failed attribute call
int device_cus();               // compute units of the current device (cached by ensure_prepared); <= 0 before / on error
struct LdsAttrs {
  template <int N> explicit LdsAttrs(const LdsAttr (&list)[N]) { register_lds_attrs(list, N); }
};


using f16 = _Float16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum : int { DT_F32 = 0, DT_F16 = 1 };

template <typename T> struct Elem;
template <> struct Elem<f16> {
  static constexpr int EPG = 8;   // elements per 16-byte k-group
  using Frag = f16x8;
};
template <> struct Elem<float> {
  static constexpr int EPG = 4;
  using Frag = f32x4;
};

// One "k-group pair" step of a 32x32 output tile.  Lane l = (r = l & 31, h = l >> 5) holds the
// 16 bytes of row/column r for k-group (2*ks + h); both operand kinds use the same mapping, so
// the sum over k covers every channel of the two k-groups exactly once.
__device__ __forceinline__ void mma32(f32x16& acc, const f16x8& a, const f16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma32(f32x16& acc, const f32x4& a, const f32x4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc, 0, 0, 0);
}

// Exact-form GELU, 0.5 x (1 + erf(x / sqrt 2)) (MONAI MLPBlock act "GELU" = nn.GELU()), with erf from Abramowitz & Stegun
// 7.1.26 (|error| <= 1.5e-7: below fp32 round-off of the surrounding arithmetic): one exp, one reciprocal, six FMAs instead
// of the ~40 instructions of erff -- the MLP's GELU runs on 21 M elements per Swin block at 48^3 tokens.
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = 1.f - p * t * __expf(-z * z);          // erf(|x| / sqrt 2)
  return 0.5f * x * (1.f + copysignf(e, x));
}

// The value the lane 32 away holds (lane ^ 32): one v_permlane32_swap on gfx950 -- __shfl_xor(v, 32) goes through
// ds_bpermute, an LDS round trip in the middle of a dependent chain.
__device__ __forceinline__ float other_half(float v) {
  const auto pr = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float((threadIdx.x & 32) ? pr[0] : pr[1]);
}

// Element offset of channel c (a multiple of 8) of voxel v inside one sample of a buffer with `stride` channels per voxel and
// `nvox` voxels: channels-last rows, or (blk) 16-channel blocks [stride / 16][nvox][16] (dua_conv3_desc.layout).
__device__ __forceinline__ long chan_off(bool blk, long v, int c, int stride, long nvox) {
  return blk ? ((long)(c >> 4) * nvox + v) * 16 + (c & 15) : v * stride + c;
}

// Row of a 32x32 MFMA accumulator held in register i (0..15) of lane-half h.
__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// Blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous run of tiles so that
// neighbouring tiles (which share halo voxels and all weights) hit the same L2.  Bijective for any n.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
  int q = n >> 3, r = n & 7, x = bid & 7, j = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

// Input transform applied while a consumer stages its operand: the producer stored the raw
// convolution output and accumulated per-(n, c) sums of it; the consumer turns the sums into
// InstanceNorm3d(affine) scale/shift in its preamble and applies
//   y = leaky_relu(x * scale[c] + shift[c]) + add[c]
// = InstanceNorm3d -> Dropout(0) -> LeakyReLU(0.1) (MONAI ADN "NDA") followed by the timestep-
// embedding bias of TwoConv.forward (models/basic_unet/denoiser.py:63-67).
constexpr int STAT_REPLICAS = 8;   // producers spread their atomics over 8 replica rows

// InstanceNorm sums are accumulated ORDER-INDEPENDENTLY: a workgroup's (sum x, sum x^2) contribution (a double, combined
// inside the workgroup in a fixed order) is split into its integer part and its fraction scaled by 2^44, and both are added
// with 64-bit INTEGER atomics -- integer addition is associative, so any arrival order of the workgroups gives the same
// words, and every consumer derives bit-identical scale / shift from them (two replays of a step agree bit for bit; fp64
// atomics differed in the last place from run to run).  Row layout: [N][8 replicas][4 words][c_pad] =
// (sum int, sum frac, sumsq int, sumsq frac).  Range: |total| < 2^62 for the integer words; resolution 2^-44 per
// contribution; the fraction words (each contribution |frac| <= 2^43) hold 2^19 contributions per replica row and the consumers
// add the 8 replica rows as int64 -- i.e. up to 2^16 contributions per row keep the 8-row sum inside int64 (a 96^3 layer at batch 4
// makes 864 per row).  Non-finite contributions: stats_add.
using stat_t = long long;
constexpr int STAT_WORDS = 4;
constexpr double STAT_FRAC = 17592186044416.0;        // 2^44

struct InXform {
  const stat_t* stats;  // [N][8][4][c_pad] fixed-point (sum x, sum x^2) replicas, or null (input already materialised)
  const float* gamma;   // [C]
  const float* beta;    // [C]
  const float* add;     // [N][add_stride] or null
  double inv_count;     // 1 / (voxels per sample) in double, from the boundary's integer count (dua_in_norm.count).  A float
                        // 1 / 884736 is off by 3e-8, and mean^2 - that much of sum x^2 / n - mean^2 is 0.3 % of the variance
                        // of a channel whose mean is 100 standard deviations (1.8e-3 on its normalised values; volumes with a
                        // power-of-two voxel count never showed it)
  int add_stride, c_pad;
  float eps, slope;
};

// (sum x, sum x^2) of channel c of sample n: the replica rows are summed as integers (exact), then converted.
__device__ __forceinline__ void stats_read(const stat_t* stats, int n, int c_pad, int c, double& S, double& Q) {
  stat_t w[STAT_WORDS] = {0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < STAT_REPLICAS; ++r) {
    const stat_t* p = stats + ((long)n * STAT_REPLICAS + r) * STAT_WORDS * c_pad + c;
#pragma unroll
    for (int k = 0; k < STAT_WORDS; ++k) w[k] += p[(long)k * c_pad];
  }
  S = (double)w[0] + (double)w[1] * (1.0 / STAT_FRAC);
  Q = (double)w[2] + (double)w[3] * (1.0 / STAT_FRAC);
}

// The same sums read by a WHOLE WAVE for 16 channels at once: lane = (part = lane >> 4, channel c0 + (lane & 15)); every lane
// loads the 4 words of TWO replica rows (8 independent 8-byte loads, all in flight together), the four parts are combined
// with cross-lane adds.  One memory round trip -- the per-thread form above, 32 loads in a register-starved kernel, was
// compiled into ~14 dependent round trips: 14 000 cycles at the head of every workgroup that normalises its input
// (in-kernel stamps, tools/stamp_conv.py).  Call with all 64 lanes active; `c` may be clamped for lanes without a channel.
__device__ __forceinline__ void stats_read_wave16(const stat_t* stats, int n, int c_pad, int c, double& S, double& Q) {
  const int part = (threadIdx.x & 63) >> 4;
  const stat_t* p = stats + ((long)n * STAT_REPLICAS + 2 * part) * STAT_WORDS * c_pad + c;
  stat_t v[2 * STAT_WORDS];
#pragma unroll
  for (int k = 0; k < 2 * STAT_WORDS; ++k) v[k] = p[(long)k * c_pad];
#ifdef DUA_STATS_F64     // diagnostic build: the words hold doubles added with fp64 atomics (the pre-round-3 arithmetic)
  S = __longlong_as_double(v[0]) + __longlong_as_double(v[STAT_WORDS]);
  Q = __longlong_as_double(v[2]) + __longlong_as_double(v[STAT_WORDS + 2]);
  S += __shfl_xor(S, 16); S += __shfl_xor(S, 32);
  Q += __shfl_xor(Q, 16); Q += __shfl_xor(Q, 32);
#else
  stat_t w[STAT_WORDS];
#pragma unroll
  for (int k = 0; k < STAT_WORDS; ++k) {
    w[k] = v[k] + v[STAT_WORDS + k];
    w[k] += __shfl_xor(w[k], 16);
    w[k] += __shfl_xor(w[k], 32);
  }
  S = (double)w[0] + (double)w[1] * (1.0 / STAT_FRAC);
  Q = (double)w[2] + (double)w[3] * (1.0 / STAT_FRAC);
#endif
}

// Preamble: the workgroup's waves compute scale/shift/add for channels [c_begin, C) into LDS arrays, 16 channels per wave
// and pass.  blockDim.x must be a multiple of 64.
__device__ __forceinline__ void xform_preamble(const InXform& xf, int n, int C, float* sc, float* sh, float* ad,
                                               int c_begin = 0) {
  if (!xf.stats) {                        // no producer statistics: the input is already materialised (identity)
    for (int c = c_begin + threadIdx.x; c < C; c += blockDim.x) {
      sc[c] = 1.f; sh[c] = 0.f;
      ad[c] = xf.add ? xf.add[(long)n * xf.add_stride + c] : 0.f;
    }
    return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  // up to UN groups of 16 channels per wave are in flight together (wide layers: 512 channels = 8 groups per wave would
  // otherwise be 8 dependent round trips to lines that the producers' atomics left outside L2)
  constexpr int UN = 4;
  for (int c0 = c_begin + wave * 16; c0 < C; c0 += UN * nw * 16) {     // wave-uniform trip count
    double S[UN], Q[UN];
    float gam[UN], bet[UN], add[UN];
    int cc[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int c = c0 + u * nw * 16 + (lane & 15);
      cc[u] = c < C ? c : C - 1;                                       // clamped address, masked store
      gam[u] = xf.gamma[cc[u]]; bet[u] = xf.beta[cc[u]];
      add[u] = xf.add ? xf.add[(long)n * xf.add_stride + cc[u]] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      S[u] = 0; Q[u] = 0;
      if (c0 + u * nw * 16 < C) stats_read_wave16(xf.stats, n, xf.c_pad, cc[u], S[u], Q[u]);     // wave-uniform condition
    }
    // every lane now holds the sums of all UN groups for channel (lane & 15): the 16-lane part u finishes group u, so the
    // double-precision reciprocal square root runs ONCE per pass, not once per group (it was ~1 000 cycles per group: the
    // preamble of a 256-channel layer was arithmetic, not the cold loads in front of it)
    const int part = lane >> 4;
    double Sm = S[0], Qm = Q[0];
    float gm = gam[0], bm = bet[0], am = add[0];
#pragma unroll
    for (int u = 1; u < UN; ++u)
      if (part == u) { Sm = S[u]; Qm = Q[u]; gm = gam[u]; bm = bet[u]; am = add[u]; }
    const int c = c0 + part * nw * 16 + (lane & 15);
    const double mean = Sm * xf.inv_count;
    double var = Qm * xf.inv_count - mean * mean;
    var = var > 0 ? var : 0;
    const float g = gm * (float)(1.0 / sqrt(var + (double)xf.eps));
    if (c0 + part * nw * 16 < C && c < C) {
      sc[c] = g;
      sh[c] = bm - (float)mean * g;
      ad[c] = am;
    }
  }
}

}  // namespace dua
#include "../../include/dua_hip.h"
namespace dua {
static inline InXform make_xform(const dua_in_norm* in, int C) {
  InXform x{};
  if (in && in->stats) {
    x.stats = in->stats; x.gamma = in->gamma; x.beta = in->beta; x.add = in->add;
    x.add_stride = in->add_stride > 0 ? in->add_stride : C;
    x.c_pad = in->c_pad; x.eps = in->eps;
    x.inv_count = in->count > 0 ? 1.0 / (double)in->count : 0.0;
  }
  x.slope = in ? in->slope : 0.f;
  return x;
}

// Epilogue side: one (sum, sum of squares) contribution per (n, c) from a workgroup.
__device__ __forceinline__ void stats_add(stat_t* stats, int n, int c_pad, int replica, int c, double S, double Q) {
  unsigned long long* p = (unsigned long long*)stats + ((long)n * STAT_REPLICAS + replica) * STAT_WORDS * c_pad + c;
#ifdef DUA_STATS_F64
  unsafeAtomicAdd((double*)p, S);
  unsafeAtomicAdd((double*)(p + 2L * c_pad), Q);
  return;
#endif
  // A non-finite or out-of-range contribution (an fp16 overflow upstream) must not turn into arbitrary finite words: it is
  // replaced by a poison value far outside anything sums of fp16 data reach (|sum x| <= 65504 * voxels < 2^41 up to 2^25 voxels),
  // so that every consumer derives an absurd mean (>= 2^47 / voxels) and a negative variance: its outputs leave the fp16 range.
  // The poison must survive being ADDED many times -- an overflow usually poisons many workgroups of a channel, and the words
  // wrap modulo 2^64: 2^47 per contribution stays below 2^62 for 2^15 poisoned contributions summed over the 8 replica rows
  // (a 96^3 layer at batch 16 makes 27 648 contributions per channel); the +-4e18 used before wrapped at the third.
  constexpr double LIM = 4.0e18;                       // < 2^62: the documented range of the fixed-point words
  constexpr double POISON = 140737488355328.0;         // 2^47
  if (!(fabs(S) < LIM) || !(fabs(Q) < LIM)) { S = POISON; Q = -POISON; }      // (the comparison is false for NaN)
  const double Si = rint(S), Qi = rint(Q);
  // two's-complement adds: negative parts wrap, the integer sum is exact either way.  System scope (sc1): the adds of all
  // eight XCDs must meet in one place; an agent-scope integer read-modify-write carries no sc bit on gfx950, and unlike the
  // fp64 adds it replaces it is not documented to execute at the memory side, so the scope is spelled out.
  __hip_atomic_fetch_add(p, (unsigned long long)(long long)Si, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_fetch_add(p + c_pad, (unsigned long long)(long long)rint((S - Si) * STAT_FRAC), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_fetch_add(p + 2L * c_pad, (unsigned long long)(long long)Qi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_fetch_add(p + 3L * c_pad, (unsigned long long)(long long)rint((Q - Qi) * STAT_FRAC), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The transform of one 16-byte fragment.  fp32 operands: the arithmetic as written above.  fp16 operands: LeakyReLU(t) + add =
// max(t + add, slope * t + add) for 0 <= slope <= 1, so with the four per-channel constants xform_prep derives
//   (sc, sh + add) and (slope * sc, slope * sh + add)
// an element is two v_fma_mix (fp16 source, fp32 arithmetic, ONE rounding into the fp16 result half) and half a v_pk_max_f16
// (rounding is monotonic: the maximum of the two rounded values is the rounded maximum) -- 2.5 instructions per element where
// the convert / fma / compare / select / add / convert sequence hipcc emits for the plain form is ~9.
template <typename T>
__device__ __forceinline__ void xform_prep(float* sc, float* sh, float* ad, float* sn, float slope) {
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int e = 0; e < Elem<T>::EPG; ++e) {
      sn[e] = slope * sc[e];
      const float b = sh[e] + ad[e];
      ad[e] = fmaf(slope, sh[e], ad[e]);
      sh[e] = b;
    }
  }
}

// one 16-byte fragment (four registers of two fp16 each); the four instructions of a register are issued a register apart, so
// that no v_fma_mixhi follows the v_fma_mixlo it merges with back to back (hipcc pads that pair with s_nop)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4_t xform_frag_mix(u32x4_t x, const float* a, const float* b, const float* an, const float* bn) {
  u32x4_t p, n, r;
#pragma unroll
  for (int j = 0; j < 4; ++j) asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(p[j]) : "v"(x[j]), "v"(a[2 * j]), "v"(b[2 * j]));
#pragma unroll
  for (int j = 0; j < 4; ++j) asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(n[j]) : "v"(x[j]), "v"(an[2 * j]), "v"(bn[2 * j]));
#pragma unroll
  for (int j = 0; j < 4; ++j)
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(p[j]) : "v"(x[j]), "v"(a[2 * j + 1]), "v"(b[2 * j + 1]));
#pragma unroll
  for (int j = 0; j < 4; ++j)
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(n[j]) : "v"(x[j]), "v"(an[2 * j + 1]), "v"(bn[2 * j + 1]));
#pragma unroll
  for (int j = 0; j < 4; ++j) asm("v_pk_max_f16 %0, %1, %2" : "=v"(r[j]) : "v"(p[j]), "v"(n[j]));
  return r;
}

// sc / sh / ad / sn as xform_prep<T> left them
template <typename T>
__device__ __forceinline__ typename Elem<T>::Frag xform_frag(typename Elem<T>::Frag v, const float* sc, const float* sh,
                                                               const float* ad, const float* sn, float slope) {
  constexpr int E = Elem<T>::EPG;
  typename Elem<T>::Frag o;
  if constexpr (sizeof(T) == 2) {
    o = __builtin_bit_cast(typename Elem<T>::Frag, xform_frag_mix(__builtin_bit_cast(u32x4_t, v), sc, sh, sn, ad));
  } else {
#pragma unroll
    for (int j = 0; j < E; ++j) {
      float y = fmaf((float)v[j], sc[j], sh[j]);
      y = y > 0.f ? y : y * slope;
      o[j] = (T)(y + ad[j]);
    }
  }
  return o;
}

}  // namespace dua
