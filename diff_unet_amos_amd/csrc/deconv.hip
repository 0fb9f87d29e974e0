// ConvTranspose3d(kernel 2, stride 2, bias) as 8 tap-GEMMs with a pixel-shuffle store.
//
// Replaces MONAI UpSample(mode="deconv") at models/basic_unet/denoiser.py:161-170 and the
// "upsampled" half of torch.cat([x_e, x_0], 1) at denoiser.py:190: the result is written
// straight into channels [Cout_off, Cout_off+Cout) of the concat buffer.
//   out[n, 2d+i, 2h+j, 2w+k, co] = bias[co] + sum_ci x[n,d,h,w,ci] * W[ci,co,i,j,k]
// GEMM view per tap: M = input voxels, N = Cout, K = Cin.  Workgroup: 256 consecutive input
// voxels x 64 output channels x one tap; wave w owns voxels [64w, 64w+64) as 2x2 MFMA 32x32
// accumulators.  The producer's InstanceNorm+LeakyReLU is applied while staging (InXform).
#include "common.hpp"
#include "../../include/dua_hip.h"

namespace dua {


namespace dc {
constexpr int TM = 256, BN = 64, KG = 4;
constexpr int VS = KG * 16 + 16;          // 80 B per voxel: conflict-free for 32 consecutive rows
constexpr int A_BYTES = TM * VS;          // 20480
constexpr int W_BYTES = KG * BN * 16;     // 4096
}  // namespace dc

struct DeconvArgs {
  const void* x; const void* w; const float* bias; void* y;
  InXform xf;
  int N, D, H, W;                  // input spatial
  int Cin, Cin_stride, Cin_off, Cout, Cout_stride, Cout_off;
  int nchunks, nct, lds_base;
  int out_blk;                     // y in 16-channel blocks (dua_conv3_desc.layout; the all-taps kernel only)
};

template <typename T>
__global__ __launch_bounds__(256) void deconv_k2s2_kernel(DeconvArgs a) {
  using namespace dc;
  using Frag = typename Elem<T>::Frag;
  constexpr int EPG = Elem<T>::EPG;
  constexpr int CK = KG * EPG;
  constexpr int OS = BN * (int)sizeof(T) + 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* alds = smem;
  char* wlds = smem + A_BYTES;
  float* xsc = (float*)(smem + a.lds_base);
  float* xsh = xsc + a.nchunks * CK;
  float* xad = xsh + a.nchunks * CK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const long vox = (long)a.D * a.H * a.W;
  const long v0 = (long)blockIdx.x * TM;
  const int tap = blockIdx.y / a.nct, ct = blockIdx.y % a.nct, n = blockIdx.z;
  const T* xin = (const T*)a.x + (long)n * vox * a.Cin_stride + a.Cin_off;
  const char* wsrc = (const char*)a.w + ((long)tap * a.nct + ct) * a.nchunks * W_BYTES;

  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][q][i] = 0.f;

  const int kg_t = tid & 3;
  if (a.xf.stats != nullptr) xform_preamble(a.xf, n, a.Cin, xsc, xsh, xad);
  // chunk ch+1 is loaded into registers while chunk ch multiplies (deep layers walk 8-16 chunks per workgroup and
  // were pure load latency without it)
  Frag pa[4];
  f32x4 pw;
  auto load_chunk = [&](int ch) {
    const int c0 = ch * CK + kg_t * EPG;
    const bool cok = c0 < a.Cin;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long v = v0 + (tid >> 2) + 64 * j;
      if (v < vox && cok) pa[j] = *(const Frag*)(xin + v * a.Cin_stride + c0);
      else
#pragma unroll
        for (int e = 0; e < EPG; ++e) pa[j][e] = (T)0.f;
    }
    pw = *(const f32x4*)(wsrc + (long)ch * W_BYTES + tid * 16);
  };
  load_chunk(0);
  for (int ch = 0; ch < a.nchunks; ++ch) {
    __syncthreads();
    const int c0 = ch * CK + kg_t * EPG;
    const bool xf = a.xf.stats != nullptr && c0 < a.Cin;
    float sc[EPG], sh[EPG], ad[EPG], sn[EPG];
    if (xf) {
#pragma unroll
      for (int e = 0; e < EPG; ++e) { sc[e] = xsc[c0 + e]; sh[e] = xsh[c0 + e]; ad[e] = xad[c0 + e]; }
      xform_prep<T>(sc, sh, ad, sn, a.xf.slope);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int vl = (tid >> 2) + 64 * j;
      Frag f = pa[j];
      if (xf && v0 + vl < vox) f = xform_frag<T>(f, sc, sh, ad, sn, a.xf.slope);
      *(Frag*)(alds + vl * VS + kg_t * 16) = f;
    }
    *(f32x4*)(wlds + tid * 16) = pw;
    __syncthreads();
    if (ch + 1 < a.nchunks) load_chunk(ch + 1);
#pragma unroll
    for (int ks = 0; ks < KG / 2; ++ks) {
      Frag a0 = *(const Frag*)(alds + (wave * 64 + r) * VS + (2 * ks + hh) * 16);
      Frag a1 = *(const Frag*)(alds + (wave * 64 + 32 + r) * VS + (2 * ks + hh) * 16);
      Frag b0 = *(const Frag*)(wlds + ((2 * ks + hh) * BN + r) * 16);
      Frag b1 = *(const Frag*)(wlds + ((2 * ks + hh) * BN + 32 + r) * 16);
      mma32(acc[0][0], a0, b0);
      mma32(acc[0][1], a0, b1);
      mma32(acc[1][0], a1, b0);
      mma32(acc[1][1], a1, b1);
    }
  }
  __syncthreads();
  char* ot = smem + wave * 64 * OS;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int co = q * 32 + r;
    const float bq = a.bias[ct * BN + co];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        *(T*)(ot + (m * 32 + acc_row(i, hh)) * OS + co * (int)sizeof(T)) = (T)(acc[m][q][i] + bq);
  }
  __syncthreads();
  constexpr int GPV = BN / EPG, VPI = 64 / GPV;
  const int ti = tap >> 2, tj = (tap >> 1) & 1, tk = tap & 1;
  const int H2 = 2 * a.H, W2 = 2 * a.W;
  T* yout = (T*)a.y + (long)n * vox * 8 * a.Cout_stride + a.Cout_off + ct * BN;
#pragma unroll
  for (int it = 0; it < 64 / VPI; ++it) {
    const int vl = it * VPI + lane / GPV, cg = lane % GPV;
    const long v = v0 + wave * 64 + vl;
    if (v < vox && ct * BN + cg * EPG < a.Cout) {
      const int w = (int)(v % a.W); const long t = v / a.W;
      const int h = (int)(t % a.H), d = (int)(t / a.H);
      const long ov = ((long)(2 * d + ti) * H2 + (2 * h + tj)) * W2 + (2 * w + tk);
      *(Frag*)(yout + ov * a.Cout_stride + cg * EPG) = *(const Frag*)(ot + vl * OS + cg * 16);
    }
  }
}

// Small inputs (<= 24^3: few voxels, many input channels): the kernel above gives a workgroup 256 voxels and walks the Cin
// chunks one after the other between two barriers each -- at 6^3 that is 32 workgroups x 16 chunks of ~2 000 cycles of staging
// and transform arithmetic around 8 MFMAs.  Here a workgroup owns 64 voxels x 64 output channels x one tap and its four
// waves SPLIT the Cin chunks (wave w: chunks w, w + 4, ...): every wave requests all its chunks at once (at most MC = 4:
// Cin <= 512), stages each in rows of its own (no workgroup barrier in the K loop; ds operations of one wave execute in
// order), and the four partial tiles are added in a fixed order through LDS (deterministic) before the pixel-shuffle store.
// Four times the workgroups, a quarter of the per-thread staging work, one memory round trip.
namespace dcs {
constexpr int TMS = 64, MC = 4;
constexpr int STAGE = TMS * dc::VS + dc::W_BYTES;            // 5120 + 4096 per wave
constexpr int RS = dc::BN * 4 + 16;                          // fp32 partial row: 272 B
constexpr int RED = 4 * TMS * RS;                            // 69632: the staging rows live inside it
}  // namespace dcs

template <typename T>
__global__ __launch_bounds__(256, 2) void deconv_k2s2_ksplit_kernel(DeconvArgs a) {
  using namespace dc;
  using namespace dcs;
  using Frag = typename Elem<T>::Frag;
  constexpr int EPG = Elem<T>::EPG;
  constexpr int CK = KG * EPG;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xsc = (float*)(smem + RED);
  float* xsh = xsc + a.nchunks * CK;
  float* xad = xsh + a.nchunks * CK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const long vox = (long)a.D * a.H * a.W;
  const long v0 = (long)blockIdx.x * TMS;
  const int tap = blockIdx.y / a.nct, ct = blockIdx.y % a.nct, n = blockIdx.z;
  const T* xin = (const T*)a.x + (long)n * vox * a.Cin_stride + a.Cin_off;
  const char* wsrc = (const char*)a.w + ((long)tap * a.nct + ct) * a.nchunks * W_BYTES;
  char* alds = smem + wave * STAGE;
  char* wlds = alds + TMS * VS;
  const int kg_t = lane & 3;
  const bool fused = a.xf.stats != nullptr;

  // every chunk of this wave is requested before anything waits
  Frag pa[MC][4];
  f32x4 pw[MC][4];
#pragma unroll
  for (int c = 0; c < MC; ++c) {
    const int ch = wave + 4 * c;
    const int c0 = ch * CK + kg_t * EPG;
    const bool cok = ch < a.nchunks && c0 < a.Cin;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long v = v0 + (lane >> 2) + 16 * j;
      pa[c][j] = *(const Frag*)(xin + (v < vox && cok ? v * a.Cin_stride + c0 : 0));
      pw[c][j] = *(const f32x4*)(wsrc + (ch < a.nchunks ? (long)ch * W_BYTES + (lane + 64 * j) * 16 : 0));
    }
  }
  if (fused) xform_preamble(a.xf, n, a.Cin, xsc, xsh, xad);
  __syncthreads();

  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][q][i] = 0.f;
#pragma unroll
  for (int c = 0; c < MC; ++c) {
    const int ch = wave + 4 * c;
    if (ch >= a.nchunks) break;                              // wave-uniform
    const int c0 = ch * CK + kg_t * EPG;
    const bool cok = c0 < a.Cin;
    float sc[EPG], sh[EPG], ad[EPG], sn[EPG];
    if (fused && cok) {
#pragma unroll
      for (int e = 0; e < EPG; ++e) { sc[e] = xsc[c0 + e]; sh[e] = xsh[c0 + e]; ad[e] = xad[c0 + e]; }
      xform_prep<T>(sc, sh, ad, sn, a.xf.slope);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int vl = (lane >> 2) + 16 * j;
      const bool ok = v0 + vl < vox && cok;
      Frag f = pa[c][j];
      if (fused && cok) f = xform_frag<T>(f, sc, sh, ad, sn, a.xf.slope);
#pragma unroll
      for (int e = 0; e < EPG; ++e) f[e] = ok ? f[e] : (T)0.f;
      *(Frag*)(alds + vl * VS + kg_t * 16) = f;
      *(f32x4*)(wlds + (lane + 64 * j) * 16) = pw[c][j];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int ks = 0; ks < KG / 2; ++ks) {
      const Frag a0 = *(const Frag*)(alds + r * VS + (2 * ks + hh) * 16);
      const Frag a1 = *(const Frag*)(alds + (32 + r) * VS + (2 * ks + hh) * 16);
      const Frag b0 = *(const Frag*)(wlds + ((2 * ks + hh) * BN + r) * 16);
      const Frag b1 = *(const Frag*)(wlds + ((2 * ks + hh) * BN + 32 + r) * 16);
      mma32(acc[0][0], a0, b0);
      mma32(acc[0][1], a0, b1);
      mma32(acc[1][0], a1, b0);
      mma32(acc[1][1], a1, b1);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  __syncthreads();                                           // the partial tiles overwrite everybody's staging rows
  float* red = (float*)(smem + wave * TMS * RS);
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) red[(m * 32 + acc_row(i, hh)) * (RS / 4) + q * 32 + r] = acc[m][q][i];
  __syncthreads();
  // thread = (voxel row, 8 output channels): partials of waves 0..3 in that order, + bias, pixel-shuffle store
  const int ti = tap >> 2, tj = (tap >> 1) & 1, tk = tap & 1;
  const int H2 = 2 * a.H, W2 = 2 * a.W;
  T* yout = (T*)a.y + (long)n * vox * 8 * a.Cout_stride + a.Cout_off + ct * BN;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int item = tid + 256 * it, vl = item >> 3, cg = item & 7;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float* p = (const float*)(smem + w * TMS * RS) + vl * (RS / 4) + cg * 8;
      const f32x4 lo = *(const f32x4*)p, hi = *(const f32x4*)(p + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] += lo[e]; o[4 + e] += hi[e]; }
    }
    const long v = v0 + vl;
    if (v < vox && ct * BN + cg * 8 < a.Cout) {
      const int w = (int)(v % a.W); const long t = v / a.W;
      const int h = (int)(t % a.H), d = (int)(t / a.H);
      const long ov = ((long)(2 * d + ti) * H2 + (2 * h + tj)) * W2 + (2 * w + tk);
      T* dst = yout + ov * a.Cout_stride + cg * 8;
      if constexpr (sizeof(T) == 2) {
        Frag f;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (T)(o[e] + a.bias[ct * BN + cg * 8 + e]);
        *(Frag*)dst = f;
      } else {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          Frag f;
#pragma unroll
          for (int e = 0; e < 4; ++e) f[e] = (T)(o[4 * g + e] + a.bias[ct * BN + cg * 8 + 4 * g + e]);
          *(Frag*)(dst + 4 * g) = f;
        }
      }
    }
  }
}

// All eight taps in one workgroup (large inputs): the input tile is staged once with its full channel depth (the one-tap
// kernel above re-reads and re-normalises it for every tap), then the workgroup walks the taps in PAIRS (tk = 0, 1): the
// weights of the next pair are prefetched while this pair multiplies, and a pair ends with one pixel-shuffle store of both
// taps -- outputs 2w and 2w + 1 lie next to each other, so a store instruction writes whole runs (in the 16-channel-block
// layout: 512 contiguous bytes per block) where one tap alone writes every other 32- or 64-byte piece.
template <typename T, int MBLK>
__global__ __launch_bounds__(256) void deconv_k2s2_alltaps_kernel(DeconvArgs a) {
  using namespace dc;
  constexpr int TM = 64 * MBLK * 2, WR = 32 * MBLK;          // voxels per workgroup / per wave (MBLK 32-row MFMA blocks each)
  using Frag = typename Elem<T>::Frag;
  constexpr int EPG = Elem<T>::EPG;
  constexpr int CK = KG * EPG;
  constexpr int RB = 32 * (int)sizeof(T) + 16;              // staging row: one 32-channel half
  constexpr int GPV = 32 / EPG, VPI = 64 / GPV, NST = 2 * WR / VPI;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int VSA = a.nchunks * 64 + 16;                      // bytes per staged voxel (odd multiple of 16: conflict-free)
  char* alds = smem;
  char* wlds = alds + TM * VSA;                             // 2 pairs x 2 taps x nchunks x 4 KB
  char* stg = wlds + 4 * a.nchunks * W_BYTES;               // 2 TM x RB
  float* xsc = (float*)(stg + 2 * TM * RB);
  float* xsh = xsc + a.nchunks * CK;
  float* xad = xsh + a.nchunks * CK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const long vox = (long)a.D * a.H * a.W;
  const long v0 = (long)blockIdx.x * TM;
  const int ct = blockIdx.y, n = blockIdx.z;
  const T* xin = (const T*)a.x + (long)n * vox * a.Cin_stride + a.Cin_off;
  const int wtap = a.nchunks * W_BYTES;                     // bytes of one tap's weights for this cout tile
  const char* wsrc = (const char*)a.w + (long)ct * wtap;    // tap t: + t * nct * wtap
  const int kg_t = tid & 3;

  // ---- the whole input tile (every chunk: nchunks <= 4) and the weights of taps 0, 1 are requested before anything waits:
  // one memory round trip in front of the first MFMA instead of one per chunk plus the statistics preamble's ----
  constexpr int MCH = 4;
  Frag f[MCH][TM / 64];
#pragma unroll
  for (int ch = 0; ch < MCH; ++ch) {
    const int c0 = ch * CK + kg_t * EPG;
    const bool cok = ch < a.nchunks && c0 < a.Cin;
#pragma unroll
    for (int j = 0; j < TM / 64; ++j) {
      const long v = v0 + (tid >> 2) + 64 * j;
      f[ch][j] = *(const Frag*)(xin + (v < vox && cok ? v * a.Cin_stride + c0 : 0));
    }
  }
  f32x4 wn[2][MCH];
  auto load_pair = [&](int tp) {
#pragma unroll
    for (int tk = 0; tk < 2; ++tk)
#pragma unroll
      for (int j = 0; j < MCH; ++j)
        wn[tk][j] = *(const f32x4*)(wsrc + (long)(2 * tp + tk) * a.nct * wtap + (j < a.nchunks ? (tid + 256 * j) * 16 : 0));
  };
  auto store_pair = [&](int tp) {
#pragma unroll
    for (int tk = 0; tk < 2; ++tk)
#pragma unroll
      for (int j = 0; j < MCH; ++j)
        if (j < a.nchunks) *(f32x4*)(wlds + ((tp & 1) * 2 + tk) * wtap + (tid + 256 * j) * 16) = wn[tk][j];
  };
  load_pair(0);
  if (a.xf.stats != nullptr) {
    xform_preamble(a.xf, n, a.Cin, xsc, xsh, xad);
    __syncthreads();
  }
#pragma unroll
  for (int ch = 0; ch < MCH; ++ch) {
    if (ch >= a.nchunks) break;
    const int c0 = ch * CK + kg_t * EPG;
    const bool cok = c0 < a.Cin;
    const bool xf = a.xf.stats != nullptr && cok;
    float sc[EPG], sh[EPG], ad[EPG], sn[EPG];
    if (xf) {
#pragma unroll
      for (int e = 0; e < EPG; ++e) { sc[e] = xsc[c0 + e]; sh[e] = xsh[c0 + e]; ad[e] = xad[c0 + e]; }
      xform_prep<T>(sc, sh, ad, sn, a.xf.slope);
    }
#pragma unroll
    for (int j = 0; j < TM / 64; ++j) {
      const int vl = (tid >> 2) + 64 * j;
      const bool ok = v0 + vl < vox && cok;
      Frag g = f[ch][j];
      if (xf) g = xform_frag<T>(g, sc, sh, ad, sn, a.xf.slope);
#pragma unroll
      for (int e = 0; e < EPG; ++e) g[e] = ok ? g[e] : (T)0.f;
      *(Frag*)(alds + vl * VSA + ch * 64 + kg_t * 16) = g;
    }
  }
  store_pair(0);
  __syncthreads();

  const int H2 = 2 * a.H, W2 = 2 * a.W;
  T* yout = (T*)a.y + (long)n * vox * 8 * a.Cout_stride;
  const float bq0 = a.bias[ct * BN + r], bq1 = a.bias[ct * BN + 32 + r];
  char* ot = stg + wave * 2 * WR * RB;                      // staged row 2 vl + tk: the two taps of a pair interleaved like their outputs
  int ovb[NST];                                             // output voxel (before the pair's (ti, tj) offset) of this lane's staged rows, or -1
#pragma unroll
  for (int it = 0; it < NST; ++it) {
    const int j = it * VPI + lane / GPV;
    const long v = v0 + wave * WR + (j >> 1);
    const int vi = (int)(v < vox ? v : 0);
    const int w = vi % a.W, t = vi / a.W, h = t % a.H, d = t / a.H;
    ovb[it] = v < vox ? ((2 * d) * H2 + 2 * h) * W2 + 2 * w + (j & 1) : -1;
  }
  // The weights of pair p + 1 travel in registers while pair p multiplies.  They are REQUESTED before the stores of pair p
  // are issued, not after: vmcnt counts loads and stores in issue order, so a wait for loads issued behind the stores is a
  // wait for those stores to be acknowledged by memory.
  load_pair(1);
  for (int tp = 0; tp < 4; ++tp) {
    f32x16 acc[2][MBLK][2];
#pragma unroll
    for (int tk = 0; tk < 2; ++tk) {
#pragma unroll
      for (int m = 0; m < MBLK; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[tk][m][0][i] = bq0; acc[tk][m][1][i] = bq1; }
      const char* wb = wlds + ((tp & 1) * 2 + tk) * wtap;
      for (int ch = 0; ch < a.nchunks; ++ch) {
#pragma unroll
        for (int ks = 0; ks < KG / 2; ++ks) {
          const Frag b0 = *(const Frag*)(wb + ch * W_BYTES + ((2 * ks + hh) * BN + r) * 16);
          const Frag b1 = *(const Frag*)(wb + ch * W_BYTES + ((2 * ks + hh) * BN + 32 + r) * 16);
#pragma unroll
          for (int m = 0; m < MBLK; ++m) {
            const Frag am = *(const Frag*)(alds + (wave * WR + m * 32 + r) * VSA + ch * 64 + (2 * ks + hh) * 16);
            mma32(acc[tk][m][0], am, b0);
            mma32(acc[tk][m][1], am, b1);
          }
        }
      }
    }
    if (tp < 3) store_pair(tp + 1);
    if (tp < 2) load_pair(tp + 2);
    const int toff = ((tp >> 1) * H2 + (tp & 1)) * W2;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int tk = 0; tk < 2; ++tk)
#pragma unroll
        for (int m = 0; m < MBLK; ++m)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            *(T*)(ot + (2 * (m * 32 + acc_row(i, hh)) + tk) * RB + r * (int)sizeof(T)) = (T)acc[tk][m][q][i];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < NST; ++it) {
        const int j = it * VPI + lane / GPV, cg = lane % GPV;
        if (ovb[it] >= 0 && ct * BN + q * 32 + cg * EPG < a.Cout)
          *(Frag*)(yout + chan_off(a.out_blk, ovb[it] + toff, a.Cout_off + ct * BN + q * 32 + cg * EPG, a.Cout_stride, vox * 8)) =
              *(const Frag*)(ot + j * RB + cg * 16);
      }
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();      // the next pair's weights are in place, this pair's buffer is free
  }
}

// ---- The 1x1x1 residual branch of a UnetResBlock over cat((ConvTranspose3d_k2s2(lo), skip)) in ONE launch (config 5, round 5) ----
//   res[o] = W3_up . up[o] + W3_skip . skip[o],   up[2p + child] = Wd[child]^T lo[p]   (no bias in either layer: MONAI
//   UnetrUpBlock / UnetResBlock as models/swin_unetr/denoiser.py:388-397 builds them)
//          = (W3_up Wd[child]^T) lo[parent(o)] + W3_skip skip[o]
// i.e. a transposed convolution with COMPOSED weights plus a pointwise term on the fine grid.  Once the block's 3x3x3 convolution
// takes its upsampled half from lo directly (upconv.hip), nobody else reads `up`: this launch replaces the transposed
// convolution (its 85 MB output write at 96^3) AND the token GEMM over the 96-channel concat (its re-read).  The all-taps
// kernel above with, per tap pair, NSK more 32-channel k-chunks against the skip weights whose A fragments every lane loads
// straight from the fine-grid tensor (its own row: one 16-byte load per k-step and tap, requested a pair ahead); the layer's InstanceNorm sums come out of the accumulators.  fp16, 128-voxel tiles, Cin <= 128.
struct DeconvResArgs {
  DeconvArgs base;
  const void* xs; const void* ws; stat_t* stats;
  int Cs, Cs_stride, Cs_off, cout_pad;
};

template <int NSK>
__global__ __launch_bounds__(256, 2) void deconv_k2s2_res_kernel(DeconvResArgs ra) {
  using namespace dc;
  using T = f16;
  const DeconvArgs& a = ra.base;
  constexpr int TM = 128, WR = 32, EPG = 8;
  using Frag = f16x8;
  constexpr int RB = 32 * 2 + 16;
  constexpr int GPV = 4, VPI = 16, NST = 2 * WR / VPI;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int VSA = a.nchunks * 64 + 16;
  char* alds = smem;
  char* wlds = alds + TM * VSA;                             // 2 pairs x 2 taps x nchunks x 4 KB
  char* stg = wlds + 4 * a.nchunks * W_BYTES;               // 2 TM x RB
  char* skw = stg + 2 * TM * RB;                            // NSK x 4 KB: the skip weights

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const long vox = (long)a.D * a.H * a.W;
  const long v0 = (long)blockIdx.x * TM;
  const int ct = blockIdx.y, n = blockIdx.z;
  const T* xin = (const T*)a.x + (long)n * vox * a.Cin_stride + a.Cin_off;
  const int wtap = a.nchunks * W_BYTES;
  const char* wsrc = (const char*)a.w + (long)ct * wtap;
  const int kg_t = tid & 3;
  const int H2 = 2 * a.H, W2 = 2 * a.W;
  const T* xsk = (const T*)ra.xs + (long)n * vox * 8 * ra.Cs_stride + ra.Cs_off;

  constexpr int MCH = 4;
  Frag f[MCH][TM / 64];
#pragma unroll
  for (int ch = 0; ch < MCH; ++ch) {
    const int c0 = ch * 32 + kg_t * EPG;
    const bool cok = ch < a.nchunks && c0 < a.Cin;
#pragma unroll
    for (int j = 0; j < TM / 64; ++j) {
      const long v = v0 + (tid >> 2) + 64 * j;
      f[ch][j] = *(const Frag*)(xin + (v < vox && cok ? v * a.Cin_stride + c0 : 0));
    }
  }
  f32x4 wn[2][MCH];
  auto load_pair = [&](int tp) {
#pragma unroll
    for (int tk = 0; tk < 2; ++tk)
#pragma unroll
      for (int j = 0; j < MCH; ++j)
        wn[tk][j] = *(const f32x4*)(wsrc + (long)(2 * tp + tk) * a.nct * wtap + (j < a.nchunks ? (tid + 256 * j) * 16 : 0));
  };
  auto store_pair = [&](int tp) {
#pragma unroll
    for (int tk = 0; tk < 2; ++tk)
#pragma unroll
      for (int j = 0; j < MCH; ++j)
        if (j < a.nchunks) *(f32x4*)(wlds + ((tp & 1) * 2 + tk) * wtap + (tid + 256 * j) * 16) = wn[tk][j];
  };
  // The skip operand needs no staging: an A fragment of the pointwise term is 8 channels of ONE fine voxel per lane -- the lane's own
  // row of the pair (tile voxel wave * 32 + r, tap (ti, tj, tk)) -- i.e. one 16-byte global load per lane, k-step and tap; the
  // fragments of pair tp + 1 are requested before pair tp's stores are issued.  (Staged through LDS like the coarse tile they cost
  // 37 KB and left one workgroup per CU: 72 us at 96^3.)
  const long vrow = v0 + wave * WR + r;
  const bool vrok = vrow < vox;
  long ovrow;
  {
    const int vi = (int)(vrok ? vrow : 0);
    const int w = vi % a.W, t = vi / a.W, h = t % a.H, d = t / a.H;
    ovrow = ((long)(2 * d) * H2 + 2 * h) * W2 + 2 * w;
  }
  Frag skf[2][2 * NSK];
  auto load_skip = [&](int tp) {
    const long base = ovrow + ((long)(tp >> 1) * H2 + (tp & 1)) * W2;
#pragma unroll
    for (int tk = 0; tk < 2; ++tk)
#pragma unroll
      for (int j = 0; j < 2 * NSK; ++j) {                       // j = (chunk, k-step): channels 32 (j >> 1) + 16 (j & 1) + 8 hh ...
        const int c0 = (j >> 1) * 32 + (2 * (j & 1) + hh) * 8;
        const bool ok = vrok && c0 < ra.Cs;
        skf[tk][j] = *(const Frag*)(xsk + (ok ? (base + tk) * ra.Cs_stride + c0 : 0));
        if (!ok) {
#pragma unroll
          for (int e = 0; e < 8; ++e) skf[tk][j][e] = (T)0.f;
        }
      }
  };
  load_pair(0);
  load_skip(0);
  for (int i = tid; i < NSK * W_BYTES / 16; i += 256) *(f32x4*)(skw + i * 16) = *(const f32x4*)((const char*)ra.ws + (long)ct * NSK * W_BYTES + i * 16);
#pragma unroll
  for (int ch = 0; ch < MCH; ++ch) {
    if (ch >= a.nchunks) break;
    const int c0 = ch * 32 + kg_t * EPG;
    const bool cok = c0 < a.Cin;
#pragma unroll
    for (int j = 0; j < TM / 64; ++j) {
      const int vl = (tid >> 2) + 64 * j;
      const bool ok = v0 + vl < vox && cok;
      Frag g = f[ch][j];
#pragma unroll
      for (int e = 0; e < EPG; ++e) g[e] = ok ? g[e] : (T)0.f;
      *(Frag*)(alds + vl * VSA + ch * 64 + kg_t * 16) = g;
    }
  }
  store_pair(0);
  __syncthreads();

  T* yout = (T*)a.y + (long)n * vox * 8 * a.Cout_stride;
  char* ot = stg + wave * 2 * WR * RB;
  int ovb[NST];
#pragma unroll
  for (int it = 0; it < NST; ++it) {
    const int j = it * VPI + lane / GPV;
    const long v = v0 + wave * WR + (j >> 1);
    const int vi = (int)(v < vox ? v : 0);
    const int w = vi % a.W, t = vi / a.W, h = t % a.H, d = t / a.H;
    ovb[it] = v < vox ? ((2 * d) * H2 + 2 * h) * W2 + 2 * w + (j & 1) : -1;
  }
  float s[2] = {0.f, 0.f}, ss[2] = {0.f, 0.f};
  load_pair(1);
  for (int tp = 0; tp < 4; ++tp) {
    f32x16 acc[2][2];
#pragma unroll
    for (int tk = 0; tk < 2; ++tk) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc[tk][0][i] = 0.f; acc[tk][1][i] = 0.f; }
      const char* wb = wlds + ((tp & 1) * 2 + tk) * wtap;
      for (int ch = 0; ch < a.nchunks; ++ch) {
#pragma unroll
        for (int ks = 0; ks < KG / 2; ++ks) {
          const Frag b0 = *(const Frag*)(wb + ch * W_BYTES + ((2 * ks + hh) * BN + r) * 16);
          const Frag b1 = *(const Frag*)(wb + ch * W_BYTES + ((2 * ks + hh) * BN + 32 + r) * 16);
          const Frag am = *(const Frag*)(alds + (wave * WR + r) * VSA + ch * 64 + (2 * ks + hh) * 16);
          mma32(acc[tk][0], am, b0);
          mma32(acc[tk][1], am, b1);
        }
      }
#pragma unroll
      for (int j = 0; j < 2 * NSK; ++j) {
        const Frag b0 = *(const Frag*)(skw + (j >> 1) * W_BYTES + ((2 * (j & 1) + hh) * BN + r) * 16);
        const Frag b1 = *(const Frag*)(skw + (j >> 1) * W_BYTES + ((2 * (j & 1) + hh) * BN + 32 + r) * 16);
        mma32(acc[tk][0], skf[tk][j], b0);
        mma32(acc[tk][1], skf[tk][j], b1);
      }
    }
    if (tp < 3) load_skip(tp + 1);                            // ahead of this pair's stores (vmcnt is in issue order)
    if (tp < 3) store_pair(tp + 1);
    if (tp < 2) load_pair(tp + 2);
    const int toff = ((tp >> 1) * H2 + (tp & 1)) * W2;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int tk = 0; tk < 2; ++tk)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float v = acc[tk][q][i];
          s[q] += v;
          ss[q] = fmaf(v, v, ss[q]);
          *(T*)(ot + (2 * acc_row(i, hh) + tk) * RB + r * 2) = (T)v;
        }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < NST; ++it) {
        const int j = it * VPI + lane / GPV, cg = lane % GPV;
        if (ovb[it] >= 0 && ct * BN + q * 32 + cg * EPG < a.Cout)
          *(Frag*)(yout + (long)(ovb[it] + toff) * a.Cout_stride + a.Cout_off + ct * BN + q * 32 + cg * EPG) = *(const Frag*)(ot + j * RB + cg * 16);
      }
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();      // the next pair's weights are in place, this pair's buffer is free
  }
  // ---- InstanceNorm sums of this layer: lane halves, waves (fixed order), one set of atomics ----
  float* ex = (float*)stg;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    s[q] += __shfl_xor(s[q], 32);
    ss[q] += __shfl_xor(ss[q], 32);
    if (hh == 0) { ex[(wave * BN + q * 32 + r) * 2] = s[q]; ex[(wave * BN + q * 32 + r) * 2 + 1] = ss[q]; }
  }
  __syncthreads();
  if (wave == 0) {
    double S = 0, Q = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { S += (double)ex[(w * BN + lane) * 2]; Q += (double)ex[(w * BN + lane) * 2 + 1]; }
    if (ct * BN + lane < a.Cout) stats_add(ra.stats, n, ra.cout_pad, blockIdx.x & (STAT_REPLICAS - 1), ct * BN + lane, S, Q);
  }
}

static inline int deconv_res_lds(int nchunks, int nsk) {
  return 128 * (nchunks * 64 + 16) + 4 * nchunks * dc::W_BYTES + 2 * 128 * 80 + nsk * dc::W_BYTES;
}

// dynamic-LDS limits (raised once per device by ensure_prepared(), common.hpp)
template <typename T> constexpr int deconv_plain_lds() {
  constexpr int OS = dc::BN * (int)sizeof(T) + 16;
  return ((dc::TM * OS > dc::A_BYTES + dc::W_BYTES) ? dc::TM * OS : dc::A_BYTES + dc::W_BYTES) + 3 * 4 * 1024;
}
static const LdsAttr kDeconvLdsAttrs[] = {
    {(const void*)deconv_k2s2_alltaps_kernel<f16, 2>, 160 * 1024},   {(const void*)deconv_k2s2_alltaps_kernel<f16, 1>, 160 * 1024},
    {(const void*)deconv_k2s2_alltaps_kernel<float, 2>, 160 * 1024}, {(const void*)deconv_k2s2_alltaps_kernel<float, 1>, 160 * 1024},
    {(const void*)deconv_k2s2_ksplit_kernel<f16>, dcs::RED + 3 * 4 * 1024},
    {(const void*)deconv_k2s2_ksplit_kernel<float>, dcs::RED + 3 * 4 * 1024},
    {(const void*)deconv_k2s2_kernel<f16>, deconv_plain_lds<f16>()},
    {(const void*)deconv_k2s2_kernel<float>, deconv_plain_lds<float>()},
    {(const void*)deconv_k2s2_res_kernel<1>, 160 * 1024}, {(const void*)deconv_k2s2_res_kernel<2>, 160 * 1024},
    {(const void*)deconv_k2s2_res_kernel<3>, 160 * 1024}, {(const void*)deconv_k2s2_res_kernel<4>, 160 * 1024},
};
static const LdsAttrs kDeconvLdsReg(kDeconvLdsAttrs);

// 2 = the all-taps kernel (large inputs: it may write 16-channel blocks), 1 = Cin chunks split over the waves, 0 = one tap per workgroup
static int deconv_kernel_kind(const dua_conv3_desc* d) {
  const int ck = dc::KG * (d->dtype == DUA_F16 ? 8 : 4);
  const int nchunks = (d->Cin + ck - 1) / ck;
  const long vox = (long)d->D * d->H * d->W;
  if (vox >= 256L * 128 && nchunks <= 4) return 2;
  if (nchunks >= 8 && nchunks <= 4 * dcs::MC && d->policy != 6) return 1;
  return 0;
}

template <typename T>
static int launch_deconv(const dua_conv3_desc* d, const void* x, const void* w, const float* bias,
                         const dua_in_norm* in, void* y, hipStream_t s) {
  constexpr int CK = dc::KG * Elem<T>::EPG;
  DeconvArgs a;
  a.x = x; a.w = w; a.bias = bias; a.y = y;
  a.xf = make_xform(in, d->Cin);
  a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W;
  a.Cin = d->Cin; a.Cin_stride = d->Cin_stride; a.Cin_off = d->Cin_off;
  a.Cout = d->Cout; a.Cout_stride = d->Cout_stride; a.Cout_off = d->Cout_off;
  a.nchunks = (d->Cin + CK - 1) / CK;
  a.nct = (d->Cout + dc::BN - 1) / dc::BN;
  const long vox = (long)d->D * d->H * d->W;
  if (int e = ensure_prepared()) return e;
  if (d->policy != 0 && d->policy != 6) return DUA_ERR_ARG;
  const int g_conv_variant = d->policy;
  a.out_blk = d->layout & DUA_OUT_BLOCKED ? 1 : 0;
  if (d->layout & DUA_IN_BLOCKED) return DUA_ERR_ARG;
  if (a.out_blk && (deconv_kernel_kind(d) != 2 || d->Cout_off % 16 || d->Cout_stride % 16 || vox * 8 * 16 >= 0x7fffffffL)) return DUA_ERR_ARG;
  if (vox >= 256L * 128 && a.nchunks <= 4) {          // enough tiles to fill the chip with one workgroup per 8 taps
    // 128-voxel tiles: two workgroups per CU up to 64 input channels (70 KB each), one's pixel-shuffle stores under the
    // other's loads; policy 6 keeps the 256-voxel form (one workgroup per CU) for A/B where it fits
    auto lds_of = [&](int tm) {
      return tm * (a.nchunks * 64 + 16) + 4 * a.nchunks * dc::W_BYTES + 2 * tm * (32 * (int)sizeof(T) + 16) +
             (a.xf.stats ? 3 * 4 * a.nchunks * CK : 0);
    };
    const int mblk = g_conv_variant == 6 && lds_of(256) <= 160 * 1024 ? 2 : 1;
    const int tm = 128 * mblk;
    const int lds = lds_of(tm);
    if (lds > 160 * 1024) return DUA_ERR_ARG;
    dim3 grid2((unsigned)((vox + tm - 1) / tm), a.nct, d->N);
    if (mblk == 2) hipLaunchKernelGGL((deconv_k2s2_alltaps_kernel<T, 2>), grid2, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((deconv_k2s2_alltaps_kernel<T, 1>), grid2, dim3(256), lds, s, a);
    return (int)hipGetLastError();
  }
  if (a.nchunks >= 8 && a.nchunks <= 4 * dcs::MC && g_conv_variant != 6) {   // Cin >= 256: waves split the Cin chunks (variant 6: the one-chunk-at-a-time kernel, A/B)
    const int lds = dcs::RED + (a.xf.stats ? 3 * 4 * a.nchunks * CK : 0);
    dim3 grid3((unsigned)((vox + dcs::TMS - 1) / dcs::TMS), 8 * a.nct, d->N);
    hipLaunchKernelGGL(deconv_k2s2_ksplit_kernel<T>, grid3, dim3(256), lds, s, a);
    return (int)hipGetLastError();
  }
  dim3 grid((unsigned)((vox + dc::TM - 1) / dc::TM), 8 * a.nct, d->N);
  constexpr int OS = dc::BN * (int)sizeof(T) + 16;
  constexpr int LDS = (dc::TM * OS > dc::A_BYTES + dc::W_BYTES) ? dc::TM * OS : dc::A_BYTES + dc::W_BYTES;
  a.lds_base = LDS;
  if (a.nchunks * CK > 1024) return DUA_ERR_ARG;
  hipLaunchKernelGGL(deconv_k2s2_kernel<T>, grid, dim3(256), LDS + (a.xf.stats ? 3 * 4 * a.nchunks * CK : 0), s, a);
  return (int)hipGetLastError();
}

}  // namespace dua

extern "C" int dua_deconv_k2s2_kernel_kind(const dua_conv3_desc* d) {
  if (!d || (d->dtype != DUA_F16 && d->dtype != DUA_F32)) return DUA_ERR_ARG;
  return dua::deconv_kernel_kind(d);
}

extern "C" int dua_deconv_k2s2_fwd(const dua_conv3_desc* d, const void* x, const void* w_packed,
                                   const float* bias_padded, const dua_in_norm* in, void* y, void* stream) {
  if (!d || !x || !w_packed || !bias_padded || !y) return DUA_ERR_ARG;
  if (in && in->stats && (!in->gamma || !in->beta || in->c_pad < d->Cin || in->count <= 0 || !(in->slope >= 0.f && in->slope <= 1.f))) return DUA_ERR_ARG;
  if (d->Cin % 8 || d->Cout % 8 || d->Cin_stride % 8 || d->Cout_stride % 8 || d->Cin_off % 8 || d->Cout_off % 8)
    return DUA_ERR_ARG;
  if (d->dtype == DUA_F16) return dua::launch_deconv<dua::f16>(d, x, w_packed, bias_padded, in, y, (hipStream_t)stream);
  if (d->dtype == DUA_F32) return dua::launch_deconv<float>(d, x, w_packed, bias_padded, in, y, (hipStream_t)stream);
  return DUA_ERR_ARG;
}

/* see include/dua_hip.h */
extern "C" int dua_deconv_k2s2_res_supported(const dua_conv3_desc* d, int Cs) {
  if (!d || d->dtype != DUA_F16 || d->N <= 0 || d->D <= 0 || d->H <= 0 || d->W <= 0) return 0;
  if (d->Cin <= 0 || d->Cin % 8 || d->Cin > 128 || d->Cin_stride % 8 || d->Cin_off % 8 || d->Cin_off + d->Cin > d->Cin_stride) return 0;
  if (d->Cout <= 0 || d->Cout % 8 || d->Cout_stride % 8 || d->Cout_off % 8 || d->Cout_off + d->Cout > d->Cout_stride) return 0;
  if (Cs <= 0 || Cs % 8 || Cs > 128 || d->layout || d->policy) return 0;
  const long vox = (long)d->D * d->H * d->W;
  if (vox * 8 * (d->Cout_stride > 256 ? d->Cout_stride : 256) >= 0x7fffffffL) return 0;
  return dua::deconv_res_lds((d->Cin + 31) / 32, (Cs + 31) / 32) <= 160 * 1024 ? 1 : 0;
}

extern "C" int dua_deconv_k2s2_res_fwd(const dua_conv3_desc* d, const void* lo, const void* w_packed, const void* xskip, int Cs,
                                       int Cs_stride, int Cs_off, const void* ws_packed, void* y, dua_stat_word* stats, void* stream) {
  using namespace dua;
  if (!dua_deconv_k2s2_res_supported(d, Cs) || !lo || !w_packed || !xskip || !ws_packed || !y || !stats) return DUA_ERR_ARG;
  if (Cs_stride % 8 || Cs_off % 8 || Cs_off + Cs > Cs_stride) return DUA_ERR_ARG;
  if ((long)d->D * d->H * d->W * 8 * Cs_stride >= 0x7fffffffL) return DUA_ERR_ARG;
  if (int e = ensure_prepared()) return e;
  DeconvResArgs ra{};
  DeconvArgs& a = ra.base;
  a.x = lo; a.w = w_packed; a.bias = nullptr; a.y = y;
  a.xf = make_xform(nullptr, d->Cin);
  a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W;
  a.Cin = d->Cin; a.Cin_stride = d->Cin_stride; a.Cin_off = d->Cin_off;
  a.Cout = d->Cout; a.Cout_stride = d->Cout_stride; a.Cout_off = d->Cout_off;
  a.nchunks = (d->Cin + 31) / 32;
  a.nct = (d->Cout + dc::BN - 1) / dc::BN;
  a.out_blk = 0;
  ra.xs = xskip; ra.ws = ws_packed; ra.stats = (stat_t*)stats;
  ra.Cs = Cs; ra.Cs_stride = Cs_stride; ra.Cs_off = Cs_off; ra.cout_pad = a.nct * dc::BN;
  const int nsk = (Cs + 31) / 32;
  const int lds = deconv_res_lds(a.nchunks, nsk);
  const long vox = (long)d->D * d->H * d->W;
  dim3 grid((unsigned)((vox + 127) / 128), a.nct, d->N);
  hipStream_t s = (hipStream_t)stream;
  switch (nsk) {
    case 1: hipLaunchKernelGGL(deconv_k2s2_res_kernel<1>, grid, dim3(256), lds, s, ra); break;
    case 2: hipLaunchKernelGGL(deconv_k2s2_res_kernel<2>, grid, dim3(256), lds, s, ra); break;
    case 3: hipLaunchKernelGGL(deconv_k2s2_res_kernel<3>, grid, dim3(256), lds, s, ra); break;
    default: hipLaunchKernelGGL(deconv_k2s2_res_kernel<4>, grid, dim3(256), lds, s, ra); break;
  }
  return (int)hipGetLastError();
}

