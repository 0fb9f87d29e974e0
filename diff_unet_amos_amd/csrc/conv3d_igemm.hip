// 3x3x3 (pad 1, stride 1) 3-D convolution as implicit GEMM on the gfx950 matrix cores.
//
// Replaces every `Conv3d k3 p1` the reference reaches through MONAI's Convolution block
// (models/basic_unet/denoiser.py:56-59, models/basic_unet/pretrained/basic_unet.py:60-63) and
// fuses around it:
//   prologue  : InstanceNorm3d(affine) + LeakyReLU(0.1) (+ timestep-embedding bias) of the
//               PRODUCER layer, applied while the halo tile is staged (denoiser.py:63-67);
//               out-of-volume halo voxels stay literal zeros (padding follows the activation)
//   epilogue  : + bias, per-(n, c) InstanceNorm statistics of THIS layer's output (sum x and sum x^2 from the
//               fp32 accumulators, combined per workgroup, then two fp64 atomics per channel into one
//               of 8 replica rows), coalesced 16-byte stores of the raw output.
//
// GEMM view: M = output voxels, N = Cout, K = 27 taps x Cin.
// Workgroup (256 threads = 4 waves): a 4x8x8 output tile x 64 output channels; wave w owns
// depth slice w (64 voxels) as 2x2 MFMA 32x32 accumulators.  Per Cin chunk of 64 bytes/voxel
// the 6x10x10 halo tile is staged once in LDS; the 9 taps of one kd plane of the packed
// weights follow it slab by slab.  A and B fragments are 16-byte ds_read_b128; the halo row
// stride (656 B) and the 4x8 row->voxel map make every fragment read bank-conflict free.
#include "common.hpp"
#include "../../include/dua_hip.h"
#include "conv3_args.hpp"
#include "stamp.hpp"
#include <algorithm>

namespace dua {

namespace c3 {
constexpr int TD = 4, TH = 8, TW = 8;
constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;
constexpr int KG = 4;                      // k-groups (16 B) per chunk
constexpr int VS = KG * 16;                // 64 B per halo voxel per chunk
constexpr int RS = HW * VS + 16;           // 656: halo row stride, padded (bank-conflict free)
constexpr int PS = HH * RS;                // 6560: halo plane stride
constexpr int HALO_BYTES = HD * PS;        // 39360
constexpr int BN = 64;                     // output channels per workgroup
}  // namespace c3

// launch form of a call (dua_conv3_desc.policy, low byte): 0 = auto (split-K for small layers, 2x8x8 tiles for mid-size ones,
// both in the kd-plane form; the wide-tile form where there are tiles to spare); 2 forces 4x8x8 tiles without split-K, 3 forces
// 2x8x8 (slab form), 6 = the automatic policy with the slab form everywhere, 7 = without the wide-tile form
static inline int conv_variant_of(const dua_conv3_desc* d) { return d->policy & 0xff; }
static inline bool conv_policy_ok(const dua_conv3_desc* d) {
  const int v = d->policy & 0xff;
  return (d->policy & ~(0xff | DUA_POLICY_NO_FINISH)) == 0 && (v == 0 || v == 2 || v == 3 || v == 6 || v == 7 || v == 8 || v == 9);
}

#ifdef DUA_ABLATE
extern int g_wgrad_abl;
#endif

// ------------------------------------------------------------------------------------------------
// v2: same tile and fragment maps, software-pipelined.  Weights arrive as 12 KB (kd,kh) slabs, loaded to
// registers one slab ahead of the MFMAs and written into the other half of a double buffer; the next
// chunk's halo is prefetched into registers during the last slab of the current chunk and written
// (transformed) to LDS between two barriers.  Operand fragments of k-step t+1 are read from LDS while
// the MFMAs of step t issue.  One barrier per slab; nothing waits on a global load that was not issued
// a full MFMA phase earlier.
namespace c3v2 {
using namespace c3;
constexpr int SLAB = 3 * KG * BN * 16;             // 12288
constexpr int LDS_MAIN = HALO_BYTES + 2 * SLAB;    // 63936
}  // namespace c3v2

__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// TDP = tile depth: 4 (4x8x8 voxels, wave = depth slice, two 32-row blocks per wave) or 2 (2x8x8 voxels, wave =
// (depth slice, h half), one block per wave) -- the small tile doubles the workgroup count of the 24^3 layers.
// NKS = k-steps of 16 channels taken per tap and chunk (2 = the whole 32-channel chunk).  NKS < 2 is the SINGLE-CHANNEL
// TAP form for first layers (fp16, one Cin chunk): the packed input holds 16 * NKS ordinary channels followed by ONE
// more real channel (the conditioning image of DiffUNet: torch.cat([image, x]) at models/basic_unet/denoiser.py:298; the
// encoder's only input channel) and zero padding.  Contracting that channel as part of a padded 16-wide k-step spends
// 27 MFMA k-steps on one channel; here its 27 taps are gathered from the halo into a [voxel][32] tile (im2col of one
// channel) and contracted in TWO k-steps against a [32 taps][64 couts] weight block that dua_pack_conv3_weights_tap
// appends to the packed weights: 27 + 2 k-steps per tile instead of 54 for DiffUNet's 17-channel first layer.
// HALF: the last Cin chunk holds at most one k-step (16 channels) of real input (Cin = 48 = 32 + 16, the Swin-UNETR widths):
// its second k-step per tap would multiply zero padding and is skipped.  A separate instantiation, so that the common
// kernel keeps its schedule.
// BIG (layers that cannot put two workgroups on every CU: <= 24^3): ONE workgroup per CU with most of the LDS.  Weights
// arrive as whole kd planes (9 taps, 36 KB) by LDS-DMA into a ring of three -- no staging registers, no store phase, the
// plane after next in flight while this one multiplies -- so a phase is 18 k-steps between two barriers instead of 6, and
// with a single wave per SIMD the fragment reads run three k-steps ahead of the MFMAs (2x8x8 tiles: the two MFMAs of a
// k-step are shorter than one LDS round trip; measured with in-kernel stamps, 1010 cycles per 384-cycle phase).
template <typename T, int TDP = 4, int NKS = 2, bool HALF = false, bool BIG = false>
__global__ __launch_bounds__(256, BIG ? 1 : 2) void conv3d_k3_v2_kernel(Conv3Args a) {
  using namespace c3v2;
  constexpr int TD = TDP, HD = TDP + 2, MB = TDP == 2 ? 1 : 2;
  constexpr int NW = 4, NTHR = 64 * NW;
  constexpr int BSLAB = 3 * SLAB;                                            // one kd plane of the packed weights
  constexpr int HALO_BYTES = HD * PS, LDS_MAIN = HALO_BYTES + (BIG ? 3 * BSLAB : 2 * SLAB);
  static_assert(!BIG || (NKS == 2 && !HALF), "the kd-plane form is the plain 32-channel-chunk kernel");
  constexpr int NITEMS = HD * HH * HW * KG, NIT = (NITEMS + NTHR - 1) / NTHR;
  constexpr int NPIECE = SLAB / 16, NSL = (NPIECE + NTHR - 1) / NTHR;      // 16-byte pieces of a weight slab per thread
  using Frag = typename Elem<T>::Frag;
  constexpr int EPG = Elem<T>::EPG;
  constexpr int CK = KG * EPG;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* wbuf = smem + HALO_BYTES;
  float* xsc = (float*)(smem + LDS_MAIN);
  float* xsh = xsc + a.nchunks * CK;
  float* xad = xsh + a.nchunks * CK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int dwave = TDP == 2 ? wave >> 1 : wave;          // depth slice of this wave
  const int hbase = TDP == 2 ? (wave & 1) * 4 : 0;        // first h row of this wave's block(s)
  const int tile = xcd_remap(blockIdx.x, a.ntiles);
  const int ct = blockIdx.y, n = blockIdx.z % a.N;
  const int tw_ = tile % a.tiles_w, th_ = (tile / a.tiles_w) % a.tiles_h, td_ = tile / (a.tiles_w * a.tiles_h);
  const int d0 = td_ * TD, h0 = th_ * TH, w0 = tw_ * TW;
  const bool fused = a.xf.stats != nullptr;

  const int kg_t = tid & (KG - 1);
  long goff[NIT]; int loff[NIT];
  const T* xin = (const T*)a.x + (long)n * a.D * a.H * a.W * a.Cin_stride + a.Cin_off;
#pragma unroll
  for (int j = 0; j < NIT; ++j) {
    int it = tid + NTHR * j;
    int hv = it >> 2;
    int hd = hv / (HH * HW), rem = hv - hd * (HH * HW), hy = rem / HW, hx = rem - hy * HW;
    int gd = d0 + hd - 1, gh = h0 + hy - 1, gw = w0 + hx - 1;
    bool ok = it < NITEMS && gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
    goff[j] = ok ? (((long)gd * a.H + gh) * a.W + gw) * a.Cin_stride + kg_t * EPG : -1;
#ifndef DUA_HALO_BRANCHFREE
    loff[j] = it < NITEMS ? hd * PS + hy * RS + hx * VS + kg_t * 16 : -1;
#else
    loff[j] = it < NITEMS ? hd * PS + hy * RS + hx * VS + kg_t * 16 : HW * VS;     // beyond the halo: the pad bytes of row 0
#endif
  }
  const char* wsrc = (const char*)a.w + (long)ct * a.nchunks * 9 * SLAB;
  // Weight slabs go global -> registers (issued one slab ahead, 3 x 16 B per thread) -> LDS.  LDS-DMA
  // (global_load_lds) would save the registers, but with one in flight hipcc (ROCm 7.2) turns every
  // counted lgkmcnt wait of the fragment pipeline below into lgkmcnt(0).
  // Two register sets: slab g is loaded during phase g-2 and written to LDS at the end of phase g-1, so
  // an L2 round trip has two MFMA phases to complete.
  f32x4 wreg[3][NSL];   // set = (slab index within chunk) % 3, compile-time (9 slabs per chunk keeps the cycle)
  auto load_slab = [&](int g, int set) {       // g = global slab index (chunk * 9 + kd * 3 + kh)
    const char* src = wsrc + (long)g * SLAB + tid * 16;
#pragma unroll
    for (int j = 0; j < NSL; ++j)
      if (NPIECE % NTHR == 0 || tid + NTHR * j < NPIECE) wreg[set][j] = *(const f32x4*)(src + j * NTHR * 16);
  };
  auto store_slab = [&](int g, int set) {
    char* dst = wbuf + (g & 1) * SLAB + tid * 16;
#pragma unroll
    for (int j = 0; j < NSL; ++j)
      if (NPIECE % NTHR == 0 || tid + NTHR * j < NPIECE) *(f32x4*)(dst + j * NTHR * 16) = wreg[set][j];
  };
  Frag hv_[NIT];
#ifndef DUA_HALO_BRANCHFREE     // default; -DDUA_HALO_BRANCHFREE selects the select-based form below (hipcc 7.2 fails to compile it with -fPIC)
  auto load_halo = [&](int ch) {
    const bool cok = ch * CK + kg_t * EPG < a.Cin;
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      if (goff[j] >= 0 && cok) hv_[j] = *(const Frag*)(xin + goff[j] + ch * CK);
      else
#pragma unroll
        for (int e = 0; e < EPG; ++e) hv_[j][e] = (T)0.f;
    }
  };
  auto store_halo = [&](int ch) {
    const int c0 = ch * CK + kg_t * EPG;
    if (fused && c0 < a.Cin) {
      float sc[EPG], sh[EPG], ad[EPG], sn[EPG];
#pragma unroll
      for (int e = 0; e < EPG; ++e) { sc[e] = xsc[c0 + e]; sh[e] = xsh[c0 + e]; ad[e] = xad[c0 + e]; }
      xform_prep<T>(sc, sh, ad, sn, a.xf.slope);
#pragma unroll
      for (int j = 0; j < NIT; ++j)
        if (goff[j] >= 0) hv_[j] = xform_frag<T>(hv_[j], sc, sh, ad, sn, a.xf.slope);
    }
#pragma unroll
    for (int j = 0; j < NIT; ++j)
      if (loff[j] >= 0) *(Frag*)(halo + loff[j]) = hv_[j];
  };
#else
  // Branch-free staging: every item loads (out-of-volume / padding-channel items from a clamped, always valid address), is
  // transformed, and is zeroed by a select afterwards -- under a lane-dependent branch hipcc gives every load a basic
  // block of its own (the loads then issue one by one); items beyond the halo store into the pad bytes of halo row 0.
  auto load_halo = [&](int ch) {
    const bool cok = ch * CK + kg_t * EPG < a.Cin;
#pragma unroll
    for (int j = 0; j < NIT; ++j) hv_[j] = *(const Frag*)(xin + (goff[j] >= 0 && cok ? goff[j] + ch * CK : 0));
  };
  auto store_halo = [&](int ch) {
    const int c0 = ch * CK + kg_t * EPG;
    const bool cok = c0 < a.Cin;
    float sc[EPG], sh[EPG], ad[EPG], sn[EPG];
#pragma unroll
    for (int e = 0; e < EPG; ++e) { sc[e] = 1.f; sh[e] = 0.f; ad[e] = 0.f; }
    if (fused && cok) {
#pragma unroll
      for (int e = 0; e < EPG; ++e) { sc[e] = xsc[c0 + e]; sh[e] = xsh[c0 + e]; ad[e] = xad[c0 + e]; }
    }
    xform_prep<T>(sc, sh, ad, sn, a.xf.slope);
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      Frag v = hv_[j];
      if (fused) v = xform_frag<T>(v, sc, sh, ad, sn, a.xf.slope);
      const bool ok = goff[j] >= 0 && cok;
#pragma unroll
      for (int e = 0; e < EPG; ++e) v[e] = ok ? v[e] : (T)0.f;
      if (NITEMS % NTHR == 0 || tid + NTHR * j < NITEMS) *(Frag*)(halo + loff[j]) = v;
    }
  };
#endif

  // BIG: one kd plane (36 pieces of 1 KB: 9 per wave) straight into ring slot `slot`.  Inline assembly (M0 saved and restored
  // inside the statement): hipcc neither counts these transfers nor waits for them; the loop below does (counted vmcnt,
  // then the barrier, then the reads).
  auto dma_unit = [&](int u, int slot) {
    const char* src = wsrc + (long)u * BSLAB + (wave * 9) * 1024 + lane * 16;
    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(wbuf + slot * BSLAB + (wave * 9) * 1024);
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src + j * 1024), "s"(__builtin_amdgcn_readfirstlane(dst + j * 1024)) : "memory");
    }
  };

  // accumulators start at the bias of the lane's output channel (split-K adds it in the finish kernel instead)
  f32x16 acc[MB][2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int bc = ct * BN + q * 32 + r;               // bias needs Cout entries only (padding channels start at zero)
    const float b0 = (a.part || bc >= a.Cout) ? 0.f : a.bias[bc];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][q][i] = b0;
  }
  // ---- work range: units u = chunk * 3 + kd, three (kd, kh) slabs each ----
  const int ks_id = blockIdx.z / a.N;
  const int u0 = ks_id * a.units_per_split;
  const int u1 = NKS == 0 ? u0 : min(a.nchunks * 3, u0 + a.units_per_split);   // NKS 0: the tap channel is the whole input
  const int g0 = u0 * 3, g1 = u1 * 3;

  DUA_STAMP_AT(0, true);
  DUA_STAMP_AT(2, false);
  // ---- prologue ----
  if constexpr (BIG) {
    load_halo(u0 / 3);
    dma_unit(u0, 0);
    if (u0 + 1 < u1) dma_unit(u0 + 1, 1);
    DUA_STAMP_AT(59, false);
    if (fused) {
      xform_preamble(a.xf, n, min(a.Cin, ((u1 + 2) / 3) * CK), xsc, xsh, xad, (u0 / 3) * CK);
      __syncthreads();
    }
    DUA_STAMP_AT(60, false);
    DUA_STAMP_AT(61, false);
    store_halo(u0 / 3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // both planes have landed (this wave's pieces; the barrier: everyone's)
    __syncthreads();
  } else {
  load_slab(g0, 0);
  load_slab(g0 + 1, 1);
  load_slab(g0 + 2, 2);
  load_halo(u0 / 3);
  DUA_STAMP_AT(59, false);
  if (fused) {     // only the Cin chunks this workgroup walks (all of them unless split-K)
    xform_preamble(a.xf, n, min(a.Cin, ((u1 + 2) / 3) * CK), xsc, xsh, xad, (u0 / 3) * CK);
    __syncthreads();
  }
  DUA_STAMP_AT(60, false);
  store_slab(g0, 0);
  DUA_STAMP_AT(61, false);
  store_halo(u0 / 3);
  if constexpr (NKS < 2) {       // tap block of this cout tile: [4 k-groups of 8 taps][64 couts][16 B], behind the slabs
    const char* wt = (const char*)a.w + (long)gridDim.y * a.nchunks * 9 * SLAB + (long)ct * 4096;
    *(f32x4*)(smem + LDS_MAIN + tid * 16) = *(const f32x4*)(wt + tid * 16);
  }
  __syncthreads();
  }

  DUA_STAMP_AT(3, false);
  const int a_base = dwave * PS + (hbase + (r >> 3)) * RS + (r & 7) * VS + hh * 16;
  const int b_base = (hh * BN + r) * 16;
  if constexpr (BIG) {
    constexpr int NTB = 9 * NKS, PF = MB == 1 ? 3 : 1;     // k-steps per plane; fragment prefetch distance
    for (int u = u0; u < u1; ++u) {
      const int kd = u % 3, slot = (u - u0) % 3;
      const bool next_chunk = kd == 2 && u + 1 < u1;
      if (next_chunk) load_halo(u / 3 + 1);                 // older than the transfers below: the counted wait covers it
      const bool ahead = u + 2 < u1;
      if (ahead) dma_unit(u + 2, (u - u0 + 2) % 3);         // that slot was last read before the previous barrier
      const char* ap = halo + a_base + kd * PS;
      const char* wb = wbuf + slot * BSLAB + b_base;
      Frag fa0[PF + 1], fa1[PF + 1], fb0[PF + 1], fb1[PF + 1];
      auto ldb = [&](int t, int b) {
        const int tap = t / NKS, ks = t % NKS, kh = tap / 3, kw = tap % 3;
        fa0[b] = *(const Frag*)(ap + kh * RS + kw * VS + ks * 32);
        fb0[b] = *(const Frag*)(wb + (tap * KG + 2 * ks) * BN * 16);
        fb1[b] = *(const Frag*)(wb + (tap * KG + 2 * ks) * BN * 16 + 32 * 16);
        if (MB == 2) fa1[b] = *(const Frag*)(ap + 4 * RS + kh * RS + kw * VS + ks * 32);
      };
#pragma unroll
      for (int t = 0; t < PF; ++t) ldb(t, t);
#pragma unroll
      for (int t = 0; t < NTB; ++t) {
        if (t + PF < NTB) ldb(t + PF, (t + PF) % (PF + 1));
        __builtin_amdgcn_sched_barrier(0);
        mma32(acc[0][0], fa0[t % (PF + 1)], fb0[t % (PF + 1)]);
        mma32(acc[0][1], fa0[t % (PF + 1)], fb1[t % (PF + 1)]);
        if constexpr (MB == 2) {
          mma32(acc[1][0], fa1[t % (PF + 1)], fb0[t % (PF + 1)]);
          mma32(acc[1][1], fa1[t % (PF + 1)], fb1[t % (PF + 1)]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // plane u + 1 (requested a whole phase ago) must have landed before anyone reads it; plane u + 2 stays in flight
      if (ahead) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (u - u0 < 55) DUA_STAMP_AT(4 + (u - u0), false);
      if (next_chunk) {
        store_halo(u / 3 + 1);
        __syncthreads();
      }
    }
  } else
  for (int u = u0; u < u1; ++u) {
    const int kd = u % 3;
    const bool next_chunk = kd == 2 && u + 1 < u1;     // the unit after this one starts a new Cin chunk
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int g = u * 3 + kh;
      // slab g+1 (requested two phases ago) goes into the other buffer first, so that the end of the phase is
      // nothing but the barrier; its register set is then free for slab g+3
      if (g + 1 < g1) store_slab(g + 1, (kh + 1) % 3);
      if (g + 3 < g1) load_slab(g + 3, kh);                       // (kh + 3) % 3 == kh
      if (kh == 0 && next_chunk) load_halo(u / 3 + 1);
      const char* ap = halo + a_base + kd * PS + kh * RS;
      const char* wb = wbuf + (g & 1) * SLAB + b_base;
      // 3 * NKS k-steps (kw x ks); fragments of step t+1 are in flight while the MFMAs of step t issue
      if constexpr (NKS > 0) {
        constexpr int NT = 3 * NKS;
        Frag fa0[2], fa1[2], fb0[2], fb1[2];
        auto ld = [&](int t, int b) {
          const int kw = t / NKS, ks = t % NKS;
          fa0[b] = *(const Frag*)(ap + kw * VS + ks * 32);
          fb0[b] = *(const Frag*)(wb + (kw * KG + 2 * ks) * BN * 16);
          fb1[b] = *(const Frag*)(wb + (kw * KG + 2 * ks) * BN * 16 + 32 * 16);
          if (MB == 2) fa1[b] = *(const Frag*)(ap + 4 * RS + kw * VS + ks * 32);
        };
        const bool short_chunk = HALF && u / 3 == a.nchunks - 1;     // uniform: skip the ks = 1 steps of this unit
        ld(0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (t + 1 < NT && !(HALF && short_chunk && ((t + 1) % NKS) == 1)) ld(t + 1, (t + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);       // keep the prefetch ahead of the MFMAs (hipcc sinks it otherwise)
          if (!(HALF && short_chunk && (t % NKS) == 1)) {
            mma32(acc[0][0], fa0[t & 1], fb0[t & 1]);
            mma32(acc[0][1], fa0[t & 1], fb1[t & 1]);
            if constexpr (MB == 2) {
              mma32(acc[1][0], fa1[t & 1], fb0[t & 1]);
              mma32(acc[1][1], fa1[t & 1], fb1[t & 1]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();   // next slab visible and everyone is done with this one
      if (g - g0 < 55) DUA_STAMP_AT(4 + (g - g0), false);
    }
    if (next_chunk) {
      store_halo(u / 3 + 1);
      __syncthreads();
    }
  }
  if constexpr (NKS < 2) {
    // ---- the tap channel: im2col of ONE input channel into the (now free) weight buffers, then two k-steps ----
    // thread tid <-> output voxel (d = tid >> 6, h = (tid >> 3) & 7, w = tid & 7) = MFMA row (tid & 31) of block
    // (tid >> 5) & 1 of wave tid >> 6: row = [27 taps | 5 zeros] fp16, 64 B + 16 B pad (80-byte stride: the 16-lane
    // groups of ds_read_b128 then touch 16 different 16-byte slots).
    constexpr int IS = 80;
    static_assert(sizeof(T) == 2 && TDP == 4, "the tap form is fp16, 4x8x8 tiles");
    const char* hp = halo + (tid >> 6) * PS + ((tid >> 3) & 7) * RS + (tid & 7) * VS + a.tap_ch * 2;
    f16 tv[32];
#pragma unroll
    for (int t = 0; t < 27; ++t) tv[t] = *(const f16*)(hp + (t / 9) * PS + ((t / 3) % 3) * RS + (t % 3) * VS);
#pragma unroll
    for (int t = 27; t < 32; ++t) tv[t] = (f16)0.f;
    char* im = wbuf + tid * IS;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = tv[g * 8 + e];
      *(f16x8*)(im + g * 16) = o;
    }
    __syncthreads();
    const char* wt = smem + LDS_MAIN;
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
      const f16x8 b0 = *(const f16x8*)(wt + ((2 * sidx + hh) * BN + r) * 16);
      const f16x8 b1 = *(const f16x8*)(wt + ((2 * sidx + hh) * BN + 32 + r) * 16);
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const f16x8 av = *(const f16x8*)(wbuf + (wave * 64 + m * 32 + r) * IS + (2 * sidx + hh) * 16);
        mma32(acc[m][0], av, b0);
        mma32(acc[m][1], av, b1);
      }
    }
    __syncthreads();     // the epilogue's staging tile reuses the halo; nobody may still be gathering from it
  }
  DUA_STAMP_AT(62, false);
  if (a.part != nullptr) {
    // ---- split-K: this workgroup's fp32 partial tile goes to part[ks][n][voxel][cout_pad] ----
    constexpr int OSF = 32 * 4 + 16;
    char* otf = smem + wave * (32 * MB) * OSF;
    const int gdz = d0 + dwave;
    float* pout = a.part + ((long)(ks_id * a.N + n) * a.D * a.H * a.W) * a.cout_pad + ct * BN;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) *(float*)(otf + (m * 32 + acc_row(i, hh)) * OSF + r * 4) = acc[m][q][i];
      __syncthreads();
      if (gdz < a.D) {
#pragma unroll
        for (int it = 0; it < 4 * MB; ++it) {
          const int v = it * 8 + (lane >> 3), cg = lane & 7;
          const int gh = h0 + hbase + (v >> 3), gw = w0 + (v & 7);
          if (gh < a.H && gw < a.W)
            *(f32x4*)(pout + (((long)gdz * a.H + gh) * a.W + gw) * a.cout_pad + q * 32 + cg * 4) =
                *(const f32x4*)(otf + v * OSF + cg * 16);
        }
      }
      if (q == 0) __syncthreads();
    }
    DUA_STAMP_AT(63, false);
    DUA_STAMP_AT(1, true);
    return;
  }
  // ---- epilogue, one 32-channel half at a time (fits fp32 too) ----
  // Statistics are taken from the fp32 accumulators (sum x, sum x^2 per lane over its 32 voxels, then fp64);
  // tiles that lie fully inside the volume -- all of them at 96/48/24^3 -- skip the per-voxel masks.
  constexpr int OS = 32 * (int)sizeof(T) + 16;
  char* ot = smem + wave * (32 * MB) * OS;
  float* ex = (float*)(smem + NW * (32 * MB) * OS);   // [NW waves][64 couts][2]
  const int gd = d0 + dwave;
  const bool dok = gd < a.D;
  const bool full = d0 + TD <= a.D && h0 + TH <= a.H && w0 + TW <= a.W;
  T* yout = (T*)a.y + (long)n * a.D * a.H * a.W * a.Cout_stride + a.Cout_off + ct * BN;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int co = q * 32 + r;
    // fp32 (parity) instantiations keep the lane's partial sums in double: a channel whose mean is ~100 standard deviations
    // loses three digits of its variance to fp32 partial sums of x^2 (sum x^2 / n - mean^2), and InstanceNorm's backward
    // amplifies that further (seen as 15 % errors on single channels of coarse-level weight gradients against an fp64 oracle
    // where torch's CPU fp32 path shows 2e-6); each wave then adds its own contribution to the statistics words.
    constexpr bool F32 = sizeof(T) == 4;
    using part_t = typename std::conditional<F32, double, float>::type;
    part_t s = 0, ss = 0;
    if (full) {
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float v = acc[m][q][i];
          s += v;
          if constexpr (F32) ss += (double)v * (double)v; else ss = fmaf(v, v, ss);
          *(T*)(ot + (m * 32 + acc_row(i, hh)) * OS + r * (int)sizeof(T)) = (T)v;
        }
    } else {
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int hl = hbase + 4 * m + (i >> 2), wl = (i & 3) + 4 * hh;
          const bool ok = dok && (h0 + hl < a.H) && (w0 + wl < a.W);
          const float v = ok ? acc[m][q][i] : 0.f;
          s += v;
          if constexpr (F32) ss += (double)v * (double)v; else ss = fmaf(v, v, ss);
          *(T*)(ot + (m * 32 + acc_row(i, hh)) * OS + r * (int)sizeof(T)) = (T)v;
        }
    }
    s += __shfl_xor(s, 32);
    ss += __shfl_xor(ss, 32);
    if constexpr (F32) {
      if (hh == 0 && ct * BN + co < a.Cout)
        stats_add(a.stats, n, a.cout_pad, blockIdx.x & (STAT_REPLICAS - 1), ct * BN + co, (double)s, (double)ss);
    } else {
      if (hh == 0) { ex[(wave * BN + co) * 2] = (float)s; ex[(wave * BN + co) * 2 + 1] = (float)ss; }
    }
    __syncthreads();
    if (dok) {
      constexpr int GPV = 32 / EPG;            // 16-byte groups per voxel in this half
      constexpr int VPI = 64 / GPV;
#pragma unroll
      for (int it = 0; it < (32 * MB) / VPI; ++it) {
        const int v = it * VPI + lane / GPV, cg = lane % GPV;
        const int gh = h0 + hbase + (v >> 3), gw = w0 + (v & 7);
        if ((full || (gh < a.H && gw < a.W)) && ct * BN + q * 32 + cg * EPG < a.Cout)
          *(Frag*)(yout + (((long)gd * a.H + gh) * a.W + gw) * a.Cout_stride + q * 32 + cg * EPG) =
              *(const Frag*)(ot + v * OS + cg * 16);
      }
    }
    if (q == 0) __syncthreads();               // staging tile is reused by the second half
  }
  if (wave == 0 && sizeof(T) == 2) {
    double S = 0, Q = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { S += (double)ex[(w * BN + lane) * 2]; Q += (double)ex[(w * BN + lane) * 2 + 1]; }
    if (ct * BN + lane < a.Cout) stats_add(a.stats, n, a.cout_pad, blockIdx.x & (STAT_REPLICAS - 1), ct * BN + lane, S, Q);
  }
  DUA_STAMP_AT(63, false);
  DUA_STAMP_AT(1, true);
}

// ------------------------------------------------------------------------------------------------
// First layer of the denoiser (16 noisy-label channels + the conditioning image -> 64, models/basic_unet/denoiser.py:298
// torch.cat([image, x]) into conv_0): K = 27 taps x 16 channels + 27 image taps is so short that the slab pipeline above
// spends its time in barriers (nine per tile for 116 MFMAs per wave).  Here ALL 27 taps of the 16 ordinary channels stay
// resident in LDS (54 KB, loaded once by a PERSISTENT workgroup that walks tiles), the image-tap weight block lives in
// registers, and a tile is: halo (prefetched into registers under the previous tile's MFMAs) -> LDS, one barrier, 116
// MFMAs per wave with no barrier between them, epilogue.  Two workgroups per CU (79.5 KB of LDS each) overlap one's
// epilogue with the other's MFMAs.
//
// Halo image: 32 B per voxel (the 16 ordinary channels; the image channel goes to an fp16 array of its own), 12 voxel slots
// per row (10 used), and the two 16-byte halves of a voxel swapped on odd halo rows: with that the four 16-lane groups of
// every ds_read_b128 fragment read cover all 64 banks once (rows r>>3 = 0..3 of a 32-voxel block land on 16-byte slots
// {0,2,4,6}+8k / {1,3,5,7}+8k).
namespace c3f {
using namespace c3;
constexpr int VSF = 32, RSF = 12 * VSF, PSF = HH * RSF;      // 384-byte rows, 3840-byte planes
constexpr int HALO_F = HD * PSF;                              // 23040
constexpr int WRES = 27 * 2 * BN * 16;                        // 55296: [tap][k-half][64 couts][16 B]
constexpr int IMG_F = 1216;                                   // 600 fp16 of the image halo (+ pad)
constexpr int LDS_F = WRES + HALO_F + IMG_F;                  // 79552: two workgroups per CU
constexpr int NV = HD * HH * HW;                              // 600 halo voxels
}  // namespace c3f

__global__ __launch_bounds__(256, 2) void conv3d_k3_first_kernel(Conv3Args a, int items) {
  using namespace c3f;
  using T = f16;
  constexpr int NTHR = 256, NW = 4, MB = 2;
  constexpr int NIT = (NV * 2 + NTHR - 1) / NTHR;             // 16-byte halo pieces per thread (5)
  constexpr int NIM = (NV + NTHR - 1) / NTHR;                 // image halo values per thread (3)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wres = smem;
  char* halo = smem + WRES;
  char* img = smem + WRES + HALO_F;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int ct = blockIdx.y;

  DUA_STAMP_AT(0, true);
  DUA_STAMP_AT(2, false);
  // ---- once per workgroup: resident weights (k-groups 0 and 1 of every tap of the packed slabs) by LDS-DMA, 54 pieces of
  // 1 KB = [64 couts][16 B], all in flight together (inline assembly as in the kd-plane form above: the explicit
  // s_waitcnt before the first barrier covers them), and the image-tap block ----
  {
    const char* wsrc = (const char*)a.w + (long)ct * a.nchunks * 9 * c3v2::SLAB + lane * 16;
    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)wres;
#pragma unroll
    for (int j = 0; j < 14; ++j) {
      const int s2 = wave + NW * j;                           // piece = (tap, k-half)
      if (s2 < 54) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(wsrc + ((s2 >> 1) * KG + (s2 & 1)) * 1024), "s"(__builtin_amdgcn_readfirstlane(dst + s2 * 1024)) : "memory");
      }
    }
  }
  f16x8 tb[2][2];                                             // image-tap weights: [k-step][cout half], this lane's B fragments
  {
    const char* wt = (const char*)a.w + (long)gridDim.y * a.nchunks * 9 * c3v2::SLAB + (long)ct * 4096;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int q = 0; q < 2; ++q) tb[s][q] = *(const f16x8*)(wt + ((2 * s + hh) * BN + q * 32 + r) * 16);
  }

  // ---- per-thread halo items (tile independent): piece it = (voxel, half).  Branch-free: every thread loads every item
  // (out-of-volume ones from a clamped address, zeroed by a select before the LDS store); threads without an item store
  // into the unused voxel slots 10 / 11 of halo row 0 and the pad of the image array. ----
  int hrel[NIT], loff[NIT];                                   // packed (hd, hy, hx) / LDS offset
#pragma unroll
  for (int j = 0; j < NIT; ++j) {
    const int it = tid + NTHR * j, hv = it >> 1, p = it & 1;
    const int hd = hv / (HH * HW), rem = hv - hd * (HH * HW), hy = rem / HW, hx = rem - hy * HW;
    hrel[j] = it < NV * 2 ? hd | (hy << 8) | (hx << 16) : 0x7fffffff;        // no item: out of every volume
    loff[j] = it < NV * 2 ? hd * PSF + hy * RSF + hx * VSF + ((p ^ (hy & 1)) << 4) : HW * VSF + (p << 4);
  }
  int irel[NIM], ioff[NIM];
#pragma unroll
  for (int j = 0; j < NIM; ++j) {
    const int hv = tid + NTHR * j;
    const int hd = hv / (HH * HW), rem = hv - hd * (HH * HW), hy = rem / HW, hx = rem - hy * HW;
    irel[j] = hv < NV ? (hd | (hy << 8) | (hx << 16)) : 0x7fffffff;
    ioff[j] = (hv < NV ? hv : NV + (tid & 7)) * 2;
  }
  const int p_t = tid & 1;

  f16x8 hreg[NIT];
  f16 ireg[NIM];
  unsigned okmask = 0;                                        // bit j: piece j is inside the volume; bit 8 + j: image value j
  auto tile_of = [&](int item, int& n, int& d0, int& h0, int& w0) {
    n = item / a.ntiles;
    const int tile = xcd_remap(item - n * a.ntiles, a.ntiles);
    const int tw_ = tile % a.tiles_w, th_ = (tile / a.tiles_w) % a.tiles_h, td_ = tile / (a.tiles_w * a.tiles_h);
    d0 = td_ * TD; h0 = th_ * TH; w0 = tw_ * TW;
  };
  auto load_halo = [&](int item) {
    int n, d0, h0, w0;
    tile_of(item, n, d0, h0, w0);
    const T* xin = (const T*)a.x + (long)n * a.D * a.H * a.W * a.Cin_stride + a.Cin_off;
    okmask = 0;
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int gd = d0 + (hrel[j] & 255) - 1, gh = h0 + ((hrel[j] >> 8) & 255) - 1, gw = w0 + (hrel[j] >> 16) - 1;
      const bool ok = (unsigned)gd < (unsigned)a.D && (unsigned)gh < (unsigned)a.H && (unsigned)gw < (unsigned)a.W;
      const long off = ok ? (((long)gd * a.H + gh) * a.W + gw) * a.Cin_stride + p_t * 8 : 0;
      hreg[j] = *(const f16x8*)(xin + off);
      okmask |= ok ? 1u << j : 0u;
    }
#pragma unroll
    for (int j = 0; j < NIM; ++j) {
      const int gd = d0 + (irel[j] & 255) - 1, gh = h0 + ((irel[j] >> 8) & 255) - 1, gw = w0 + (irel[j] >> 16) - 1;
      const bool ok = (unsigned)gd < (unsigned)a.D && (unsigned)gh < (unsigned)a.H && (unsigned)gw < (unsigned)a.W;
      const long off = ok ? (((long)gd * a.H + gh) * a.W + gw) * a.Cin_stride + a.tap_ch : 0;
      ireg[j] = xin[off];
      okmask |= ok ? 256u << j : 0u;
    }
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const f32x4 raw = __builtin_bit_cast(f32x4, hreg[j]);
      const bool ok = (okmask >> j) & 1;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = ok ? raw[e] : 0.f;
      *(f32x4*)(halo + loff[j]) = v;
    }
#pragma unroll
    for (int j = 0; j < NIM; ++j) *(T*)(img + ioff[j]) = ((okmask >> (8 + j)) & 1) ? ireg[j] : (T)0.f;
  };

  // fragment addresses: wave = depth slice, lane row r = (h = r >> 3, w = r & 7) of block m (h + 4 m), k-half hh
  const int sw = (r >> 3) & 1;
  const int a_base = wave * PSF + (r >> 3) * RSF + (r & 7) * VSF;
  const int a_even = a_base + ((hh ^ sw) << 4), a_odd = a_base + ((hh ^ sw ^ 1) << 4);     // halo row parity of h + kh
  const int b_base = (hh * BN + r) * 16;
  const int i_base = (wave * (HH * HW) + (r >> 3) * HW + (r & 7)) * 2;

  float bias_q[2];                                            // read once: a load inside the tile loop would have to wait for
#pragma unroll                                                // the halo prefetch issued before it (vmcnt counts in order)
  for (int q = 0; q < 2; ++q) {
    const int bc = ct * BN + q * 32 + r;
    bias_q[q] = bc >= a.Cout ? 0.f : a.bias[bc];
  }
  int item = blockIdx.x;
  if (item < items) load_halo(item);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the weight pieces (not counted by the compiler) and the halo
  store_halo();
  __syncthreads();
  DUA_STAMP_AT(3, false);
  int slot = 4;                                               // stamp build: 4 slots per tile (MFMAs | barrier | epilogue | next halo)
  for (; item < items; item += gridDim.x) {
    int n, d0, h0, w0;
    tile_of(item, n, d0, h0, w0);
    const bool more = item + (int)gridDim.x < items;
    if (more) load_halo(item + gridDim.x);                   // lands under the MFMAs below
    f32x16 acc[MB][2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][q][i] = bias_q[q];
    // image taps of this lane's two voxel rows: k = 32 taps (27 real; the weights of the rest are zero, any finite value serves)
    f16x8 ta[MB][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int t0 = s * 16 + e, t1 = t0 + 8;
        const int o0 = ((t0 / 9) * (HH * HW) + ((t0 / 3) % 3) * HW + t0 % 3) * 2;
        const int o1 = t1 < 27 ? ((t1 / 9) * (HH * HW) + ((t1 / 3) % 3) * HW + t1 % 3) * 2 : 0;
        const int o = i_base + (hh ? o1 : o0);
#pragma unroll
        for (int m = 0; m < MB; ++m) ta[m][s][e] = *(const f16*)(img + o + m * 4 * HW * 2);
      }
    {
      constexpr int NT = 27, PF = 2;
      f16x8 fa0[PF + 1], fa1[PF + 1], fb0[PF + 1], fb1[PF + 1];
      auto ld = [&](int t, int b) {
        const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
        const char* ap = halo + ((kh & 1) ? a_odd : a_even) + kd * PSF + kh * RSF + kw * VSF;
        fa0[b] = *(const f16x8*)ap;
        fa1[b] = *(const f16x8*)(ap + 4 * RSF);
        fb0[b] = *(const f16x8*)(wres + b_base + t * 2 * BN * 16);
        fb1[b] = *(const f16x8*)(wres + b_base + t * 2 * BN * 16 + 32 * 16);
      };
#pragma unroll
      for (int t = 0; t < PF; ++t) ld(t, t);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t + PF < NT) ld(t + PF, (t + PF) % (PF + 1));
        __builtin_amdgcn_sched_barrier(0);
        mma32(acc[0][0], fa0[t % (PF + 1)], fb0[t % (PF + 1)]);
        mma32(acc[0][1], fa0[t % (PF + 1)], fb1[t % (PF + 1)]);
        mma32(acc[1][0], fa1[t % (PF + 1)], fb0[t % (PF + 1)]);
        mma32(acc[1][1], fa1[t % (PF + 1)], fb1[t % (PF + 1)]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        mma32(acc[m][0], ta[m][s], tb[s][0]);
        mma32(acc[m][1], ta[m][s], tb[s][1]);
      }
    if (slot < 56) DUA_STAMP_AT(slot, false);
    __syncthreads();                                          // everyone is done with the halo: the epilogue stages through it
    if (slot < 56) DUA_STAMP_AT(slot + 1, false);

    // ---- epilogue: statistics from the fp32 accumulators; each wave stages one 32-voxel block at a time in rows of its own
    // (32 x 128 B: whole 128-byte voxel lines leave in one store instruction, eight voxels each) -- no workgroup barrier
    // inside, ds operations of one wave execute in order ----
    char* ot = halo + wave * 4096;
    float* ex = (float*)(smem + LDS_F);                       // [NW waves][64 couts][2], beyond the image halo
    const int gd = d0 + wave;
    const bool dok = gd < a.D;
    const bool full = d0 + TD <= a.D && h0 + TH <= a.H && w0 + TW <= a.W;
    const long nvox = (long)a.D * a.H * a.W;
    T* yout = (T*)a.y + (long)n * nvox * a.Cout_stride;
    float s[2] = {0.f, 0.f}, ss[2] = {0.f, 0.f};
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      if (full) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float v = acc[m][q][i];
            s[q] += v;
            ss[q] = fmaf(v, v, ss[q]);
            *(T*)(ot + acc_row(i, hh) * 128 + (q * 32 + r) * 2) = (T)v;
          }
      } else {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int hl = 4 * m + (i >> 2), wl = (i & 3) + 4 * hh;
            const bool ok = dok && (h0 + hl < a.H) && (w0 + wl < a.W);
            const float v = ok ? acc[m][q][i] : 0.f;
            s[q] += v;
            ss[q] = fmaf(v, v, ss[q]);
            *(T*)(ot + acc_row(i, hh) * 128 + (q * 32 + r) * 2) = (T)v;
          }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (dok) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int v = it * 8 + (lane >> 3), cg = lane & 7;
          const int gh = h0 + 4 * m + (v >> 3), gw = w0 + (v & 7);
          if ((full || (gh < a.H && gw < a.W)) && ct * BN + cg * 8 < a.Cout)
            *(f16x8*)(yout + chan_off(a.out_blk, ((long)gd * a.H + gh) * a.W + gw, a.Cout_off + ct * BN + cg * 8, a.Cout_stride, nvox)) =
                *(const f16x8*)(ot + v * 128 + cg * 16);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      s[q] += __shfl_xor(s[q], 32);
      ss[q] += __shfl_xor(ss[q], 32);
      if (hh == 0) { ex[(wave * BN + q * 32 + r) * 2] = s[q]; ex[(wave * BN + q * 32 + r) * 2 + 1] = ss[q]; }
    }
    if (slot < 56) DUA_STAMP_AT(slot + 2, false);
    __syncthreads();                                          // every wave is done with its staging rows; ex is complete
    if (wave == 0) {
      double S = 0, Q = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) { S += (double)ex[(w * BN + lane) * 2]; Q += (double)ex[(w * BN + lane) * 2 + 1]; }
      if (ct * BN + lane < a.Cout) stats_add(a.stats, n, a.cout_pad, item & (STAT_REPLICAS - 1), ct * BN + lane, S, Q);
    }
    if (more) {
      store_halo();
      __syncthreads();
    }
    if (slot < 56) DUA_STAMP_AT(slot + 3, false);
    slot += 4;
  }
  DUA_STAMP_AT(62, false);
  DUA_STAMP_AT(63, false);
  DUA_STAMP_AT(1, true);
}

// ---- split-K finish: y = sum_k part[k] + bias (stored as T), and this layer's InstanceNorm sums ----
// block = 256 threads = VL voxel lanes x G channel groups of 4; each thread walks ITER voxels.
template <typename T>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ part, int ksplit, int N, long vox,
                                                            int cout_pad, int Cout, const float* __restrict__ bias,
                                                            T* __restrict__ y, int Cout_stride, int Cout_off,
                                                            stat_t* stats, int G, int VL, int ITER) {
  using part_t = typename std::conditional<sizeof(T) == 4, double, float>::type;       // see the convolution epilogue
  __shared__ part_t red[256][8];
  const int n = blockIdx.y;
  const int cg = threadIdx.x % G, vl = threadIdx.x / G;
  const int c = cg * 4;
  part_t s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
  if (c < Cout) b4 = *(const f32x4*)(bias + c);      // Cout is a multiple of 8, c of 4: bias needs Cout entries only
  if (vl < VL && ITER == 2 && ksplit <= 8) {
    // both voxels of the thread are requested before the first is stored (a thread that takes them in turn waits for the
    // acknowledgement of its first store before its second batch of loads counts as landed: vmcnt is in issue order)
    const long va = ((long)blockIdx.x * 2) * VL + vl, vb = va + VL;
    const long kstride = (long)N * vox * cout_pad;
    const float* pa = part + ((long)n * vox + (va < vox ? va : 0)) * cout_pad + c;
    const float* pb = part + ((long)n * vox + (vb < vox ? vb : 0)) * cout_pad + c;
    f32x4 qa[8], qb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kk = j < ksplit ? j : ksplit - 1;
      qa[j] = *(const f32x4*)(pa + kk * kstride);
      qb[j] = *(const f32x4*)(pb + kk * kstride);
    }
    f32x4 acc[2] = {b4, b4};
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < ksplit)
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[0][e] += qa[j][e]; acc[1][e] += qb[j][e]; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long v = i ? vb : va;
      if (v >= vox) break;
      T o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (T)acc[i][e];
        const float f = (float)o[e];
        s[e] += f; q[e] += (part_t)f * (part_t)f;
      }
      if (c < Cout) {
        T* yp = y + ((long)n * vox + v) * Cout_stride + Cout_off + c;
#pragma unroll
        for (int e = 0; e < 4; ++e) yp[e] = o[e];
      }
    }
  } else if (vl < VL) {
    for (int i = 0; i < ITER; ++i) {
      const long v = ((long)blockIdx.x * ITER + i) * VL + vl;
      if (v >= vox) break;
      f32x4 acc = b4;
      const float* pp = part + ((long)n * vox + v) * cout_pad + c;
      const long kstride = (long)N * vox * cout_pad;
      // The partial tiles of one output element are read in batches of up to 16 independent 16-byte loads (one memory round
      // trip per batch: the kernel is a few microseconds of dependent latency, not bandwidth), and summed in a fixed order.
      int k = 0;
      for (; k + 16 <= ksplit; k += 16) {
        f32x4 pv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) pv[j] = *(const f32x4*)(pp + (k + j) * kstride);
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] += pv[j][e];
      }
      if (k < ksplit) {
        f32x4 pv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int kk = k + j < ksplit ? k + j : ksplit - 1;          // clamped address, masked use: branch-free loads
          pv[j] = *(const f32x4*)(pp + kk * kstride);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (k + j < ksplit)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += pv[j][e];
      }
      T o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (T)acc[e];
        const float f = (float)o[e];
        s[e] += f; q[e] += (part_t)f * (part_t)f;
      }
      if (c < Cout) {
        T* yp = y + ((long)n * vox + v) * Cout_stride + Cout_off + c;
#pragma unroll
        for (int e = 0; e < 4; ++e) yp[e] = o[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[threadIdx.x][e] = s[e]; red[threadIdx.x][4 + e] = q[e]; }
  __syncthreads();
  // one lane per CHANNEL for the statistics words: consecutive lanes add to consecutive words (one 512-byte atomic
  // instruction per word and wave; the per-channel-group form issued 16 strided ones per wave)
  for (int ch = threadIdx.x; ch < 4 * G; ch += 256) {
    const int g4 = ch >> 2, e = ch & 3;
    double S = 0, Q = 0;
    for (int j = 0; j < VL; ++j) { S += (double)red[j * G + g4][e]; Q += (double)red[j * G + g4][4 + e]; }
    if (ch < Cout) stats_add(stats, n, cout_pad, blockIdx.x & (STAT_REPLICAS - 1), ch, S, Q);
  }
}


static inline void choose_split(int base_wgs, int units, int* ksplit, int* ups, int target = 320, int max_base = 64) {
  *ksplit = 1; *ups = units;
  if (base_wgs > max_base || units <= 1) return;  // 24^3 and up: the partial-tile round trip costs more than it buys
  int want = target == 320 ? (320 + base_wgs - 1) / base_wgs : target / base_wgs;   // ~one workgroup per CU, each keeping >= a few units of work
  if (want < 1) want = 1;
  if (want > units) want = units;
  *ups = (units + want - 1) / want;
  *ksplit = (units + *ups - 1) / *ups;
}

// A background launch (dua_conv3_desc.background) runs on a second stream UNDER other launches: it asks for enough
// extra LDS that only ONE of its workgroups fits a CU, so that every CU keeps 64 KB and half its wave slots free for the
// main stream's workgroups (two of them per CU leave no room: the co-running launches then wait for retiring workgroups).
constexpr int PARTIAL_LDS_PAD = 36 * 1024;

// Kernels of this file that ask for more than 64 KB of dynamic LDS (raised once per device by ensure_prepared(), common.hpp)
static const LdsAttr kConvLdsAttrs[] = {
    {(const void*)conv3d_k3_v2_kernel<f16, 4>, c3v2::LDS_MAIN + 3 * 4 * 1024 + PARTIAL_LDS_PAD},
    {(const void*)conv3d_k3_v2_kernel<float, 4>, c3v2::LDS_MAIN + 3 * 4 * 1024 + PARTIAL_LDS_PAD},
    {(const void*)conv3d_k3_v2_kernel<f16, 4, 1>, c3v2::LDS_MAIN + 4096 + PARTIAL_LDS_PAD},
    {(const void*)conv3d_k3_v2_kernel<f16, 4, 0>, c3v2::LDS_MAIN + 4096 + PARTIAL_LDS_PAD},
    {(const void*)conv3d_k3_v2_kernel<f16, 4, 2, true>, c3v2::LDS_MAIN + 3 * 4 * 1024 + PARTIAL_LDS_PAD},
    {(const void*)conv3d_k3_first_kernel, c3f::LDS_F + 2048 + PARTIAL_LDS_PAD},
    {(const void*)conv3d_k3_v2_kernel<f16, 2>, c3v2::LDS_MAIN + 3 * 4 * 1024},
    {(const void*)conv3d_k3_v2_kernel<float, 2>, c3v2::LDS_MAIN + 3 * 4 * 1024},
    {(const void*)conv3d_k3_v2_kernel<f16, 4, 2, false, true>, 160 * 1024},
    {(const void*)conv3d_k3_v2_kernel<float, 4, 2, false, true>, 160 * 1024},
    {(const void*)conv3d_k3_v2_kernel<f16, 2, 2, false, true>, 160 * 1024},
    {(const void*)conv3d_k3_v2_kernel<float, 2, 2, false, true>, 160 * 1024},
};
static const LdsAttrs kConvLdsReg(kConvLdsAttrs);

// Which kernel a launch takes (also exported: dua_conv3d_k3_kernel_kind): 1 = the resident-weight first-layer kernel, 2 = the
// wide-tile form, 0 = conv3d_k3_v2_kernel in one of its launch shapes.
static int conv3_kernel_kind(const dua_conv3_desc* d, bool fused) {
  const int g_conv_variant = conv_variant_of(d);
  if (d->dtype != DUA_F16) return 0;
  if (d->tap_channel_plus1 > 0) return (d->tap_channel_plus1 == 17 && g_conv_variant == 0) ? 1 : 0;
  const long tiles = (long)((d->D + 3) / 4) * ((d->H + 7) / 8) * ((d->W + 7) / 8) * ((d->Cout + c3::BN - 1) / c3::BN) * d->N;
  const long vox = (long)d->D * d->H * d->W;
  if ((g_conv_variant == 0 || g_conv_variant == 8 || g_conv_variant == 9) && !d->background && tiles >= 1024 && d->D % 8 == 0 && d->H % 8 == 0 && d->W % 8 == 0 &&
      d->Cin % 16 == 0 && d->Cin <= (fused ? 256 : 384) && vox * d->Cin_stride < 0x7fffffffL)
    return 2;
  return 0;
}

// the owner of a data-gradient launch's output (dua_conv3d_k3_dgrad_reduce)
struct BwdSums { const void* raw; int stride, off; const dua_in_norm* in; double* sums; };

template <typename T>
static int launch_conv3(const dua_conv3_desc* d, const void* x, const void* w, const float* bias,
                        const dua_in_norm* in, void* y, stat_t* stats, float* ws, long ws_bytes, hipStream_t s,
                        const BwdSums* bw = nullptr) {
  using namespace c3;
  constexpr int CK = KG * Elem<T>::EPG;
  Conv3Args a;
  a.x = x; a.w = w; a.bias = bias; a.y = y; a.stats = stats;
  a.bw_raw = nullptr; a.bw_stride = a.bw_off = 0; a.bw_xf = make_xform(nullptr, d->Cout); a.bw_sums = nullptr;
  if (bw) {
    a.bw_raw = bw->raw; a.bw_stride = bw->stride; a.bw_off = bw->off; a.bw_xf = make_xform(bw->in, d->Cout); a.bw_sums = bw->sums;
  }
  a.xf = make_xform(in, d->Cin);
  a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W;
  a.Cin = d->Cin; a.Cin_stride = d->Cin_stride; a.Cin_off = d->Cin_off;
  a.Cout = d->Cout; a.Cout_stride = d->Cout_stride; a.Cout_off = d->Cout_off;
  a.nchunks = (d->Cin + CK - 1) / CK;
  const int td = (d->D + TD - 1) / TD;
  a.tiles_h = (d->H + TH - 1) / TH; a.tiles_w = (d->W + TW - 1) / TW;
  a.ntiles = td * a.tiles_h * a.tiles_w;
  const int nct = (d->Cout + BN - 1) / BN;
  a.cout_pad = nct * BN;
  a.ksplit = 1; a.units_per_split = a.nchunks * 3; a.part = nullptr; a.tap_ch = -1;
  a.in_blk = d->layout & DUA_IN_BLOCKED ? 1 : 0; a.out_blk = d->layout & DUA_OUT_BLOCKED ? 1 : 0;
  if (a.nchunks * CK > 1024 || !conv_policy_ok(d)) return DUA_ERR_ARG;
  const int g_conv_variant = conv_variant_of(d);
  const int kind = conv3_kernel_kind(d, in && in->stats);
  // 16-channel-blocked buffers: read by the wide-tile form only, written by it and by the first-layer kernel only
  if ((a.in_blk && kind != 2) || (a.out_blk && kind == 0)) return DUA_ERR_ARG;
  if ((a.in_blk && (d->Cin_off % 16 || d->Cin_stride % 16)) || (a.out_blk && (d->Cout_off % 16 || d->Cout_stride % 16))) return DUA_ERR_ARG;
  const int xf_bytes = in && in->stats ? 3 * 4 * a.nchunks * CK : 0;
  if (int e = ensure_prepared()) return e;
  const long vox = (long)d->D * d->H * d->W;
  const int bg_pad = d->background ? PARTIAL_LDS_PAD : 0;     // one workgroup per CU, see PARTIAL_LDS_PAD
  if (d->tap_channel_plus1 > 0) {
    // single-channel tap form (see the kernel): fp16, one Cin chunk, no fused input transform, 16 * NKS ordinary
    // channels in front of the tap channel, zero padding behind it; weights from dua_pack_conv3_weights_tap
    a.tap_ch = d->tap_channel_plus1 - 1;
    if constexpr (sizeof(T) != 2) return DUA_ERR_ARG;
    else {
      if (a.nchunks != 1 || (in && in->stats) || (a.tap_ch != 0 && a.tap_ch != 16) || d->Cin != a.tap_ch + 8) return DUA_ERR_ARG;
      dim3 grid(a.ntiles, nct, d->N);
      if (kind == 1) {
        // resident-weight form: two persistent workgroups per CU walk the (sample, tile) items; a background launch takes one
        // per CU (and the LDS pad that keeps a second one off the CU)
        const int cus = device_cus();
        if (cus <= 0) return DUA_ERR_ARG;
        const int items = a.ntiles * d->N;
        const int wgs = std::max(1, (bg_pad ? 1 : 2) * cus / nct);
        hipLaunchKernelGGL(conv3d_k3_first_kernel, dim3(std::min(items, wgs), nct, 1), dim3(256), c3f::LDS_F + 2048 + bg_pad, s, a, items);
        return (int)hipGetLastError();
      }
      if (a.tap_ch == 16) hipLaunchKernelGGL((conv3d_k3_v2_kernel<T, 4, 1>), grid, dim3(256), c3v2::LDS_MAIN + 4096 + bg_pad, s, a);
      else hipLaunchKernelGGL((conv3d_k3_v2_kernel<T, 4, 0>), grid, dim3(256), c3v2::LDS_MAIN + 4096 + bg_pad, s, a);
      return (int)hipGetLastError();
    }
  }
  // wide-tile form (conv3d_wide.hip): fp16 layers with tiles to spare (96^3); variant 7 keeps them on the 4x8x8 kernel (A/B)
  if constexpr (sizeof(T) == 2) {
    // 8: persistent workgroups with the accumulators held by name (round 5, conv3d_k3_wide_pt_kernel), 9: the same with the
    // odd-slot workgroup of a CU starting two sleeps late -- both measured 1.5-2.5 % SLOWER than one tile per workgroup
    // (profiles/r5_conv_wide_persistent_named_acc_ab.txt) and kept for that A/B only
    if (kind == 2) return launch_conv3_wide(a, d->D, s, g_conv_variant >= 8, g_conv_variant == 9 ? 2 : 0);
  }
  if (bw) return DUA_ERR_ARG;                                  // only the wide-tile form has the backward-sums epilogue
  const bool autop = g_conv_variant == 0 || g_conv_variant == 2 || g_conv_variant == 6 || g_conv_variant == 7 || g_conv_variant >= 8;       // the automatic policy; 6 = without the kd-plane form, 7 = without the wide-tile form (A/B)
  const bool big = g_conv_variant == 0 || g_conv_variant == 2 || g_conv_variant == 7 || g_conv_variant >= 8;          // kd-plane form for the layers that cannot put two workgroups on every CU
  if (ws != nullptr && autop) {
    int ks, ups;
    if (g_conv_variant == 2) choose_split(a.ntiles * nct * d->N, a.nchunks * 3, &ks, &ups, 512, 256);    // A/B: K split up to 256 base workgroups
    else choose_split(a.ntiles * nct * d->N, a.nchunks * 3, &ks, &ups, big ? 256 : 320);
    if (ks > 1 && (long)ks * d->N * vox * a.cout_pad * 4 <= ws_bytes) { a.ksplit = ks; a.units_per_split = ups; a.part = ws; }
  }
  // 24^3-sized layers (too few 4x8x8 tiles for 256 CUs, too big for split-K to pay): 2x8x8 tiles, twice the workgroups
  if (a.ksplit == 1 && ((autop && a.ntiles * nct * d->N < 200 && a.ntiles * nct * d->N > 64) || g_conv_variant == 3)) {
    const int td2 = (d->D + 1) / 2;
    a.ntiles = td2 * a.tiles_h * a.tiles_w;
    dim3 grid2(a.ntiles, nct, d->N);
    constexpr int LDS2 = 4 * c3::HH * c3::RS + 2 * c3v2::SLAB;
    // the kd-plane form (nine slabs resident: one workgroup per CU) only while every workgroup has a CU of its own; with more of
    // them (12^3 at batch 4: 384) two slab-pipeline workgroups per CU are faster (64.3 vs 79.8 us on 256 -> 256, tools/bench_conv.py)
    const bool big2 = big && (long)a.ntiles * nct * d->N <= 256;
    if (big2) hipLaunchKernelGGL((conv3d_k3_v2_kernel<T, 2, 2, false, true>), grid2, dim3(256), 4 * c3::HH * c3::RS + 9 * c3v2::SLAB + xf_bytes, s, a);
    else hipLaunchKernelGGL((conv3d_k3_v2_kernel<T, 2>), grid2, dim3(256), LDS2 + xf_bytes, s, a);
    return (int)hipGetLastError();
  }
  dim3 grid(a.ntiles, nct, d->N * a.ksplit);
  if (a.ksplit > 1) {
    if (big) hipLaunchKernelGGL((conv3d_k3_v2_kernel<T, 4, 2, false, true>), grid, dim3(256), c3::HALO_BYTES + 9 * c3v2::SLAB + xf_bytes, s, a);
    else hipLaunchKernelGGL(conv3d_k3_v2_kernel<T>, grid, dim3(256), c3v2::LDS_MAIN + xf_bytes, s, a);
    if (d->policy & DUA_POLICY_NO_FINISH) return (int)hipGetLastError();
    const int G = a.cout_pad / 4 > 256 ? 256 : a.cout_pad / 4;       // channel groups handled per block pass
    if (a.cout_pad / 4 > 256) return DUA_ERR_ARG;
    const int VL = 256 / G;
    int ITER = 8;
    while (ITER > 1 && (vox + (long)VL * ITER - 1) / ((long)VL * ITER) < 128) ITER >>= 1;   // >= ~128 blocks; every block ends with
                                                                                            // 16 atomic instructions (measured: 432
                                                                                            // blocks 14.8 us, 216 blocks 10.9 us at 12^3)
    dim3 fgrid((unsigned)((vox + (long)VL * ITER - 1) / ((long)VL * ITER)), d->N);
    hipLaunchKernelGGL(splitk_finish_kernel<T>, fgrid, dim3(256), 0, s, (const float*)ws, a.ksplit, d->N, vox, a.cout_pad,
                       d->Cout, bias, (T*)y, d->Cout_stride, d->Cout_off, stats, G, VL, ITER);
    return (int)hipGetLastError();
  }
  if (sizeof(T) == 2 && d->Cin - (a.nchunks - 1) * CK <= CK / 2)       // e.g. Cin = 48: the last chunk is half padding
    hipLaunchKernelGGL((conv3d_k3_v2_kernel<T, 4, 2, true>), grid, dim3(256), c3v2::LDS_MAIN + xf_bytes + bg_pad, s, a);
  else
    hipLaunchKernelGGL(conv3d_k3_v2_kernel<T>, grid, dim3(256), c3v2::LDS_MAIN + xf_bytes + bg_pad, s, a);
  return (int)hipGetLastError();
}

}  // namespace dua

extern "C" {

#ifdef DUA_STAMP
// diagnostic build: copy the stamp buffer out ([8192 workgroups][64 slots] of 64-bit counters) and clear it
long dua_debug_stamps(void* host, long bytes) { return dua::stamps_out(host, bytes); }
#endif

#ifdef DUA_ABLATE
// diagnostic builds only (tools/build_diag.sh ... -DDUA_ABLATE): ablation mask of the weight-gradient kernel
int dua_set_option(int key, int value) {
  if (key == 3 && value >= 0 && value < 16) { dua::g_wgrad_abl = value; return 0; }
  return DUA_ERR_ARG;
}
#endif

int dua_conv3d_k3_kernel_kind(const dua_conv3_desc* d, int fused, int has_workspace) {
  if (!d || !dua::conv_policy_ok(d)) return DUA_ERR_ARG;
  (void)has_workspace;              // split-K applies below 1024 tiles only, where the answer is 0 anyway
  return dua::conv3_kernel_kind(d, fused != 0);
}

long dua_conv3d_k3_workspace(const dua_conv3_desc* d) {
  using namespace dua::c3;
  if (!d || (d->dtype != DUA_F16 && d->dtype != DUA_F32)) return DUA_ERR_ARG;
  const int ck = 4 * (d->dtype == DUA_F16 ? 8 : 4);
  const int nch = (d->Cin + ck - 1) / ck, nct = (d->Cout + BN - 1) / BN;
  const int tiles = ((d->D + TD - 1) / TD) * ((d->H + TH - 1) / TH) * ((d->W + TW - 1) / TW);
  int ks, ups;
  dua::choose_split(tiles * nct * d->N, nch * 3, &ks, &ups);
  return ks > 1 ? (long)ks * d->N * d->D * d->H * d->W * nct * BN * 4 : 0;
}

int dua_conv3d_k3_fwd(const dua_conv3_desc* d, const void* x, const void* w_packed, const float* bias_padded,
                      const dua_in_norm* in, void* y, dua_stat_word* out_stats, void* workspace, long workspace_bytes,
                      void* stream) {
  if (!d || !x || !w_packed || !bias_padded || !y || !out_stats) return DUA_ERR_ARG;
  if (d->Cin % 8 || d->Cout % 8 || d->Cin_stride % 8 || d->Cout_stride % 8 || d->Cin_off % 8 || d->Cout_off % 8)
    return DUA_ERR_ARG;
  if (in && in->stats && (!in->gamma || !in->beta || in->c_pad < d->Cin || in->count <= 0 || !(in->slope >= 0.f && in->slope <= 1.f))) return DUA_ERR_ARG;
  if (d->dtype == DUA_F16)
    return dua::launch_conv3<dua::f16>(d, x, w_packed, bias_padded, in, y, out_stats, (float*)workspace, workspace_bytes, (hipStream_t)stream);
  if (d->dtype == DUA_F32)
    return dua::launch_conv3<float>(d, x, w_packed, bias_padded, in, y, out_stats, (float*)workspace, workspace_bytes, (hipStream_t)stream);
  return DUA_ERR_ARG;
}

int dua_conv3d_k3_dgrad_reduce_supported(const dua_conv3_desc* d) {
  if (!d || d->dtype != DUA_F16 || d->layout || d->tap_channel_plus1 || d->background || dua::conv_variant_of(d) != 0) return 0;
  return dua::conv3_kernel_kind(d, false) == 2 ? 1 : 0;
}

int dua_conv3d_k3_dgrad_reduce(const dua_conv3_desc* d, const void* dy, const void* w_packed, const float* bias_padded, void* dx,
                               const void* raw, int raw_stride, int raw_off, const dua_in_norm* raw_in, double* sums, void* stream) {
  if (!dua_conv3d_k3_dgrad_reduce_supported(d) || !dy || !w_packed || !bias_padded || !dx || !raw || !raw_in || !sums) return DUA_ERR_ARG;
  if (d->Cin % 8 || d->Cout % 8 || d->Cin_stride % 8 || d->Cout_stride % 8 || d->Cin_off % 8 || d->Cout_off % 8) return DUA_ERR_ARG;
  if (!raw_in->stats || !raw_in->gamma || !raw_in->beta || raw_in->c_pad < d->Cout || raw_in->count <= 0 ||
      !(raw_in->slope >= 0.f && raw_in->slope <= 1.f) || raw_stride % 8 || raw_off % 8 || raw_off + d->Cout > raw_stride)
    return DUA_ERR_ARG;
  const dua::BwdSums bw{raw, raw_stride, raw_off, raw_in, sums};
  return dua::launch_conv3<dua::f16>(d, dy, w_packed, bias_padded, nullptr, dx, nullptr, nullptr, 0, (hipStream_t)stream, &bw);
}

}  // extern "C"
