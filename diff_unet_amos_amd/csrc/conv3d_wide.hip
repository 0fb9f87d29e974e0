// 3x3x3 convolution, wide-tile form (round 4): 8 MFMA accumulators per wave.
//
// Same operation, operands and packed weights as conv3d_k3_v2_kernel (conv3d_igemm.hip: nn.Conv3d k3 p1 of MONAI's
// Convolution block, models/basic_unet/denoiser.py:56-59, with the producer's InstanceNorm + LeakyReLU + temb add fused
// into the halo staging and this layer's InstanceNorm sums in the epilogue), for the fp16 layers that have tiles to spare
// (96^3): there a 4x8x8-voxel workgroup spends 19 000 of its 52 000 cycles outside the K loop (statistics preamble, halo
// staging + transform, epilogue; in-kernel stamps), and two such workgroups per CU keep the matrix pipes 53 %
// busy.  Here a workgroup owns 8x8x8 voxels x 64 output channels and a wave 128 voxels x 64 channels = 4 x 2
// accumulators of 32x32:
//   * per k-step 6 fragment reads feed 8 MFMAs (4 per 4 before): 768 B of LDS reads per MFMA instead of 1024;
//   * weights, barriers and the statistics preamble are paid once per 512 voxels instead of once per 256;
//   * the halo is 1000 voxels for 512 outputs (1.95 per output) instead of 600 for 256 (2.34).
// Two workgroups per CU must still fit (one alone leaves the pipes idle during every prologue / epilogue), i.e. <= 80 KB
// of LDS and <= 256 registers per lane with 128 of them accumulators:
//   * the input is walked in HALF chunks of 16 channels: 32-byte halo voxels (12 slots per 10-voxel row, the two 16-byte
//     halves swapped on odd rows: the layout of the first-layer kernel, every ds_read_b128 fragment read conflict-free),
//     38.4 KB for the 10x10x10 halo; one k-step per tap;
//   * weights arrive as kd planes of a half chunk (9 taps x 2 k-groups x 1 KB = 18 KB) by LDS-DMA straight out of the
//     packed slab layout into a ring of two: no staging registers, a phase is 72 MFMAs per wave between two barriers;
//   * operand fragments are prefetched half a k-step ahead (the B pair double-buffered, the A pairs of the two depth
//     slices alternate): 32 registers;
//   * the next half chunk's halo (8 pieces of 16 bytes per thread) is requested at the start of a half chunk's last
//     phase and transformed + stored after it.
// Tile count: 96^3 gives 1728 such tiles for 512 resident workgroups (3.4 rounds); ending the launch with 4x8x8 tiles of the
// same code (template MB = 2) to fill the last round was built and measured: the all-wide launch is faster
// (profiles/r4_conv_wide_tile_ab.txt), so the kernel has the one form.
#include "common.hpp"
#include "../../include/dua_hip.h"
#include "conv3_args.hpp"
#include "stamp.hpp"
#include "named_acc.hpp"

namespace dua {

namespace c3w {
constexpr int TH = 8, TW = 8, HH = TH + 2, HW = TW + 2;
constexpr int VSF = 32, RSF = 12 * VSF, PSF = HH * RSF;      // 384-byte rows, 3840-byte planes
constexpr int BN = 64;
constexpr int WPLANE = 18 * 1024;                              // [9 taps][2 k-groups][64 couts][16 B]
constexpr int HALO_MAX = 10 * PSF;                             // 38400: the 8-deep tile
constexpr int LDS_FIXED = HALO_MAX + 2 * WPLANE;               // 75264
constexpr int SLAB = 3 * 4 * BN * 16;                          // 12288: one (kd, kh) slab of the packed weights (32-channel chunk)
}  // namespace c3w

// one 1 KB piece global -> LDS (64 lanes x 16 B, LDS address = M0 base + lane * 16); M0 saved / restored in the statement
__device__ __forceinline__ void dma_piece(const char* src_lane, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src_lane), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}

#ifndef WIDE_LA
#define WIDE_LA 1         // half-steps of A fragments in flight ahead of the MFMAs of the K loop
#endif
// MB = 32-voxel blocks per wave: 4 (8x8x8 tile, wave = depth slices 2w and 2w+1) or 2 (4x8x8 tile, wave = depth slice w)
template <int MB, bool BWD = false>
__device__ __forceinline__ void wide_tile(const Conv3Args& a, char* smem, const int d0, const int h0, const int w0,
                                          const int ct, const int n, const int replica) {
  using namespace c3w;
  using T = f16;
  constexpr int TD = 2 * MB, HD = TD + 2, NH = MB / 2;         // NH half-steps (pairs of blocks) per tap
  char* halo = smem;
  char* wbuf = smem + HALO_MAX;
  float* xsc = (float*)(smem + LDS_FIXED);
  float* xsh = xsc + a.Cin;                                    // Cin is a multiple of 16 here
  float* xad = xsh + a.Cin;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const bool fused = a.xf.stats != nullptr;
  const int nhc = (a.Cin + 15) >> 4;                           // half chunks
  const int U = nhc * 3;

  // ---- halo pieces of this thread: thread = (position (hy, hx) inside a halo plane, 16-byte half p) -- 200 of the 256 threads
  // -- and piece j = halo plane j: the global address advances by one plane of the volume and the LDS address by one halo
  // plane per piece (immediate offsets), and whether plane j lies inside the volume is wave-uniform.  (Dealing the 2000
  // pieces out evenly, 8 per thread, cost ~40 address instructions per piece, at every half chunk.) ----
  const int p_t = tid & 1, pos = tid >> 1;
  const int hy = pos / HW, hx = pos - hy * HW;
  const int gh = h0 + hy - 1, gw = w0 + hx - 1;
  const bool ok_hw = pos < HH * HW && (unsigned)gh < (unsigned)a.H && (unsigned)gw < (unsigned)a.W;
  // input layout: channels-last rows [voxel][Cin_stride], or (in_blk) 16-channel blocks [Cin_stride / 16][voxel][16]: there a
  // half chunk is a contiguous 32 bytes per voxel and consecutive voxels are adjacent -- a piece request touches ~10 cache lines
  // instead of 32 and no line is fetched four times
  const int vstride = a.in_blk ? 16 : a.Cin_stride;
  const long hcstride = a.in_blk ? (long)a.D * a.H * a.W * 16 : 16;
  const T* xin = (const T*)a.x + (long)n * a.D * a.H * a.W * a.Cin_stride + (a.in_blk ? (long)(a.Cin_off >> 4) * a.D * a.H * a.W * 16 : a.Cin_off) + p_t * 8;
  const int voff = ok_hw ? (((d0 - 1) * a.H + gh) * a.W + gw) * vstride : 0;      // of halo plane 0 (never read where it is outside)
  const int pstep = a.H * a.W * vstride;
  // threads without a position store into the unused voxel slots 10 / 11 of row 0 of each plane
  const int loff = pos < HH * HW ? hy * RSF + hx * VSF + ((p_t ^ (hy & 1)) << 4) : HW * VSF + (p_t << 4);
  f16x8 hreg[HD];
  auto load_halo = [&](int hc, int j0, int j1) {
    const T* src = xin + hc * hcstride;
#pragma unroll
    for (int j = 0; j < HD; ++j) {
      if (j < j0 || j >= j1) continue;
      const bool dok = (unsigned)(d0 + j - 1) < (unsigned)a.D;       // wave-uniform
      hreg[j] = *(const f16x8*)(src + (ok_hw && dok ? voff + j * pstep : 0));
    }
  };
  auto store_halo = [&](int hc) {
    float sc[8], sh[8], ad[8], sn[8];
    if (fused) {
      const int c0 = hc * 16 + p_t * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc[e] = xsc[c0 + e]; sh[e] = xsh[c0 + e]; ad[e] = xad[c0 + e]; }
      xform_prep<T>(sc, sh, ad, sn, a.xf.slope);
    }
#pragma unroll
    for (int j = 0; j < HD; ++j) {
      f16x8 v = hreg[j];
      if (fused) v = xform_frag<T>(v, sc, sh, ad, sn, a.xf.slope);
      const f32x4 raw = __builtin_bit_cast(f32x4, v);
      const bool ok = ok_hw && (unsigned)(d0 + j - 1) < (unsigned)a.D;
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = ok ? raw[e] : 0.f;
      *(f32x4*)(halo + loff + j * PSF) = o;
    }
  };

  // ---- weights: kd plane `kd` of half chunk `hc` -> ring slot (18 pieces of 1 KB; wave w takes pieces w, w + 4, ...) ----
  const char* wsrc = (const char*)a.w + (long)ct * a.nchunks * 9 * SLAB + lane * 16;
  const unsigned wlds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)wbuf;
  auto dma_plane = [&](int u, int slot) {
    const int hc = u / 3, kd = u - hc * 3;
    const char* src = wsrc + (long)((hc >> 1) * 3 + kd) * 3 * SLAB + (hc & 1) * 2048;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int p = wave + 4 * j;                              // piece = (tap p >> 1, k-group p & 1)
      if (p < 18) dma_piece(src + ((p >> 1) * 4 + (p & 1)) * 1024, wlds + slot * WPLANE + p * 1024);
    }
  };

  // ---- prologue.  Everything that must come from memory is requested before anything waits, oldest first in the order it
  // is needed (vmcnt retires in order): bias, the producer's statistics words (fused input), the halo, the first weight plane.
  DUA_STAMP_AT(0, true);
  DUA_STAMP_AT(2, false);
  float bias_q[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int bc = ct * BN + q * 32 + r;
    bias_q[q] = bc >= a.Cout ? 0.f : a.bias[bc];
  }
  // scale / shift / add tables of the fused input transform: the arithmetic of xform_preamble (common.hpp) with its loads
  // split off, 16 channels per (wave, group u), up to UN groups per wave = 256 channels
  constexpr int UN = 4;
  stat_t sv[UN][2 * STAT_WORDS];
  float gam[UN], bet[UN], addv[UN];
  if (fused) {
    const int part = lane >> 4;
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int c = (wave + 4 * u) * 16 + (lane & 15), cc = c < a.Cin ? c : a.Cin - 1;
      gam[u] = a.xf.gamma[cc]; bet[u] = a.xf.beta[cc];
      addv[u] = a.xf.add ? a.xf.add[(long)n * a.xf.add_stride + cc] : 0.f;
      const stat_t* sp = a.xf.stats + ((long)n * STAT_REPLICAS + 2 * part) * STAT_WORDS * a.xf.c_pad + cc;
      // one block per group (wave-uniform condition): inside a per-load conditional hipcc waits for every load on its own
      if ((wave + 4 * u) * 16 < a.Cin) {
#pragma unroll
        for (int k = 0; k < 2 * STAT_WORDS; ++k) sv[u][k] = sp[(long)k * a.xf.c_pad];
      } else {
#pragma unroll
        for (int k = 0; k < 2 * STAT_WORDS; ++k) sv[u][k] = 0;
      }
    }
  }
  // backward-sums mode: mean / rstd / gamma / beta of the 64 channels this tile writes, from the forward statistics of the layer
  // that owns them (wave w: channels 16 w ...), into 1 KB behind the transform tables
  float* bwc = (float*)(smem + LDS_FIXED) + (fused ? 3 * a.Cin : 0);      // [4][64]: mean, rstd, gamma, beta
  double bwS = 0, bwQ = 0;
  float bwg = 0.f, bwb = 0.f;
  constexpr bool bwd = BWD;                                     // its own kernel: in one kernel with the forward epilogue hipcc spilled 33 registers of BOTH
  if (bwd) {
    const int c = ct * BN + wave * 16 + (lane & 15), cc = c < a.Cout ? c : a.Cout - 1;
    bwg = a.bw_xf.gamma[cc]; bwb = a.bw_xf.beta[cc];
    stats_read_wave16(a.bw_xf.stats, n, a.bw_xf.c_pad, cc, bwS, bwQ);
  }
  DUA_STAMP_AT(56, false);
  load_halo(0, 0, HD);
  DUA_STAMP_AT(57, false);
  dma_plane(0, 0);
  DUA_STAMP_AT(59, false);
  if (fused) {
    const int part = lane >> 4;
    double Sm = 0, Qm = 0;
    float gm = 0.f, bm = 0.f, am = 0.f;
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      stat_t w[STAT_WORDS];
#pragma unroll
      for (int k = 0; k < STAT_WORDS; ++k) {
        w[k] = sv[u][k] + sv[u][STAT_WORDS + k];
        w[k] += __shfl_xor(w[k], 16);
        w[k] += __shfl_xor(w[k], 32);
      }
      if (part == u) {                     // the 16-lane part u finishes group u: one reciprocal square root per lane
        Sm = (double)w[0] + (double)w[1] * (1.0 / STAT_FRAC);
        Qm = (double)w[2] + (double)w[3] * (1.0 / STAT_FRAC);
        gm = gam[u]; bm = bet[u]; am = addv[u];
      }
    }
    const int c = (wave + 4 * part) * 16 + (lane & 15);
    const double mean = Sm * a.xf.inv_count;
    double var = Qm * a.xf.inv_count - mean * mean;
    var = var > 0 ? var : 0;
    const float g = gm * (float)(1.0 / sqrt(var + (double)a.xf.eps));
    if (c < a.Cin) {
      xsc[c] = g;
      xsh[c] = bm - (float)mean * g;
      xad[c] = am;
    }
    __syncthreads();
  }
  if (bwd && lane < 16) {
    const double mean = bwS * a.bw_xf.inv_count;
    double var = bwQ * a.bw_xf.inv_count - mean * mean;
    var = var > 0 ? var : 0;
    const int cl = wave * 16 + lane;
    bwc[cl] = (float)mean;
    bwc[64 + cl] = (float)(1.0 / sqrt(var + (double)a.bw_xf.eps));
    bwc[128 + cl] = bwg;
    bwc[192 + cl] = bwb;
  }
  DUA_STAMP_AT(60, false);
  DUA_STAMP_AT(61, false);
  store_halo(0);
  // ---- accumulators: [block m][cout half q]; block m = depth slice dsl + (m >> 1), h rows 4 (m & 1) .. +3; they start at
  // the bias ----
  const int dsl = MB == 4 ? 2 * wave : wave;
  f32x16 acc[MB][2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][q][i] = bias_q[q];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's weight pieces (the compiler does not count them)
  __syncthreads();
  DUA_STAMP_AT(3, false);

  const int sw = (r >> 3) & 1;
  const int a_base = dsl * PSF + (r >> 3) * RSF + (r & 7) * VSF;
  const int a_even = a_base + ((hh ^ sw) << 4), a_odd = a_base + ((hh ^ sw ^ 1) << 4);      // parity of halo row h + kh
  const int b_base = (hh * BN + r) * 16;

  // One phase = the 9 taps of kd plane `kd` of the current half chunk against ring slot `slot`: half-step h = (tap t = h / NH,
  // block pair sub = h % NH) is 4 MFMAs on A pair (t, sub) and B pair t; the fragments of half-step h + 1 are requested
  // before the MFMAs of half-step h issue.
  auto phase = [&](int kd, int slot) __attribute__((always_inline)) {
    const char* hp = halo + kd * PSF;
    const char* wb = wbuf + slot * WPLANE + b_base;
    f16x8 fa[WIDE_LA + 1][2], fb[2][2];                          // [buffer][block of the pair] / [buffer][cout half]
    auto ldA = [&](int t, int sub, int b) {
      const int kh = t / 3, kw = t - kh * 3;
      const char* ap = hp + ((kh & 1) ? a_odd : a_even) + sub * PSF + kh * RSF + kw * VSF;
      fa[b][0] = *(const f16x8*)ap;
      fa[b][1] = *(const f16x8*)(ap + 4 * RSF);
    };
    auto ldB = [&](int t, int b) {
      fb[b][0] = *(const f16x8*)(wb + t * 2048);
      fb[b][1] = *(const f16x8*)(wb + t * 2048 + 512);
    };
    ldB(0, 0);
#pragma unroll
    for (int h = 0; h < WIDE_LA; ++h) ldA(h / NH, h % NH, h);
#pragma unroll
    for (int h = 0; h < 9 * NH; ++h) {
      const int t = h / NH, sub = h % NH;
      if (h + WIDE_LA < 9 * NH) {
        const int t1 = (h + WIDE_LA) / NH, sub1 = (h + WIDE_LA) % NH;
        if (WIDE_LA == 1 && sub1 == 0) ldB(t1, t1 & 1);
        ldA(t1, sub1, (h + WIDE_LA) % (WIDE_LA + 1));
      }
      if (WIDE_LA > 1 && sub == 0 && t + 1 < 9) ldB(t + 1, (t + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);                         // keep the requests ahead of the MFMAs (hipcc sinks them otherwise)
      mma32(acc[2 * sub][0], fa[h % (WIDE_LA + 1)][0], fb[t & 1][0]);
      mma32(acc[2 * sub][1], fa[h % (WIDE_LA + 1)][0], fb[t & 1][1]);
      mma32(acc[2 * sub + 1][0], fa[h % (WIDE_LA + 1)][1], fb[t & 1][0]);
      mma32(acc[2 * sub + 1][1], fa[h % (WIDE_LA + 1)][1], fb[t & 1][1]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // The next half chunk's halo pieces (each touches 32 cache lines: ten back-to-back requests hold the wave for ~3 000 cycles)
  // are requested in three groups, at the head of the three phases of the current half chunk, behind that phase's weight
  // pieces: planes [0, J1), [J1, J2), [J2, HD).
  constexpr int J1 = HD == 10 ? 4 : 2, J2 = HD == 10 ? 7 : 4;
  for (int hc = 0; hc < nhc; ++hc) {
    const bool more = hc + 1 < nhc;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      const int u = hc * 3 + kd, slot = u & 1;
      if (u + 1 < U) dma_plane(u + 1, slot ^ 1);                // that slot was last read before the previous barrier
      if (more) load_halo(hc + 1, kd == 0 ? 0 : kd == 1 ? J1 : J2, kd == 0 ? J1 : kd == 1 ? J2 : HD);
      phase(kd, slot);
      // the next weight plane has landed (this wave's pieces; the barrier: everyone's); the halo pieces requested behind it may
      // still fly, except before the store that follows the last phase
      if (more && kd == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(J1) : "memory");
      else if (more && kd == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(J2 - J1) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (u < 26) DUA_STAMP_AT(4 + u, false);
    }
    if (more) {
      store_halo(hc + 1);
      if (hc < 7) DUA_STAMP_AT(30 + 2 * hc, false);
      __syncthreads();
      if (hc < 7) DUA_STAMP_AT(31 + 2 * hc, false);
    }
  }
  DUA_STAMP_AT(62, false);

  // ---- epilogue: statistics from the fp32 accumulators; each wave stages TWO 32-voxel blocks at a time in rows of its own
  // (2 x 32 x 128 B: whole voxel lines leave in one store instruction) -- the loop above ended with a workgroup barrier, so the
  // halo is free; no workgroup barrier inside ----
  char* ot = halo + wave * 8192;
  float* ex = (float*)wbuf;                                      // [4 waves][64 couts][2]
  const long nvox = (long)a.D * a.H * a.W;
  T* yout = (T*)a.y + (long)n * nvox * a.Cout_stride;
  float s[2] = {0.f, 0.f}, ss[2] = {0.f, 0.f};
  if (bwd) {
    // ---- backward-sums mode (training: this launch is a data gradient and its output the dA of the layer whose raw output is
    // bw_raw).  The sums are taken where the rows leave: a lane of the store loop holds 8 channels of one voxel, so the layer's raw
    // output arrives as one 16-byte load beside the 16-byte store, and the rounded dA it stores is what in_bwd_apply will read. ----
    const int cg = lane & 7;
    float m8[8], r8[8], g8[8], b8[8], s0[8], s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      m8[e] = bwc[cg * 8 + e]; r8[e] = bwc[64 + cg * 8 + e]; g8[e] = bwc[128 + cg * 8 + e]; b8[e] = bwc[192 + cg * 8 + e];
      s0[e] = s1[e] = s2[e] = 0.f;
    }
    const float slope = a.bw_xf.slope;
    const T* rawn = (const T*)a.bw_raw + (long)n * nvox * a.bw_stride + a.bw_off + ct * BN + cg * 8;
    const bool cok = ct * BN + cg * 8 < a.Cout;
#pragma unroll
    for (int pr = 0; pr < MB / 2; ++pr) {
      const int gd = d0 + dsl + pr;
      f16x8 yv[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int v = it * 8 + (lane >> 3);
        const long vx = ((long)gd * a.H + h0 + (v >> 3)) * a.W + w0 + (v & 7);
        yv[it] = *(const f16x8*)(rawn + (cok ? vx * a.bw_stride : 0));
      }
#pragma unroll
      for (int mm = 0; mm < 2; ++mm)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            *(T*)(ot + mm * 4096 + acc_row(i, hh) * 128 + (q * 32 + r) * 2) = (T)acc[2 * pr + mm][q][i];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int v = it * 8 + (lane >> 3);
        const int gh = h0 + (v >> 3), gw = w0 + (v & 7);
        const f16x8 o = *(const f16x8*)(ot + v * 128 + cg * 16);
        if (cok) {
          *(f16x8*)(yout + chan_off(a.out_blk, ((long)gd * a.H + gh) * a.W + gw, a.Cout_off + ct * BN + cg * 8, a.Cout_stride, nvox)) = o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float zh = ((float)yv[it][e] - m8[e]) * r8[e];
            const float z = fmaf(zh, g8[e], b8[e]);
            const float dv = (float)o[e];
            const float dz = z > 0.f ? dv : dv * slope;
            s0[e] += dv; s1[e] += dz; s2[e] = fmaf(dz, zh, s2[e]);
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // lanes with the same channel group (8 apart), then the four waves in a fixed order, then one double atomic per (sum, channel)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int sft = 8; sft < 64; sft <<= 1) {
        s0[e] += __shfl_xor(s0[e], sft); s1[e] += __shfl_xor(s1[e], sft); s2[e] += __shfl_xor(s2[e], sft);
      }
    }
    __syncthreads();                                             // every wave is past its staging rows and the tables
    if (lane < 8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ex[((wave * BN) + lane * 8 + e) * 3] = s0[e];
        ex[((wave * BN) + lane * 8 + e) * 3 + 1] = s1[e];
        ex[((wave * BN) + lane * 8 + e) * 3 + 2] = s2[e];
      }
    }
    __syncthreads();
    if (tid < 3 * BN) {
      const int which = tid / BN, c = tid - which * BN;
      double acc_d = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) acc_d += (double)ex[(w * BN + c) * 3 + which];
      if (ct * BN + c < a.Cout)
        unsafeAtomicAdd(a.bw_sums + (((long)n * STAT_REPLICAS + replica) * a.bw_xf.c_pad + ct * BN + c) * 4 + which, acc_d);
    }
    DUA_STAMP_AT(63, false);
    DUA_STAMP_AT(1, true);
    return;
  }
#pragma unroll
  for (int pr = 0; pr < MB / 2; ++pr) {
#pragma unroll
    for (int mm = 0; mm < 2; ++mm)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float v = acc[2 * pr + mm][q][i];
          s[q] += v;
          ss[q] = fmaf(v, v, ss[q]);
          *(T*)(ot + mm * 4096 + acc_row(i, hh) * 128 + (q * 32 + r) * 2) = (T)v;
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int gd = d0 + dsl + pr;                                // blocks 2 pr and 2 pr + 1: depth slice dsl + pr, h rows 0..3 and 4..7
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int v = it * 8 + (lane >> 3), cg = lane & 7;         // v = 0..63: the 8 x 8 voxels of the depth slice
      const int gh = h0 + (v >> 3), gw = w0 + (v & 7);
      if (ct * BN + cg * 8 < a.Cout)
        *(f16x8*)(yout + chan_off(a.out_blk, ((long)gd * a.H + gh) * a.W + gw, a.Cout_off + ct * BN + cg * 8, a.Cout_stride, nvox)) =
            *(const f16x8*)(ot + v * 128 + cg * 16);
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
  __syncthreads();
  if (wave == 0) {
    double S = 0, Q = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { S += (double)ex[(w * BN + lane) * 2]; Q += (double)ex[(w * BN + lane) * 2 + 1]; }
    if (ct * BN + lane < a.Cout) stats_add(a.stats, n, a.cout_pad, replica, ct * BN + lane, S, Q);
  }
  DUA_STAMP_AT(63, false);
  DUA_STAMP_AT(1, true);
}

// grid = (8x8x8 tiles in XCD-contiguous order, cout tiles, N)
__global__ __launch_bounds__(256, 2) void conv3d_k3_wide_kernel(Conv3Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int per_slab = a.tiles_h * a.tiles_w;
  const int tile = xcd_remap(blockIdx.x, a.ntiles);
  const int td = tile / per_slab, rem = tile - td * per_slab, th = rem / a.tiles_w, tw = rem - th * a.tiles_w;
  wide_tile<4>(a, smem, td * 8, th * 8, tw * 8, blockIdx.y, blockIdx.z, blockIdx.x & (STAT_REPLICAS - 1));
}


// the data-gradient launches of training whose output is another layer's dA (Conv3Args::bw_sums): backward-sums epilogue
__global__ __launch_bounds__(256, 2) void conv3d_k3_wide_bwd_kernel(Conv3Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int per_slab = a.tiles_h * a.tiles_w;
  const int tile = xcd_remap(blockIdx.x, a.ntiles);
  const int td = tile / per_slab, rem = tile - td * per_slab, th = rem / a.tiles_w, tw = rem - th * a.tiles_w;
  wide_tile<4, true>(a, smem, td * 8, th * 8, tw * 8, blockIdx.y, blockIdx.z, blockIdx.x & (STAT_REPLICAS - 1));
}

// ---- round 5: the same tile with PERSISTENT workgroups and the accumulators in v[128:255] by name (named_acc.hpp) ----
// A workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... of its (cout tile, sample): bias, the statistics -> scale / shift
// tables of a fused input and the kernel's set-up are paid once per workgroup; the NEXT tile's first half-chunk halo and first
// weight plane are requested under the last three phases of the current tile exactly like a next half chunk (they land while
// the matrix pipes run), so a tile boundary costs an epilogue + a halo transform/store instead of an epilogue + a cold prologue
// (round 4 measured that at -3 ... -8 % per launch with the tile loop in plain C++, and lost it again to 190-230 spilled
// registers: hipcc copied the eight accumulator tuples around the loop -- profiles/r4_conv_wide_persistent_tiles_ab.txt.  With
// the tuples held by name they are not values of the program and the allocator sees a ~110-register kernel).
// InstanceNorm sums: a lane adds its tile sums into double accumulators and the workgroup issues ONE set of atomics at the end.
__global__ __launch_bounds__(256, 2) DUA_NAMED_ACC_KERNEL void conv3d_k3_wide_pt_kernel(Conv3Args a) {
  using namespace c3w;
  using T = f16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* wbuf = smem + HALO_MAX;
  float* xsc = (float*)(smem + LDS_FIXED);
  float* xsh = xsc + a.Cin;
  float* xad = xsh + a.Cin;
  float* ex = xad + (a.xf.stats ? a.Cin : 0) - (a.xf.stats ? 0 : 2 * a.Cin);          // behind the tables (or at LDS_FIXED)
  constexpr int HD = 10;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int ct = blockIdx.y, n = blockIdx.z, replica = blockIdx.x & (STAT_REPLICAS - 1);
  const bool fused = a.xf.stats != nullptr;
  const int nhc = (a.Cin + 15) >> 4;
  const int per_slab = a.tiles_h * a.tiles_w;

  // ---- halo pieces of this thread (see wide_tile): position (hy, hx) x half p, piece j = halo plane j ----
  const int p_t = tid & 1, pos = tid >> 1;
  const int hy = pos / HW, hx = pos - hy * HW;
  const int vstride = a.in_blk ? 16 : a.Cin_stride;
  const long hcstride = a.in_blk ? (long)a.D * a.H * a.W * 16 : 16;
  const T* xin = (const T*)a.x + (long)n * a.D * a.H * a.W * a.Cin_stride + (a.in_blk ? (long)(a.Cin_off >> 4) * a.D * a.H * a.W * 16 : a.Cin_off) + p_t * 8;
  const int pstep = a.H * a.W * vstride;
  const int loff = pos < HH * HW ? hy * RSF + hx * VSF + ((p_t ^ (hy & 1)) << 4) : HW * VSF + (p_t << 4);
  f16x8 hreg[HD];
  // tile geometry as the loads / stores of a halo need it: (d0, voff, ok_hw) -- `cur` for the tile being computed, `nxt` for
  // the tile whose first halo is in flight
  struct Geo { int d0, h0, w0, voff; bool ok_hw; };
  auto geo_of = [&](int t) {
    Geo g;
    const int tile = xcd_remap(t, a.ntiles);
    const int td = tile / per_slab, rem = tile - td * per_slab, th = rem / a.tiles_w, tw = rem - th * a.tiles_w;
    g.d0 = td * 8; g.h0 = th * 8; g.w0 = tw * 8;
    const int gh = g.h0 + hy - 1, gw = g.w0 + hx - 1;
    g.ok_hw = pos < HH * HW && (unsigned)gh < (unsigned)a.H && (unsigned)gw < (unsigned)a.W;
    g.voff = g.ok_hw ? (((g.d0 - 1) * a.H + gh) * a.W + gw) * vstride : 0;
    return g;
  };
  auto load_halo = [&](const Geo& g, int hc, int j0, int j1) {
    const T* src = xin + hc * hcstride;
#pragma unroll
    for (int j = 0; j < HD; ++j) {
      if (j < j0 || j >= j1) continue;
      const bool dok = (unsigned)(g.d0 + j - 1) < (unsigned)a.D;
      hreg[j] = *(const f16x8*)(src + (g.ok_hw && dok ? g.voff + j * pstep : 0));
    }
  };
  auto store_halo = [&](const Geo& g, int hc) {
    float sc[8], sh[8], ad[8], sn[8];
    if (fused) {
      const int c0 = hc * 16 + p_t * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc[e] = xsc[c0 + e]; sh[e] = xsh[c0 + e]; ad[e] = xad[c0 + e]; }
      xform_prep<T>(sc, sh, ad, sn, a.xf.slope);
    }
#pragma unroll
    for (int j = 0; j < HD; ++j) {
      f16x8 v = hreg[j];
      if (fused) v = xform_frag<T>(v, sc, sh, ad, sn, a.xf.slope);
      const f32x4 raw = __builtin_bit_cast(f32x4, v);
      const bool ok = g.ok_hw && (unsigned)(g.d0 + j - 1) < (unsigned)a.D;
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = ok ? raw[e] : 0.f;
      *(f32x4*)(halo + loff + j * PSF) = o;
    }
  };

  const char* wsrc = (const char*)a.w + (long)ct * a.nchunks * 9 * SLAB + lane * 16;
  const unsigned wlds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)wbuf;
  auto dma_plane = [&](int hc, int kd, int slot) {
    const char* src = wsrc + (long)((hc >> 1) * 3 + kd) * 3 * SLAB + (hc & 1) * 2048;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int p = wave + 4 * j;
      if (p < 18) dma_piece(src + ((p >> 1) * 4 + (p & 1)) * 1024, wlds + slot * WPLANE + p * 1024);
    }
  };

  // ---- once per workgroup: bias, statistics -> tables, first tile's halo and first weight plane ----
  float bias_q[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int bc = ct * BN + q * 32 + r;
    bias_q[q] = bc >= a.Cout ? 0.f : a.bias[bc];
  }
  int t = blockIdx.x;
  Geo cur = geo_of(t);
  if (fused) xform_preamble(a.xf, n, a.Cin, xsc, xsh, xad);
  load_halo(cur, 0, 0, HD);
  dma_plane(0, 0, 0);
  int slot = 0;                                                  // ring slot of the plane the next phase reads
  if (fused) __syncthreads();

  const int sw = (r >> 3) & 1;
  const int a_base = 2 * wave * PSF + (r >> 3) * RSF + (r & 7) * VSF;
  const int a_even = a_base + ((hh ^ sw) << 4), a_odd = a_base + ((hh ^ sw ^ 1) << 4);
  const int b_base = (hh * BN + r) * 16;

  auto phase = [&](int kd, int sl) __attribute__((always_inline)) {
    const char* hp = halo + kd * PSF;
    const char* wb = wbuf + sl * WPLANE + b_base;
    f16x8 fa[2][2], fb[2][2];
    auto ldA = [&](int tp, int sub, int b) {
      const int kh = tp / 3, kw = tp - kh * 3;
      const char* ap = hp + ((kh & 1) ? a_odd : a_even) + sub * PSF + kh * RSF + kw * VSF;
      fa[b][0] = *(const f16x8*)ap;
      fa[b][1] = *(const f16x8*)(ap + 4 * RSF);
    };
    auto ldB = [&](int tp, int b) {
      fb[b][0] = *(const f16x8*)(wb + tp * 2048);
      fb[b][1] = *(const f16x8*)(wb + tp * 2048 + 512);
    };
    ldB(0, 0);
    ldA(0, 0, 0);
#pragma unroll
    for (int h = 0; h < 18; ++h) {
      const int tp = h >> 1, sub = h & 1;
      if (h + 1 < 18) {
        const int t1 = (h + 1) >> 1, sub1 = (h + 1) & 1;
        if (sub1 == 0) ldB(t1, t1 & 1);
        ldA(t1, sub1, (h + 1) & 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      named_mfma_sel(4 * sub + 0, fa[h & 1][0], fb[tp & 1][0]);          // tuple (2 sub + mm) * 2 + q
      named_mfma_sel(4 * sub + 1, fa[h & 1][0], fb[tp & 1][1]);
      named_mfma_sel(4 * sub + 2, fa[h & 1][1], fb[tp & 1][0]);
      named_mfma_sel(4 * sub + 3, fa[h & 1][1], fb[tp & 1][1]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  const long nvox = (long)a.D * a.H * a.W;
  T* yout = (T*)a.y + (long)n * nvox * a.Cout_stride;
  double Sacc[2] = {0, 0}, Qacc[2] = {0, 0};
  constexpr int J1 = 4, J2 = 7;

  if (a.ksplit > 0) {
    // The two workgroups of a CU start together and walk identical tiles: they would stay in step, both in their epilogues at
    // once with the matrix pipes idle.  The one in the odd wave slot starts a fraction of a tile late (s_sleep 127 = 8 128 cycles);
    // a.ksplit (unused by this form otherwise) carries the number of sleeps.
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (hwid & 1)
      for (int k = 0; k < a.ksplit; ++k) __builtin_amdgcn_s_sleep(127);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the first tile's halo and weight plane
  for (;;) {
    const int tn = t + gridDim.x;
    const bool has_next = tn < a.ntiles;
    Geo nxt = cur;
    if (has_next) nxt = geo_of(tn);
    // (the halo in hreg and the weight plane have been waited for: before the loop for the first tile, at the end of the last
    // phase for every later one -- a wait here would be a wait for the epilogue's output stores, vmcnt retires in order)
    store_halo(cur, 0);
    {
      float v[16];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = bias_q[q];
#pragma unroll
        for (int m = 0; m < 4; ++m) named_write16_sel(m * 2 + q, v);
      }
    }
    named_acc_fence_init();
    __syncthreads();

    for (int hc = 0; hc < nhc; ++hc) {
      const bool more = hc + 1 < nhc;
      const bool pre = more || has_next;                         // something to prefetch under this half chunk's phases
#pragma unroll
      for (int kd = 0; kd < 3; ++kd) {
        if (kd < 2) dma_plane(hc, kd + 1, slot ^ 1);
        else if (more) dma_plane(hc + 1, 0, slot ^ 1);
        else if (has_next) dma_plane(0, 0, slot ^ 1);
        const int j0 = kd == 0 ? 0 : kd == 1 ? J1 : J2, j1 = kd == 0 ? J1 : kd == 1 ? J2 : HD;
        if (more) load_halo(cur, hc + 1, j0, j1);
        else if (has_next) load_halo(nxt, 0, j0, j1);
        phase(kd, slot);
        if (pre && kd == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(J1) : "memory");
        else if (pre && kd == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(J2 - J1) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        slot ^= 1;
      }
      if (more) {
        store_halo(cur, hc + 1);
        __syncthreads();
      }
    }

    // ---- epilogue of this tile (the K loop ended with a workgroup barrier: the halo is free); hreg holds the next tile's halo ----
    {
      char* ot = halo + wave * 8192;
      float s[2] = {0.f, 0.f}, ss[2] = {0.f, 0.f};
      named_acc_fence_read();
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            float av[16];
            named_read16_sel((2 * pr + mm) * 2 + q, av);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const float v = av[i];
              s[q] += v;
              ss[q] = fmaf(v, v, ss[q]);
              *(T*)(ot + mm * 4096 + acc_row(i, hh) * 128 + (q * 32 + r) * 2) = (T)v;
            }
            asm volatile("" : "+v"(s[q]), "+v"(ss[q]));          // keeps the sum chain here (hipcc sinks it behind the stores)
            __builtin_amdgcn_sched_barrier(0);
          }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int gd = cur.d0 + 2 * wave + pr;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int v = it * 8 + (lane >> 3), cg = lane & 7;
          const int gh = cur.h0 + (v >> 3), gw = cur.w0 + (v & 7);
          if (ct * BN + cg * 8 < a.Cout)
            *(f16x8*)(yout + chan_off(a.out_blk, ((long)gd * a.H + gh) * a.W + gw, a.Cout_off + ct * BN + cg * 8, a.Cout_stride, nvox)) =
                *(const f16x8*)(ot + v * 128 + cg * 16);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) { Sacc[q] += (double)s[q]; Qacc[q] += (double)ss[q]; }
    }
    if (!has_next) break;
    __syncthreads();                                            // every wave is done with its staging rows: the halo may be overwritten
    t = tn;
    cur = nxt;
  }

  // ---- the workgroup's InstanceNorm sums: lane halves, then waves, in a fixed order; one set of atomics ----
  double* exd = (double*)(smem);                                 // [4 waves][64 couts][2] doubles = 4 KB of the (finished) halo
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const double S = Sacc[q] + __shfl_xor(Sacc[q], 32), Q = Qacc[q] + __shfl_xor(Qacc[q], 32);
    if (hh == 0) { exd[(wave * BN + q * 32 + r) * 2] = S; exd[(wave * BN + q * 32 + r) * 2 + 1] = Q; }
  }
  __syncthreads();
  if (wave == 0) {
    double S = 0, Q = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { S += exd[(w * BN + lane) * 2]; Q += exd[(w * BN + lane) * 2 + 1]; }
    if (ct * BN + lane < a.Cout) stats_add(a.stats, n, a.cout_pad, replica, ct * BN + lane, S, Q);
  }
  (void)ex;
}

static const LdsAttr kWideLdsAttrs[] = {{(const void*)conv3d_k3_wide_kernel, 80 * 1024}, {(const void*)conv3d_k3_wide_pt_kernel, 80 * 1024},
                                        {(const void*)conv3d_k3_wide_bwd_kernel, 80 * 1024}};
static const LdsAttrs kWideLdsReg(kWideLdsAttrs);

#ifdef DUA_STAMP
extern "C" long dua_debug_stamps_wide(void* host, long bytes) { return stamps_out(host, bytes); }
#endif

bool conv3_wide_takes_bwd_sums(const Conv3Args& a) {
  return a.bw_raw && a.bw_xf.stats && a.bw_xf.gamma && a.bw_xf.beta && a.bw_xf.c_pad >= a.Cout && a.bw_stride % 8 == 0 && a.bw_off % 8 == 0 &&
         (long)a.D * a.H * a.W * a.bw_stride < 0x7fffffffL;
}

int launch_conv3_wide(Conv3Args a, int D, hipStream_t s, bool persistent, int stagger) {
  using namespace c3w;
  if (int e = ensure_prepared()) return e;
  if (D % 8 || a.H % 8 || a.W % 8 || a.Cin % 16 || (a.xf.stats && a.Cin > 256)) return DUA_ERR_ARG;
  if (a.in_blk && (a.Cin_off % 16 || a.Cin_stride % 16)) return DUA_ERR_ARG;
  if (a.out_blk && (a.Cout_off % 16 || a.Cout_stride % 16)) return DUA_ERR_ARG;
  a.tiles_h = a.H / 8; a.tiles_w = a.W / 8;
  a.ntiles = (D / 8) * a.tiles_h * a.tiles_w;
  const int lds = LDS_FIXED + (a.xf.stats ? 3 * 4 * a.Cin : 0) + (a.bw_sums ? 1024 : 0);
  if (lds > 80 * 1024) return DUA_ERR_ARG;
  if (a.bw_sums && (persistent || !conv3_wide_takes_bwd_sums(a))) return DUA_ERR_ARG;
  if (persistent) {
    // two workgroups per CU over the whole launch, shared by the (cout tile, sample) pairs; each walks its tiles with stride grid.x
    const int cus = device_cus();
    if (cus <= 0) return DUA_ERR_ARG;
    const int pairs = (a.cout_pad / BN) * a.N;
    int gx = (2 * cus + pairs - 1) / pairs;
    gx = (gx + 7) & ~7;                                          // whole XCD rounds: tile t and t + grid.x stay on one XCD
    if (gx > a.ntiles) gx = a.ntiles;
    a.ksplit = stagger;
    hipLaunchKernelGGL(conv3d_k3_wide_pt_kernel, dim3(gx, a.cout_pad / BN, a.N), dim3(256), lds, s, a);
    return (int)hipGetLastError();
  }
  if (a.bw_sums) hipLaunchKernelGGL(conv3d_k3_wide_bwd_kernel, dim3(a.ntiles, a.cout_pad / BN, a.N), dim3(256), lds, s, a);
  else hipLaunchKernelGGL(conv3d_k3_wide_kernel, dim3(a.ntiles, a.cout_pad / BN, a.N), dim3(256), lds, s, a);
  return (int)hipGetLastError();
}

}  // namespace dua
