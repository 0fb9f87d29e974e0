// UpCat's first convolution with the transposed convolution in front of it folded in (round 5).
//
// models/basic_unet/denoiser.py:172-194 (UpCat.forward):   x_0 = upsample(x)            ConvTranspose3d(k2, s2) + bias
//                                                           y   = conv(cat([x_e, x_0]))  Conv3d(k3, p1) + bias   (TwoConv.conv_0)
// Nothing non-linear sits between the two, so for an output voxel o = 2 m + phi (phi in {0,1}^3 its parity, m its parent cell)
//   y[o] = b'(o) + sum_{taps t} Wc_skip[t] x_e[o + t - 1]  +  sum_{delta in {0,1}^3} W'[phi][delta] u[m + delta - 1 + phi]
// with W'[phi][delta] = sum over the taps t whose input voxel o + t - 1 is a child of parent m + delta - 1 + phi of
// Wc_up[t] Wd[child] (1, 2, 4 or 8 products), and b'(o) = bc + sum_{t inside the volume} Wc_up[t] bd (27 border classes).
// The upsampled half of the convolution therefore contracts 8 parents x Cu channels instead of 27 taps x Cmid channels:
// at 96^3 (Cu = Cmid = 64) 58 GFLOP instead of 195.7 + 7.2 for the transposed convolution itself, whose launch, output
// write (113 MB) and re-read disappear -- the same function of the same parameters, regrouped (composed weights in fp32,
// rounded once to fp16; dua_pack_upconv_weights below).
//
// Kernel = the wide-tile form of conv3d_wide.hip (8x8x8 output voxels x 64 channels per workgroup, 8 accumulators per wave,
// 16-channel half chunks, kd planes of weights by LDS-DMA) with PHASE-MAJOR accumulators: an MFMA block applies ONE weight
// matrix to its 32 rows, so the rows of a block must share their parity.  Wave w = (pd, ph) = (w >> 1, w & 1) owns the voxels
// of the tile with d, h parity (pd, ph); its four blocks are (cz_hi, pw): 32 cells (cz_lo 2 x cy 4 x cx 4) of w-parity pw.
//   part 1 (skip half): as the wide kernel, A fragments gathered with doubled strides from a halo laid out for exactly that
//     (rows of 336 B, 16-byte halves swapped by (plane >> 1) & 1: every ds_read_b128 conflict-free, checked by emulation);
//   part 2 (upsampled half): the 6x6x6 parent cells of 64 coarse channels (30 KB, normalised + activated on the way in) are
//     staged where the weight ring was; per (delta_d, delta_h, half chunk, pw) a wave multiplies 2 x 2 A fragments (x offsets,
//     cz_hi) with the 2 x 2 B fragments (delta_w, cout half) of ITS OWN parities -- those weights are wave-private, so they
//     come straight from L2 into registers (1 KB per load instruction, three iterations ahead), not through LDS.
#include "common.hpp"
#include "../../include/dua_hip.h"
#include "stamp.hpp"
#include "named_acc.hpp"

namespace dua {

namespace upc {
constexpr int VS = 32, RS = 336, PS = 10 * RS;                 // skip halo: 32-byte voxels, 10 voxels + 16 B per row, 10 rows per plane
constexpr int HALO = 10 * PS;                                  // 33600
constexpr int WPLANE = 18 * 1024, RING = 2 * WPLANE;           // [9 taps][2 k-groups][64 couts][16 B] x 2
constexpr int CR = 6 * VS, CP = 6 * CR + 128, CHC = 6 * CP;    // coarse halo per half chunk: rows 192 B, planes 1280 B, 7680 B
constexpr int CGRP = 4 * CHC;                                  // 64 coarse channels: 30720 B (<= RING and <= HALO)
constexpr int SLAB = 3 * 4 * 64 * 16;                          // one (kd, kh) slab of the packed skip weights (32-channel chunk)
constexpr int BN = 64;
constexpr int NPIECE = 7;                                      // coarse pieces per thread and group: 216 cells x 8 parts / 256
static_assert(CGRP <= RING && CGRP <= HALO && 4 * 8192 <= HALO, "regions");
}  // namespace upc

struct UpConvArgs {
  const void* xs; const void* u; const void* w; const void* wu; const float* btab; void* y;
  stat_t* stats;
  InXform xf;                          // producer of u
  int N, D, H, W;                      // output (fine) extents
  int Cs, Cs_stride, Cs_off, in_blk;
  int Cu, Cu_stride, Cu_off;
  int Cout, Cout_stride, Cout_off, out_blk, cout_pad;
  int nchunks, ntiles, tiles_h, tiles_w;
};

__device__ __forceinline__ void upc_dma_piece(const char* src_lane, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src_lane), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}

// GG = groups of 64 coarse channels (1 or 2): compile-time, so that the registers of a second group's prefetch and of the
// statistics words of channels that do not exist are not allocated
#ifndef UPC_LA
#define UPC_LA 1          // half-steps of A fragments in flight ahead of the MFMAs of the skip half's K loop
#endif
template <int GG>
__global__ __launch_bounds__(256, 2) DUA_NAMED_ACC_KERNEL void upconv_k3_kernel(UpConvArgs a) {
  using namespace upc;
  using T = f16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* ring = smem + HALO;
  float* xsc = (float*)(smem + HALO + RING);
  float* xsh = xsc + a.Cu;
  float* xad = xsh + a.Cu;
  float* ex = xad + a.Cu;                                      // [4 waves][64 couts][2]

  const int per_slab = a.tiles_h * a.tiles_w;
  const int tile = xcd_remap(blockIdx.x, a.ntiles);
  const int td = tile / per_slab, rem = tile - td * per_slab, th = rem / a.tiles_w, tw = rem - th * a.tiles_w;
  const int d0 = td * 8, h0 = th * 8, w0 = tw * 8, ct = blockIdx.y, n = blockIdx.z, replica = blockIdx.x & (STAT_REPLICAS - 1);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int pd = wave >> 1, ph = wave & 1;
  const int nhc = a.Cs >> 4, U = nhc * 3;
  constexpr int G = GG;
  const bool fused = a.xf.stats != nullptr;                   // u is a raw convolution output (else: already materialised)

  // ---- skip halo pieces: thread = (position (hy, hx) in a halo plane, 16-byte half p), piece j = halo plane j ----
  const int p_t = tid & 1, pos = tid >> 1;
  const bool has_pos = pos < 100;
  const int hy = pos / 10, hx = pos - hy * 10;
  const int gh = h0 + hy - 1, gw = w0 + hx - 1;
  const bool ok_hw = has_pos && (unsigned)gh < (unsigned)a.H && (unsigned)gw < (unsigned)a.W;
  const int vstride = a.in_blk ? 16 : a.Cs_stride;
  const long hcstride = a.in_blk ? (long)a.D * a.H * a.W * 16 : 16;
  // (wave-uniform base pointers stay in scalar registers; what depends on the lane is a 32-bit element offset)
  const T* xin = (const T*)a.xs + (long)n * a.D * a.H * a.W * a.Cs_stride +
                 (a.in_blk ? (long)(a.Cs_off >> 4) * a.D * a.H * a.W * 16 : a.Cs_off);
  const int voff = (ok_hw ? (((d0 - 1) * a.H + gh) * a.W + gw) * vstride : 0) + p_t * 8;
  const int pstep = a.H * a.W * vstride;
  const int lbase = has_pos ? hy * RS + hx * VS : 0;
  const int lsw0 = lbase + (p_t << 4), lswd = 16 - (p_t << 5);                 // planes with (j >> 1) & 1 = 1: the other half
  f16x8 hreg[10];
  auto load_halo = [&](int hc, int j0, int j1) {
    const T* src = xin + hc * hcstride;
    int vo = voff;
    asm volatile("" : "+v"(vo));      // 64-bit addresses are formed per call (kept across the loop they cost 16 registers of scratch
                                      // and a reload -- behind the weight pieces, vmcnt retires in order -- at the head of every phase)
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      if (j < j0 || j >= j1) continue;
      const bool dok = (unsigned)(d0 + j - 1) < (unsigned)a.D;       // wave-uniform
      hreg[j] = *(const f16x8*)(src + (ok_hw && dok ? vo + j * pstep : p_t * 8));
    }
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      const f32x4 raw = __builtin_bit_cast(f32x4, hreg[j]);
      const bool ok = ok_hw && (unsigned)(d0 + j - 1) < (unsigned)a.D;
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = ok ? raw[e] : 0.f;
      if (has_pos) *(f32x4*)(halo + lsw0 + (((j >> 1) & 1) ? lswd : 0) + j * PS) = o;
    }
  };

  // ---- skip weights: kd plane `kd` of half chunk `hc` -> ring slot (18 pieces of 1 KB; wave w takes pieces w, w + 4, ...) ----
  const char* wsrc = (const char*)a.w + (long)ct * a.nchunks * 9 * SLAB;
  const int wlane = lane * 16;
  const unsigned wlds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;
  auto dma_plane = [&](int u, int slot) {
    const int hc = u / 3, kd = u - hc * 3;
    const char* src = wsrc + (long)((hc >> 1) * 3 + kd) * 3 * SLAB + (hc & 1) * 2048;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int p = wave + 4 * j;                              // piece = (tap p >> 1, k-group p & 1)
      if (p < 18) upc_dma_piece(src + ((p >> 1) * 4 + (p & 1)) * 1024 + wlane, wlds + slot * WPLANE + p * 1024);
    }
  };

  // ---- coarse halo pieces of one 64-channel group: cell = (tid >> 3) + 32 j (6 x 6 x 6 parent cells), part = tid & 7 ----
  // (lane-dependent values of the later sections are derived from a thread id made opaque at the head of each section:
  // otherwise hipcc computes them all in the prologue, keeps them through the K loops of a 128-register kernel in scratch, and the
  // reloads in the epilogue wait -- vmcnt retires in order -- for the output stores in front of them: 23 000 cycles of epilogue)
  int tid2 = tid;
  const int Dc = a.D >> 1, Hc = a.H >> 1, Wc = a.W >> 1;
  f16x8 creg[NPIECE];
  auto load_coarse = [&](int g, int j0, int j1) {
    int tl = tid2;
    asm volatile("" : "+v"(tl));      // addresses are formed HERE: hoisted out of the half-chunk loop they sat in scratch through part 1
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
      if (j < j0 || j >= j1) continue;
      const int cell = (tl >> 3) + 32 * j;
      const int z = cell / 36, rm = cell - z * 36, y = rm / 6, x = rm - y * 6;
      const int gz = (d0 >> 1) - 1 + z, gy = (h0 >> 1) - 1 + y, gx = (w0 >> 1) - 1 + x;
      const bool ok = cell < 216 && (unsigned)gz < (unsigned)Dc && (unsigned)gy < (unsigned)Hc && (unsigned)gx < (unsigned)Wc;
      const T* uin = (const T*)a.u + (long)n * Dc * Hc * Wc * a.Cu_stride + a.Cu_off + (tl & 7) * 8;
      creg[j] = *(const f16x8*)(uin + (ok ? (long)((gz * Hc + gy) * Wc + gx) * a.Cu_stride + g * 64 : 0));
    }
  };
  auto store_coarse = [&](int g, char* dst) {
    float sc[8], sh[8], ad[8], sn[8];
    const int part = tid2 & 7;
    const int c0 = g * 64 + part * 8;
    if (fused) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc[e] = xsc[c0 + e]; sh[e] = xsh[c0 + e]; ad[e] = xad[c0 + e]; }
      xform_prep<T>(sc, sh, ad, sn, a.xf.slope);
    }
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
      const int cell = (tid2 >> 3) + 32 * j;
      const int z = cell / 36, rm = cell - z * 36, y = rm / 6, x = rm - y * 6;
      const int gz = (d0 >> 1) - 1 + z, gy = (h0 >> 1) - 1 + y, gx = (w0 >> 1) - 1 + x;
      const bool ok = (unsigned)gz < (unsigned)Dc && (unsigned)gy < (unsigned)Hc && (unsigned)gx < (unsigned)Wc;
      f16x8 v = creg[j];
      if (fused) v = xform_frag<T>(v, sc, sh, ad, sn, a.xf.slope);
      const f32x4 raw = __builtin_bit_cast(f32x4, v);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = ok ? raw[e] : 0.f;
      if (cell < 216) *(f32x4*)(dst + (part >> 1) * CHC + z * CP + y * CR + x * VS + (((part & 1) ^ (y & 1)) << 4)) = o;
    }
  };

  // ---- prologue: everything that must come from memory is requested before anything waits ----
  DUA_STAMP_AT(0, true);
  DUA_STAMP_AT(2, false);
  const bool border = d0 == 0 || d0 + 8 == a.D || h0 == 0 || h0 + 8 == a.H || w0 == 0 || w0 + 8 == a.W;
  float bias_q[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int bc = ct * BN + q * 32 + r;
    bias_q[q] = bc >= a.Cout ? 0.f : a.btab[13 * a.cout_pad + bc];          // interior class (1, 1, 1)
  }
  constexpr int UN = GG;                                       // 16 channels per (wave, group): 64 channels per group
  stat_t sv[UN][2 * STAT_WORDS];
  float gam[UN], bet[UN], addv[UN];
  if (fused) {
    const int pr = lane >> 4;
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int c = (wave + 4 * u) * 16 + (lane & 15), cc = c < a.Cu ? c : a.Cu - 1;
      gam[u] = a.xf.gamma[cc]; bet[u] = a.xf.beta[cc];
      addv[u] = a.xf.add ? a.xf.add[(long)n * a.xf.add_stride + cc] : 0.f;
      const stat_t* sp = a.xf.stats + ((long)n * STAT_REPLICAS + 2 * pr) * STAT_WORDS * a.xf.c_pad + cc;
      if ((wave + 4 * u) * 16 < a.Cu) {
#pragma unroll
        for (int k = 0; k < 2 * STAT_WORDS; ++k) sv[u][k] = sp[(long)k * a.xf.c_pad];
      } else {
#pragma unroll
        for (int k = 0; k < 2 * STAT_WORDS; ++k) sv[u][k] = 0;
      }
    }
  }
  load_halo(0, 0, 10);
  dma_plane(0, 0);
  float btv[7];
  if (border) {                                                // 27 classes x 64 channels of this cout tile -> ring slot 1
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int e = tid + 256 * j, cls = e >> 6, c = ct * BN + (e & 63);
      btv[j] = (e < 27 * 64 && c < a.Cout) ? a.btab[cls * a.cout_pad + c] : 0.f;
    }
  }
  if (fused) {
    const int pr = lane >> 4;
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
      if (pr == u) {
        Sm = (double)w[0] + (double)w[1] * (1.0 / STAT_FRAC);
        Qm = (double)w[2] + (double)w[3] * (1.0 / STAT_FRAC);
        gm = gam[u]; bm = bet[u]; am = addv[u];
      }
    }
    const int c = (wave + 4 * pr) * 16 + (lane & 15);
    const double mean = Sm * a.xf.inv_count;
    double var = Qm * a.xf.inv_count - mean * mean;
    var = var > 0 ? var : 0;
    const float g = gm * (float)(1.0 / sqrt(var + (double)a.xf.eps));
    if (c < a.Cu) {
      xsc[c] = g;
      xsh[c] = bm - (float)mean * g;
      xad[c] = am;
    }
  }
  float* btl = (float*)(ring + WPLANE);
  if (border) {
#pragma unroll
    for (int j = 0; j < 7; ++j)
      if (tid + 256 * j < 27 * 64) btl[tid + 256 * j] = btv[j];
  }
  __syncthreads();
  store_halo();

  // ---- accumulators: tuple (m * 2 + q) = a[16 (2 m + q) ..] for block m = cz_hi * 2 + pw and cout half q, held in
  // v[128:255] by name (named_acc.hpp); register i of lane half hh = cell (cz_lo = i >> 3, cy = hh + 2 ((i >> 2) & 1),
  // cx = i & 3); they start at the bias of their voxel's border class ----
  if (!border) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = bias_q[q];
#pragma unroll
      for (int m = 0; m < 4; ++m) named_write16_sel(m * 2 + q, v);
    }
  } else {
    const bool lo_d = d0 == 0, hi_d = d0 + 8 == a.D, lo_h = h0 == 0, hi_h = h0 + 8 == a.H, lo_w = w0 == 0, hi_w = w0 + 8 == a.W;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int ld = 2 * (2 * (m >> 1) + (i >> 3)) + pd, lh = 2 * (hh + 2 * ((i >> 2) & 1)) + ph, lw = 2 * (i & 3) + (m & 1);
          const int cd = (lo_d && ld == 0) ? 0 : (hi_d && ld == 7) ? 2 : 1;
          const int ch = (lo_h && lh == 0) ? 0 : (hi_h && lh == 7) ? 2 : 1;
          const int cw = (lo_w && lw == 0) ? 0 : (hi_w && lw == 7) ? 2 : 1;
          v[i] = btl[(cd * 9 + ch * 3 + cw) * 64 + q * 32 + r];
        }
        named_write16_sel(m * 2 + q, v);
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  named_acc_fence_init();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's weight pieces (the compiler does not count them)
  __syncthreads();
  DUA_STAMP_AT(3, false);

  asm volatile("" : "+v"(tid2));
  const int lane1 = tid2 & 63, r1 = lane1 & 31, hh1 = lane1 >> 5;
  const int cz_lo = r1 >> 4, cy = (r1 >> 2) & 3, cx = r1 & 3;
  const int La = 2 * cz_lo * PS + 2 * cy * RS + 2 * cx * VS + pd * PS + ph * RS;
  const int A0 = La + ((hh1 ^ cz_lo) << 4), A1 = La + ((hh1 ^ cz_lo ^ 1) << 4);
  const int b_base = (hh1 * BN + r1) * 16;

  // One phase = the 9 taps of kd plane `kd` of the current half chunk against ring slot `slot`: half-step h = (tap t = h / 2,
  // sub = h % 2 = cz_hi) is 4 MFMAs on the A pair (pw = 0, 1) of (t, sub) and the B pair of t.
  auto phase = [&](int kd, int slot) __attribute__((always_inline)) {
    const char* hp = halo + kd * PS + (((pd + kd) >> 1) ? A1 : A0);
    const char* wb = ring + slot * WPLANE + b_base;
    f16x8 fa[UPC_LA + 1][2], fb[2][2];
    auto ldA = [&](int t, int sub, int b) {
      const int kh = t / 3, kw = t - kh * 3;
      const char* ap = hp + sub * 4 * PS + kh * RS + kw * VS;
      fa[b][0] = *(const f16x8*)ap;
      fa[b][1] = *(const f16x8*)(ap + VS);
    };
    auto ldB = [&](int t, int b) {
      fb[b][0] = *(const f16x8*)(wb + t * 2048);
      fb[b][1] = *(const f16x8*)(wb + t * 2048 + 512);
    };
    // the fragments of half-step h + UPC_LA are requested before the MFMAs of half-step h issue (B pairs: one tap ahead)
    ldB(0, 0);
#pragma unroll
    for (int h = 0; h < UPC_LA; ++h) ldA(h >> 1, h & 1, h);
#pragma unroll
    for (int h = 0; h < 18; ++h) {
      const int t = h >> 1, sub = h & 1;
      if (h + UPC_LA < 18) {
        const int t1 = (h + UPC_LA) >> 1, sub1 = (h + UPC_LA) & 1;
        if (UPC_LA == 1 && sub1 == 0) ldB(t1, t1 & 1);
        ldA(t1, sub1, (h + UPC_LA) % (UPC_LA + 1));
      }
      if (UPC_LA > 1 && sub == 0 && t + 1 < 9) ldB(t + 1, (t + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      named_mfma_sel(4 * sub + 0, fa[h % (UPC_LA + 1)][0], fb[t & 1][0]);
      named_mfma_sel(4 * sub + 1, fa[h % (UPC_LA + 1)][0], fb[t & 1][1]);
      named_mfma_sel(4 * sub + 2, fa[h % (UPC_LA + 1)][1], fb[t & 1][0]);
      named_mfma_sel(4 * sub + 3, fa[h % (UPC_LA + 1)][1], fb[t & 1][1]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- part 1: the skip half.  The next half chunk's halo -- or, under the last one, the first group of the coarse halo --
  // is requested in three groups at the head of the three phases, behind that phase's weight pieces ----
  for (int hc = 0; hc < nhc; ++hc) {
    const bool more = hc + 1 < nhc;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      const int u = hc * 3 + kd, slot = u & 1;
      if (u + 1 < U) dma_plane(u + 1, slot ^ 1);
      if (more) load_halo(hc + 1, kd == 0 ? 0 : kd == 1 ? 4 : 7, kd == 0 ? 4 : kd == 1 ? 7 : 10);
      else load_coarse(0, kd == 0 ? 0 : kd == 1 ? 3 : 5, kd == 0 ? 3 : kd == 1 ? 5 : NPIECE);
      phase(kd, slot);
      if (more) {
        if (kd == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (kd == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        if (kd == 0) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (kd == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      }
      __syncthreads();
      if (u < 24) DUA_STAMP_AT(4 + u, false);
    }
    if (more) {
      store_halo();
      __syncthreads();
    }
  }

  // ---- part 2: the upsampled half.  B fragments of iteration it = (delta_d, delta_h, half chunk, pw): 4 KB = (delta_w, q) ----
  asm volatile("" : "+v"(tid2));
  const int lane2 = tid2 & 63, r2 = lane2 & 31, hh2 = lane2 >> 5;
  const int cz2 = r2 >> 4, cy2 = (r2 >> 2) & 3, cx2 = r2 & 3;
  const char* wub = (const char*)a.wu + (long)(ct * 4 + wave) * G * (128 * 1024) + lane2 * 16;
  const int Lc = cz2 * CP + cy2 * CR + cx2 * VS + pd * CP + ph * CR;
  const int C0 = Lc + ((hh2 ^ ((cy2 + ph) & 1)) << 4), C1 = Lc + ((hh2 ^ ((cy2 + ph) & 1) ^ 1) << 4);
#ifndef UPC_RB1
#define UPC_RB1 3
#endif
  constexpr int RB = GG == 1 ? UPC_RB1 : 2;                    // iterations of B fragments in flight
  f16x8 fb[RB][4], fa[2][4];
  auto ldBu = [&](const char* wg, int it, int b) {
#pragma unroll
    for (int k = 0; k < 4; ++k) fb[b][k] = *(const f16x8*)(wg + (it * 4 + k) * 1024);
  };
  store_coarse(0, ring);
  if (G > 1) load_coarse(1, 0, NPIECE);
#pragma unroll
  for (int b = 0; b < RB; ++b) ldBu(wub, b, b);
  __syncthreads();
  DUA_STAMP_AT(30, false);
  for (int g = 0; g < G; ++g) {
    const char* cb = (g & 1) ? halo : ring;
    const char* wg = wub + (long)g * (128 * 1024);
    auto ldAu = [&](int it, int b) {
      const int dd = it >> 4, dh = (it >> 3) & 1, hcl = (it >> 1) & 3, pw = it & 1;
      const char* ap = cb + (dh ? C1 : C0) + hcl * CHC + dd * CP + dh * CR + pw * VS;
#pragma unroll
      for (int dw = 0; dw < 2; ++dw)
#pragma unroll
        for (int czh = 0; czh < 2; ++czh) fa[b][dw * 2 + czh] = *(const f16x8*)(ap + czh * 2 * CP + dw * VS);
    };
    if (g > 0) {
#pragma unroll
      for (int b = 0; b < RB; ++b) ldBu(wg, b, b);
    }
    ldAu(0, 0);
#pragma unroll
    for (int it = 0; it < 32; ++it) {
      const int pw = it & 1;
      if (it + 1 < 32) ldAu(it + 1, (it + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dw = 0; dw < 2; ++dw)
#pragma unroll
        for (int czh = 0; czh < 2; ++czh) {
          named_mfma_sel((czh * 2 + pw) * 2 + 0, fa[it & 1][dw * 2 + czh], fb[it % RB][dw * 2 + 0]);
          named_mfma_sel((czh * 2 + pw) * 2 + 1, fa[it & 1][dw * 2 + czh], fb[it % RB][dw * 2 + 1]);
        }
      __builtin_amdgcn_sched_barrier(0);
      if (it + RB < 32) ldBu(wg, it + RB, it % RB);                 // into the buffer the MFMAs above have just read
    }
    DUA_STAMP_AT(31 + g, false);
    if (g + 1 < G) {
      store_coarse(g + 1, (g & 1) ? ring : halo);
      if (g + 2 < G) load_coarse(g + 2, 0, NPIECE);
      __syncthreads();
    }
  }
  if (!(G & 1)) __syncthreads();                                 // the last group was read from the region the epilogue stages in

  // ---- epilogue: statistics from the fp32 accumulators; each wave stages the two blocks (pw = 0, 1) of one cz_hi at a
  // time as 64 voxel rows [cz_lo][cy][w = 2 cx + pw] x 128 B in rows of its own, whole voxel lines leave in 16-byte stores ----
  DUA_STAMP_AT(62, false);
  asm volatile("" : "+v"(tid2));
  const int lane3 = tid2 & 63, r3 = lane3 & 31, hh3 = lane3 >> 5;
  char* ot = halo + wave * 8192;
  const long nvox = (long)a.D * a.H * a.W;
  T* yout = (T*)a.y + (long)n * nvox * a.Cout_stride;
  float s[2] = {0.f, 0.f}, ss[2] = {0.f, 0.f};
  named_acc_fence_read();
#pragma unroll
  for (int czh = 0; czh < 2; ++czh) {
#pragma unroll
    for (int pw = 0; pw < 2; ++pw)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float av[16];
        named_read16_sel((czh * 2 + pw) * 2 + q, av);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float v = av[i];
          s[q] += v;
          ss[q] = fmaf(v, v, ss[q]);
          const int row = (i >> 3) * 32 + (hh3 + 2 * ((i >> 2) & 1)) * 8 + (i & 3) * 2 + pw;
          *(T*)(ot + row * 128 + (q * 32 + r3) * 2) = (T)v;
        }
        // one tuple at a time: the sums are made opaque here -- hipcc otherwise sinks the whole `s += v` chain of a tuple to the
        // end of the kernel, keeps its sixteen values in scratch meanwhile, and reloads them behind the output stores
        asm volatile("" : "+v"(s[q]), "+v"(ss[q]));
        __builtin_amdgcn_sched_barrier(0);
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int v = it * 8 + (lane3 >> 3), cg = lane3 & 7;         // v = staged row: cz_lo = v >> 5, cy = (v >> 3) & 3, w = v & 7
      const int gd = d0 + 2 * (2 * czh + (v >> 5)) + pd, ghh = h0 + 2 * ((v >> 3) & 3) + ph, gww = w0 + (v & 7);
      if (ct * BN + cg * 8 < a.Cout)
        *(f16x8*)(yout + chan_off(a.out_blk, ((long)gd * a.H + ghh) * a.W + gww, a.Cout_off + ct * BN + cg * 8, a.Cout_stride, nvox)) =
            *(const f16x8*)(ot + v * 128 + cg * 16);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  DUA_STAMP_AT(60, false);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    s[q] += __shfl_xor(s[q], 32);
    ss[q] += __shfl_xor(ss[q], 32);
    if (hh3 == 0) { ex[(wave * BN + q * 32 + r3) * 2] = s[q]; ex[(wave * BN + q * 32 + r3) * 2 + 1] = ss[q]; }
  }
  __syncthreads();
  DUA_STAMP_AT(61, false);
  if (wave == 0) {
    double S = 0, Q = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { S += (double)ex[(w * BN + lane3) * 2]; Q += (double)ex[(w * BN + lane3) * 2 + 1]; }
    if (ct * BN + lane3 < a.Cout) stats_add(a.stats, n, a.cout_pad, replica, ct * BN + lane3, S, Q);
  }
  DUA_STAMP_AT(63, false);
  DUA_STAMP_AT(1, true);
}

#ifdef DUA_STAMP
extern "C" long dua_debug_stamps_upconv(void* host, long bytes) { return stamps_out(host, bytes); }
#endif

static const LdsAttr kUpconvLdsAttrs[] = {{(const void*)upconv_k3_kernel<1>, 80 * 1024}, {(const void*)upconv_k3_kernel<2>, 80 * 1024}};
static const LdsAttrs kUpconvLdsReg(kUpconvLdsAttrs);

// ---- weights.  wc: Conv3d weight fp32 [Cout][Cskip + Cmid][27]; wd: ConvTranspose3d weight fp32 [Cu][Cmid][8].
// Composed element (co, ci, phi, delta): sum over the taps of (phi, delta) (per dimension: phi 0, delta 0 -> {k = 0, child 1};
// phi 0, delta 1 -> {k = 1, child 0; k = 2, child 1}; phi 1, delta 0 -> {k = 0, child 0; k = 1, child 1}; phi 1, delta 1 ->
// {k = 2, child 0}) and over cm, in a fixed order, fp32; stored as fp16 in the order the kernel streams it:
// [cout tile][wave = (pd, ph)][group g][delta_d][delta_h][half chunk][pw][delta_w][q][k-group hh][r][8 channels]. ----
__device__ __forceinline__ int upc_taps(int phi, int delta, int* k, int* child) {
  if (phi == 0 && delta == 0) { k[0] = 0; child[0] = 1; return 1; }
  if (phi == 0) { k[0] = 1; child[0] = 0; k[1] = 2; child[1] = 1; return 2; }
  if (delta == 0) { k[0] = 0; child[0] = 0; k[1] = 1; child[1] = 1; return 2; }
  k[0] = 2; child[0] = 0; return 1;
}

// One workgroup = one output channel co x 32 coarse channels: its Conv3d rows wc[co][up_off + cm][27] sit in LDS; thread = (ci, phi)
// keeps the 27 (tap, child) sums of its phase -- per dimension the three pairs (k, child) of phi, every k once -- over cm and adds
// them into its 8 deltas in the order (x, y, z) of upc_taps.  (Round 5's first form, one thread per packed element with two
// strided 4-byte loads per multiply, took 104 us for the 96^3 level -- every training step packs.)
__global__ __launch_bounds__(256) void upconv_pack_kernel(int Cout, int Cskip, int Cmid, int Cu, int up_off, const float* __restrict__ wc,
                                                          const float* __restrict__ wd, f16* __restrict__ out, long total) {
  extern __shared__ float arow[];                                // [Cmid][27]
  const int nci = Cu >> 5;
  const int co = blockIdx.x / nci, ci = (blockIdx.x - co * nci) * 32 + (threadIdx.x >> 3), phi = threadIdx.x & 7;
  const int Cin = Cskip + Cmid;
  const int cout_pad = (Cout + 63) / 64 * 64;
  if (co < Cout)
    for (int i = threadIdx.x; i < Cmid * 27; i += 256) arow[i] = wc[((long)co * Cin + up_off) * 27 + i];
  __syncthreads();
  const int pd = phi >> 2, ph = (phi >> 1) & 1, pw = phi & 1;
  // per dimension: the pairs (k, child) of phase p in the order delta 0, delta 1 (upc_taps): p = 0: (0,1) | (1,0) (2,1);  p = 1: (0,0) (1,1) | (2,0)
  float acc[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) acc[i] = 0.f;
  if (co < Cout && ci < Cu) {
    const float* wdp = wd + (long)ci * Cmid * 8;
#pragma unroll 4                                                // four rows of the transposed-convolution weights in flight (the loop was one dependent load per k)
    for (int cm = 0; cm < Cmid; ++cm) {
      const f32x4 w0 = *(const f32x4*)(wdp + cm * 8), w1 = *(const f32x4*)(wdp + cm * 8 + 4);
      float wv[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
      // child of tap k under phase p, per dimension: ((k + 1) & 1) ^ p -- the children are permuted by phi once (three conditional
      // swaps), then every tap's child index is a constant
#pragma unroll
      for (int c = 0; c < 4; ++c) { const float x = wv[c], y = wv[c + 4]; wv[c] = pd ? y : x; wv[c + 4] = pd ? x : y; }
#pragma unroll
      for (int c = 0; c < 8; c += 4)
#pragma unroll
        for (int j = 0; j < 2; ++j) { const float x = wv[c + j], y = wv[c + j + 2]; wv[c + j] = ph ? y : x; wv[c + j + 2] = ph ? x : y; }
#pragma unroll
      for (int c = 0; c < 8; c += 2) { const float x = wv[c], y = wv[c + 1]; wv[c] = pw ? y : x; wv[c + 1] = pw ? x : y; }
      const float* ar = arow + cm * 27;
#pragma unroll
      for (int kd = 0; kd < 3; ++kd)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int child0 = (((kd + 1) & 1) * 2 + ((kh + 1) & 1)) * 2 + ((kw + 1) & 1);
            acc[(kd * 3 + kh) * 3 + kw] = fmaf(ar[(kd * 3 + kh) * 3 + kw], wv[child0], acc[(kd * 3 + kh) * 3 + kw]);
          }
    }
  }
  // delta of tap k under phase p: p = 0: k 0 -> delta 0, k 1, 2 -> delta 1;  p = 1: k 0, 1 -> delta 0, k 2 -> delta 1
  const int ct = co >> 6, q = (co >> 5) & 1, r = co & 31;
  const int G = Cu >> 6, g = ci >> 6, hcl = (ci >> 4) & 3, hh = (ci >> 3) & 1, e = ci & 7;
  const int wave = pd * 2 + ph;
  if (co >= cout_pad || ci >= Cu) return;
#pragma unroll
  for (int dd = 0; dd < 2; ++dd)
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
      for (int dw = 0; dw < 2; ++dw) {
        float v = 0.f;
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              const bool in_d = (pd == 0 ? (kd == 0 ? 0 : 1) : (kd == 2 ? 1 : 0)) == dd;
              const bool in_h = (ph == 0 ? (kh == 0 ? 0 : 1) : (kh == 2 ? 1 : 0)) == dh;
              const bool in_w = (pw == 0 ? (kw == 0 ? 0 : 1) : (kw == 2 ? 1 : 0)) == dw;
              if (in_d && in_h && in_w) v += acc[(kd * 3 + kh) * 3 + kw];
            }
        const long idx = (((((((((((long)ct * 4 + wave) * G + g) * 2 + dd) * 2 + dh) * 4 + hcl) * 2 + pw) * 2 + dw) * 2 + q) * 2 + hh) * 32 + r) * 8 + e;
        if (idx < total) out[idx] = (f16)v;
      }
}

// bias table [27 classes = (cd, ch, cw), 0 = low border, 1 = interior, 2 = high border][cout_pad]: one workgroup per output
// channel; T[tap] = sum_cm wc[co][up_off + cm][tap] * bd[cm] from coalesced reads (9 groups of 27 threads over cm), then one
// thread per class adds the taps that stay inside the volume
__global__ __launch_bounds__(256) void upconv_bias_kernel(int Cout, int Cskip, int Cmid, int up_off, int cout_pad, const float* __restrict__ wc,
                                                          const float* __restrict__ bc, const float* __restrict__ bd, float* __restrict__ out) {
  __shared__ float part[9][27];
  __shared__ float T[27];
  const int co = blockIdx.x, t = threadIdx.x;
  const int Cin = Cskip + Cmid;
  if (t < 243) {
    const int grp = t / 27, tap = t - grp * 27;
    float sacc = 0.f;
    if (co < Cout && bd)
      for (int cm = grp; cm < Cmid; cm += 9) sacc = fmaf(wc[((long)co * Cin + up_off + cm) * 27 + tap], bd[cm], sacc);
    part[grp][tap] = sacc;
  }
  __syncthreads();
  if (t < 27) {
    float v = 0.f;
#pragma unroll
    for (int gq = 0; gq < 9; ++gq) v += part[gq][t];
    T[t] = v;
  }
  __syncthreads();
  if (t < 27) {
    const int cd = t / 9, ch = (t / 3) % 3, cw = t % 3;
    float v = (co < Cout && bc) ? bc[co] : 0.f;
    if (co < Cout)
      for (int kd = 0; kd < 3; ++kd) {
        if ((cd == 0 && kd == 0) || (cd == 2 && kd == 2)) continue;
        for (int kh = 0; kh < 3; ++kh) {
          if ((ch == 0 && kh == 0) || (ch == 2 && kh == 2)) continue;
          for (int kw = 0; kw < 3; ++kw) {
            if ((cw == 0 && kw == 0) || (cw == 2 && kw == 2)) continue;
            v += T[(kd * 3 + kh) * 3 + kw];
          }
        }
      }
    out[t * cout_pad + co] = v;
  }
}

static bool upconv_desc_ok(const dua_upconv_desc* d) {
  if (!d || d->dtype != DUA_F16 || d->N <= 0) return false;
  if (d->D <= 0 || d->H <= 0 || d->W <= 0 || d->D % 8 || d->H % 8 || d->W % 8) return false;
  if (d->Cskip <= 0 || d->Cskip % 16 || d->Cskip_off % 8 || d->Cskip_off + d->Cskip > d->Cskip_stride) return false;
  if (d->Cu <= 0 || d->Cu % 64 || d->Cu > 128 || d->Cu_off % 8 || d->Cu_off + d->Cu > d->Cu_stride || d->Cu_stride % 8) return false;
  if (d->Cout <= 0 || d->Cout % 8 || d->Cout_off % 8 || d->Cout_off + d->Cout > d->Cout_stride) return false;
  if (d->layout & ~(DUA_IN_BLOCKED | DUA_OUT_BLOCKED)) return false;
  if ((d->layout & DUA_IN_BLOCKED) && (d->Cskip_off % 16 || d->Cskip_stride % 16)) return false;
  if ((d->layout & DUA_OUT_BLOCKED) && (d->Cout_off % 16 || d->Cout_stride % 16)) return false;
  if (!(d->layout & DUA_IN_BLOCKED) && d->Cskip_stride % 8) return false;
  // index arithmetic of the kernel is 32-bit per sample
  if ((long)d->D * d->H * d->W * (d->Cskip_stride > d->Cout_stride ? d->Cskip_stride : d->Cout_stride) >= (1L << 31)) return false;
  return true;
}

}  // namespace dua

extern "C" {

int dua_upconv_k3_supported(const dua_upconv_desc* d) { return dua::upconv_desc_ok(d) ? 1 : 0; }

long dua_pack_upconv_weights(int dtype, int Cout, int Cskip, int Cmid, int Cu, int up_first, const float* wc, const float* bc,
                             const float* wd, const float* bd, void* wu_packed, float* bias_table, void* stream) {
  if (dtype != DUA_F16 || Cout <= 0 || Cskip < 0 || Cmid <= 0 || Cu <= 0 || Cu % 64) return DUA_ERR_ARG;
  const int up_off = up_first ? 0 : Cskip;                     // first input channel of the upsampled half in wc
  const int nct = (Cout + 63) / 64, G = Cu >> 6;
  const long total = (long)nct * 4 * G * 128 * 512;             // fp16 elements
  if (!wu_packed) return total * 2;
  if (!wc || !wd || !bias_table) return DUA_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int cout_pad = nct * 64;
  if ((size_t)Cmid * 27 * sizeof(float) > 64 * 1024) return DUA_ERR_ARG;
  hipLaunchKernelGGL(dua::upconv_pack_kernel, dim3((unsigned)(cout_pad * (Cu / 32))), dim3(256), (size_t)Cmid * 27 * sizeof(float), s, Cout, Cskip,
                     Cmid, Cu, up_off, wc, wd, (dua::f16*)wu_packed, total);
  hipLaunchKernelGGL(dua::upconv_bias_kernel, dim3(cout_pad), dim3(256), 0, s, Cout, Cskip, Cmid, up_off, cout_pad, wc, bc, bd, bias_table);
  const int e = (int)hipGetLastError();
  return e ? -(long)e : total * 2;
}

int dua_upconv_k3_fwd(const dua_upconv_desc* d, const void* xskip, const void* u, const dua_in_norm* u_in, const void* w_skip_packed,
                      const void* wu_packed, const float* bias_table, void* y, dua_stat_word* out_stats, void* stream) {
  using namespace dua;
  if (!upconv_desc_ok(d) || !xskip || !u || !w_skip_packed || !wu_packed || !bias_table || !y || !out_stats) return DUA_ERR_ARG;
  if (u_in && u_in->stats && (!u_in->gamma || !u_in->beta || u_in->c_pad < d->Cu || u_in->count <= 0 ||
                             !(u_in->slope >= 0.f && u_in->slope <= 1.f)))
    return DUA_ERR_ARG;
  if (int e = ensure_prepared()) return e;
  UpConvArgs a{};
  a.xs = xskip; a.u = u; a.w = w_skip_packed; a.wu = wu_packed; a.btab = bias_table; a.y = y; a.stats = out_stats;
  a.xf = make_xform(u_in, d->Cu);
  a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W;
  a.Cs = d->Cskip; a.Cs_stride = d->Cskip_stride; a.Cs_off = d->Cskip_off; a.in_blk = (d->layout & DUA_IN_BLOCKED) ? 1 : 0;
  a.Cu = d->Cu; a.Cu_stride = d->Cu_stride; a.Cu_off = d->Cu_off;
  a.Cout = d->Cout; a.Cout_stride = d->Cout_stride; a.Cout_off = d->Cout_off; a.out_blk = (d->layout & DUA_OUT_BLOCKED) ? 1 : 0;
  a.cout_pad = (d->Cout + 63) / 64 * 64;
  a.nchunks = (d->Cskip + 31) / 32;
  a.tiles_h = d->H / 8; a.tiles_w = d->W / 8;
  a.ntiles = (d->D / 8) * a.tiles_h * a.tiles_w;
  const int lds = upc::HALO + upc::RING + 3 * 4 * d->Cu + 4 * 64 * 2 * 4;
  if (lds > 80 * 1024) return DUA_ERR_ARG;
  if (d->Cu == 64) hipLaunchKernelGGL(upconv_k3_kernel<1>, dim3(a.ntiles, a.cout_pad / 64, a.N), dim3(256), lds, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(upconv_k3_kernel<2>, dim3(a.ntiles, a.cout_pad / 64, a.N), dim3(256), lds, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

}  // extern "C"
