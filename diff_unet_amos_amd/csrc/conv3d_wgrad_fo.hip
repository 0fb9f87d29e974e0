// Weight gradient of the 3x3x3 convolution, "fetch once" form (round 4, fp16 operands):
//
//   dW[co][ci][kd][kh][kw] += sum_{n,d,h,w} dy[n,d,h,w,co] * x[n,d+kd-1,h+kh-1,w+kw-1,ci]
//
// (backward-by-weights of every Conv3d of the training step, train.py:258-268 through models/basic_unet/denoiser.py:56-59).
// The first form (conv3d_wgrad.hip) gives every kd plane of taps a workgroup of its own: the three of them fetch the same dy
// tile and overlapping x tiles, a tile is 24 MFMAs per wave behind 41.6 KB of transfers, and each launch writes 113 MB of
// fp32 partial tiles.  Here ONE workgroup (8 waves) owns all 27 taps of (64 output channels x a 32-channel input
// slice) and walks 4x8x8-voxel tiles: per tile the dy tile (256 voxels x 64 channels) and the x halo (6x10x10 voxels x 32
// channels) are fetched ONCE (70 KB by LDS-DMA, two buffers) for 864 MFMAs; a wave keeps one 32-channel half of the output
// channels x 7 taps = 7 accumulators of 32x32 (8 waves, two per SIMD: the 12-wave split at 168 registers spilled, and a spilled
// register reloaded between two transfers serialises them), so one output-gradient fragment feeds 7 MFMAs; the partial sums of
// a workgroup are 27 x 64 x 32 floats (56.6 MB per 96^3 launch).
// GEMM view as before: M = Cout, N = taps x Cin, K = voxels (the slow axis of channels-last data): tiles are staged as they
// lie in HBM ([voxel][32 channels], 64-byte rows) and read with ds_read_b64_tr_b16; a tap is an address offset on the x image.
#include "common.hpp"
#include "../../include/dua_hip.h"

namespace dua {

namespace wf {
constexpr int TD = 4, TH = 8, TW = 8, TV = TD * TH * TW;        // 256 output voxels per tile
constexpr int XD = TD + 2, XH = TH + 2, XW = TW + 2, XV = XD * XH * XW;   // 600 input voxels per tile
constexpr int NW = 8, NT = 64 * NW;                             // 8 waves: two per SIMD, 256 registers each
constexpr int RSB = 64;                                         // LDS row: 32 fp16 channels of one voxel
constexpr int XIMG = XV * RSB + 64, YIMG = TV * RSB + 64;       // x image; one 32-channel half of the dy image
constexpr int BUFB = XIMG + 2 * YIMG;                           // 71 360 bytes per buffer
constexpr int XP = (XV + 15) / 16, YP = 2 * (TV / 16);          // 1 KB pieces: 38 (the last one 8 voxels) + 32
constexpr int NPD = (XP + YP + NW - 1) / NW;                     // pieces per wave: 9
constexpr int NTAP = 7;                                         // taps per wave: grp, grp + 4, ... (grp = 3: six of them)
typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

struct Args {
  const void* x; const void* dy; float* part;
  int N, D, H, W;
  int Cin, Cin_stride, Cin_off;
  int Cout, Cout_stride, Cout_off;
  int ncs, ncombo, tiles_d, tiles_h, tiles_w, total_tiles, P;
};
}  // namespace wf

__device__ __forceinline__ void wf_glds16(const void* g, unsigned lds_wave_base) {
  // M0 written in the statement that reads it (nothing else in this kernel uses M0); s_nop 0: the ReadM0 -> LDS-DMA wait state
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_wave_base)) : "memory");
}

__global__ __launch_bounds__(wf::NT) void conv3d_k3_wgrad_fo_kernel(wf::Args a) {
  using namespace wf;
  using T = f16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: everything derived from it alone lives in SGPRs
  // block -> (partition, combo = (co tile, ci slice)): the combos of one partition read the same dy / x tiles, so they get block
  // ids 8 apart
  // (ids 8 apart = same XCD under round-robin placement: speed only)
  int part_id, combo;
  if (a.P % 8 == 0) {
    const int g = blockIdx.x / (8 * a.ncombo), r = blockIdx.x % (8 * a.ncombo);
    combo = r >> 3; part_id = g * 8 + (r & 7);
  } else {
    part_id = blockIdx.x % a.P; combo = blockIdx.x / a.P;
  }
  if (part_id >= a.P || combo >= a.ncombo) return;
  const int ct = combo / a.ncs, cs = combo % a.ncs;
  const int hl = lane >> 5;

  // ---- DMA pieces of this wave: piece q = wave + 8 k; q < XP: x voxels [16 q, 16 q + 16); else dy half (q - XP) / 16, voxels
  // 16 ((q - XP) % 16) ...; a lane moves 16 bytes = 8 channels (g4) of one voxel (vloc).  Nothing per piece is kept in
  // registers across tiles (coordinates are re-derived from q and the lane): a spilled register reloaded between two
  // transfers would put an s_waitcnt vmcnt in front of the next one -- vmcnt retires in order, so every piece would wait for
  // the one before it to LAND (measured: 567 us for 299 us of MFMA phases and 178 us of transfers on 96^3 64->64).
  const int vloc = lane >> 2, g4 = lane & 3;
  const bool xchan = cs * 32 + 8 * g4 < a.Cin;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const long vox = (long)a.D * a.H * a.W;
  // tile -> (sample, tile coordinates) without a division per tile: the walk advances by P tiles, so the coordinates advance by
  // P's own decomposition with carries (three runtime divisions per tile and wave were ~100 of the ~500 scalar instructions a
  // wave spends per tile, profiles/r4_pmc_wgrad_fo.json: 3.2 SALU instructions per MFMA)
  int nx_w, nx_h, nx_d, nx_n;                                  // coordinates of the NEXT tile to request (wave-uniform)
  {
    int t = part_id;
    nx_w = t % a.tiles_w; t /= a.tiles_w;
    nx_h = t % a.tiles_h; t /= a.tiles_h;
    nx_d = t % a.tiles_d; nx_n = t / a.tiles_d;
  }
  int st_w, st_h, st_d, st_n;                                  // P in the same mixed radix
  {
    int t = a.P;
    st_w = t % a.tiles_w; t /= a.tiles_w;
    st_h = t % a.tiles_h; t /= a.tiles_h;
    st_d = t % a.tiles_d; st_n = t / a.tiles_d;
  }
  auto advance = [&]() {
    nx_w += st_w;
    int c = nx_w >= a.tiles_w; nx_w -= c ? a.tiles_w : 0;
    nx_h += st_h + c;
    c = nx_h >= a.tiles_h; nx_h -= c ? a.tiles_h : 0;
    nx_d += st_d + c;
    c = nx_d >= a.tiles_d; nx_d -= c ? a.tiles_d : 0;
    nx_n += st_n + c;
  };
  auto issue_tile = [&](int buf) -> unsigned {                 // requests tile (nx_n, nx_d, nx_h, nx_w) and advances the walk
    unsigned okbits = 0;
    const int n = nx_n;
    const int d0 = nx_d * TD, h0 = nx_h * TH, w0 = nx_w * TW;
    advance();
    const T* xb = (const T*)a.x + n * vox * a.Cin_stride + a.Cin_off + cs * 32 + 8 * g4;
    const T* yb = (const T*)a.dy + n * vox * a.Cout_stride + a.Cout_off + ct * 64 + 8 * g4;
#pragma unroll
    for (int k = 0; k < NPD; ++k) {
      const int q = wave + NW * k;                               // wave-uniform
      if (q >= XP + YP) continue;
      if (q < XP) {
        const int v = 16 * q + vloc;
        const int pd = v / (XH * XW), rem = v - pd * (XH * XW), hy = rem / XW, hx = rem - hy * XW;
        const int gd = d0 - 1 + pd, gh = h0 - 1 + hy, gw = w0 - 1 + hx;
        const bool ok = xchan && v < XV && (unsigned)gd < (unsigned)a.D && (unsigned)gh < (unsigned)a.H && (unsigned)gw < (unsigned)a.W;
        okbits |= ok ? 1u << k : 0u;
        const int off = ok ? ((gd * a.H + gh) * a.W + gw) * a.Cin_stride : 0;        // (a sample's voxels x stride < 2^31: launcher)
        if (v < XV) wf_glds16(xb + off, lds0 + buf * BUFB + 16 * q * RSB);          // (the lanes behind the last x voxel move nothing)
      } else {
        const int qq = q - XP, h = qq / (TV / 16), b = qq % (TV / 16);
        const int v = 16 * b + vloc, gd = d0 + (v >> 6), gh = h0 + ((v >> 3) & 7), gw = w0 + (v & 7);
        const bool ok = ct * 64 + 32 * h + 8 * g4 < a.Cout && gd < a.D && gh < a.H && gw < a.W;
        okbits |= ok ? 1u << k : 0u;
        const int off = ok ? ((gd * a.H + gh) * a.W + gw) * a.Cout_stride + 32 * h : 0;
        wf_glds16(yb + off, lds0 + buf * BUFB + XIMG + h * YIMG + 16 * b * RSB);
      }
    }
    return okbits;
  };
  auto fix_tile = [&](int buf, unsigned okbits) {             // after the transfer has landed: zeros over what is not data
    f16x8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = (T)0.f;
#pragma unroll
    for (int k = 0; k < NPD; ++k) {
      const int q = wave + NW * k;
      if (q >= XP + YP || (okbits >> k & 1)) continue;
      if (q < XP) {
        if (16 * q + vloc < XV) *(f16x8*)(smem + buf * BUFB + 16 * q * RSB + lane * 16) = z;
      } else {
        const int qq = q - XP;
        *(f16x8*)(smem + buf * BUFB + XIMG + (qq / (TV / 16)) * YIMG + 16 * (qq % (TV / 16)) * RSB + lane * 16) = z;
      }
    }
  };

  // ---- roles: wave = (32-channel half of the output channels coh, tap group grp): taps grp, grp + 4, ..., i.e. 7 accumulators
  // of 32x32 (6 for grp = 3) that share ONE output-gradient fragment per k-step ----
  const int coh = wave & 1, grp = wave >> 1;
  const int ntap = grp < 3 ? 7 : 6;
  f32x16 acc[NTAP];
#pragma unroll
  for (int j = 0; j < NTAP; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  // operand fetch (the lane constants of conv3d_wgrad.hip): a transposed read returns, for the 16 lanes of a quarter wave, a
  // 4-voxel x 16-channel block channel-per-lane; two reads (voxel quads r = 0, 1) make the 8 k-values of a lane
  const int i16 = lane & 15, qv = i16 >> 2, pch = i16 & 3, g1 = (lane >> 4) & 1;
  const int b_col = (16 * g1 + 4 * pch) * 2;
  // byte offsets inside a buffer: dy fragment (+ step * 16 * RSB), x fragment (+ step offset + tap offset)
  const int ya0 = XIMG + coh * YIMG + (8 * hl + qv) * RSB + b_col;      // (+ 4 * RSB: the second voxel quad)
  const int xa0 = (hl * XW + qv) * RSB + b_col;
  int tap_off[NTAP];                                             // the tap's shift inside the x image (wave-uniform)
#pragma unroll
  for (int j = 0; j < NTAP; ++j) {
    const int t = j < ntap ? grp + 4 * j : grp;
    const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
    tap_off[j] = ((kd * XH + kh) * XW + kw) * RSB;
  }

  typedef __attribute__((address_space(3))) h4* lds_h4p;
  auto kloop = [&](int bufoff) {
    // Every wave runs SEVEN taps per k-step: the two waves with six real taps (grp = 3) repeat their first tap into an accumulator
    // nobody stores.  They finish a tile no later than the seven-tap waves they meet at the barrier, and the wave-uniform test
    // "does my seventh tap exist" -- two branches per k-step in front of the fragment reads and the MFMAs -- is gone.
    constexpr int NTW = NTAP;
    // k-step s: voxels 16 s + 8 hl + 4 r + q: d = s >> 2, h = 2 (s & 3) + hl, w = 4 r + q; the eight fragments of step s + 1 are
    // requested before the seven MFMAs of step s issue.  (One x-fragment set, each fragment re-requested right behind the MFMA
    // that consumed it, with the transfers issued from inside the loop for interior tiles, was built to shorten the head of
    // the tile: 476 -> 555 us on 96^3 64->64; not kept.)
    // Fragment addresses: ONE vector register per tap (lane offset + the tap's shift + the buffer) and one for dy, made once per
    // tile; the k-step's offset is a compile-time constant and rides in the instruction's offset field.  Written as
    // buf + lane offset + step offset + tap_off[j] the compiler kept the 16 x 7 scalar sums (step, tap) live across the tile
    // loop -- 92 scalar registers spilled into vector lanes (v_readlane per use) and a v_add in front of every fragment read.
    const unsigned yv = lds0 + (unsigned)(bufoff + ya0);
    unsigned xv[NTAP];
#pragma unroll
    for (int j = 0; j < NTW; ++j) xv[j] = lds0 + (unsigned)(bufoff + xa0 + tap_off[j]);
    auto rd = [&](int s, f16x8& pa, f16x8* pb) {
      const int yo = s * 16 * RSB, xo = ((s >> 2) * XH + 2 * (s & 3)) * XW * RSB;
      {
        h4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4p)(size_t)(yv + yo));
        h4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4p)(size_t)(yv + yo + 4 * RSB));
        pa = __builtin_shufflevector(__builtin_bit_cast(f16x4, lo), __builtin_bit_cast(f16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        h4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4p)(size_t)(xv[j] + xo));
        h4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4p)(size_t)(xv[j] + xo + 4 * RSB));
        pb[j] = __builtin_shufflevector(__builtin_bit_cast(f16x4, lo), __builtin_bit_cast(f16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
      }
    };
    auto mm = [&](const f16x8& pa, const f16x8* pb) {
#pragma unroll
      for (int j = 0; j < NTW; ++j) mma32(acc[j], pa, pb[j]);
    };
    f16x8 A0, A1, B0[NTAP], B1[NTAP];
    rd(0, A0, B0);
#pragma unroll
    for (int s = 0; s < TV / 16; s += 2) {
      rd(s + 1, A1, B1);
      __builtin_amdgcn_sched_barrier(0);
      mm(A0, B0);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 2 < TV / 16) rd(s + 2, A0, B0);
      __builtin_amdgcn_sched_barrier(0);
      mm(A1, B1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- tile loop: two buffers, one tile in flight ----
  int tile = part_id, cur = 0;
  unsigned okbits = 0;
  if (tile < a.total_tiles) okbits = issue_tile(0);
  for (; tile < a.total_tiles; tile += a.P) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this tile has landed ...
    fix_tile(cur, okbits);
    __syncthreads();                                           // ... for every wave, and the other buffer's readers are through
#if !defined(WF_ABL) || WF_ABL != 1        // diagnostic builds: WF_ABL=1 no transfers, 2 no k loop
    if (tile + a.P < a.total_tiles) okbits = issue_tile(cur ^ 1);
#endif
#if !defined(WF_ABL) || WF_ABL != 2
    kloop(cur * BUFB);
#endif
    cur ^= 1;
  }

  // ---- partial sums [P][combo][27 taps][64 co][32 ci]; acc[j]: lane column = ci (lane & 31), register i -> co row ----
  float* pp = a.part + ((long)part_id * a.ncombo + combo) * 27 * 2048 + (lane & 31);
#pragma unroll
  for (int j = 0; j < NTAP; ++j) {
    if (j == NTAP - 1 && ntap < NTAP) continue;
    const int t = grp + 4 * j;
#pragma unroll
    for (int i = 0; i < 16; ++i) pp[t * 2048 + (coh * 32 + acc_row(i, hl)) * 32] = acc[j][i];
  }
}

// dw[co][ci_src][tap] += sum_p part[p][combo][tap][co 64][ci 32]; one thread per (combo, tap, co, ci); fixed summation order
__global__ __launch_bounds__(256) void wgrad_fo_reduce_kernel(const float* __restrict__ part, int P, int ncs, int ncombo, int Cin,
                                                              int Cin_src, int Cout, const int* __restrict__ perm,
                                                              float* __restrict__ dw) {
  const long per_p = (long)ncombo * 27 * 2048;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < per_p; i += (long)gridDim.x * 256) {
    const int cil = (int)(i & 31), col = (int)((i >> 5) & 63);
    const int tap = (int)((i >> 11) % 27), combo = (int)((i >> 11) / 27);
    const int co = (combo / ncs) * 64 + col, cip = (combo % ncs) * 32 + cil;
    if (co >= Cout || cip >= Cin) continue;
    const int ci = perm ? perm[cip] : cip;
    if (ci < 0 || ci >= Cin_src) continue;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int p = 0;
    for (; p + 15 < P; p += 16) {
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = part[(long)(p + j) * per_p + i];
#pragma unroll
      for (int j = 0; j < 16; j += 4) { s0 += v[j]; s1 += v[j + 1]; s2 += v[j + 2]; s3 += v[j + 3]; }
    }
    for (; p + 3 < P; p += 4) {
      s0 += part[(long)p * per_p + i]; s1 += part[(long)(p + 1) * per_p + i];
      s2 += part[(long)(p + 2) * per_p + i]; s3 += part[(long)(p + 3) * per_p + i];
    }
    for (; p < P; ++p) s0 += part[(long)p * per_p + i];
    dw[((long)co * Cin_src + ci) * 27 + tap] += (s0 + s1) + (s2 + s3);
  }
}

// The same sum for the layers with MANY (co, ci) slabs and few partitions (the <= 12^3 levels: 32-128 slabs, dW of 7-28 MB): there the
// one-thread-per-element kernel above spends its time on the OUTPUT -- consecutive lanes are consecutive ci, 27 floats apart in
// dW[co][ci][tap], so a wave's read-modify-write touched ~55 lines for 256 useful bytes and every line 27 times over the launch
// (73 us for the 6^3 512 -> 512 layer, most of that launch).  Here a workgroup owns 8 output channels x 32 input channels of one
// slab for ALL 27 taps: a thread sums the 27 taps of its (co, ci) over the partitions (coalesced reads, fixed order), the sums
// pass through LDS in dW's own order, and dW is updated in whole 1 KB runs.  Needs the identity channel map.
__global__ __launch_bounds__(256) void wgrad_fo_reduce_taps_kernel(const float* __restrict__ part, int P, int ncs, int ncombo, int Cin,
                                                                   int Cin_src, int Cout, float* __restrict__ dw) {
  __shared__ float st[256 * 27];
  const int combo = blockIdx.x >> 3, cob = blockIdx.x & 7, t = threadIdx.x;
  const long per_p = (long)ncombo * 27 * 2048;
  const float* src = part + ((long)combo * 27 * 64 + cob * 8 + (t >> 5)) * 32 + (t & 31);
  float s[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) s[k] = 0.f;
  for (int p = 0; p < P; ++p) {
    float v[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) v[k] = src[(long)p * per_p + k * 2048];
#pragma unroll
    for (int k = 0; k < 27; ++k) s[k] += v[k];
  }
#pragma unroll
  for (int k = 0; k < 27; ++k) st[t * 27 + k] = s[k];          // stride 27 (odd): conflict-free
  __syncthreads();
  const int co0 = (combo / ncs) * 64 + cob * 8, ci0 = (combo % ncs) * 32;
  const int nci = min(min(Cin, Cin_src) - ci0, 32);              // valid input channels of this slab
#pragma unroll
  for (int j = 0; j < 27; ++j) {
    const int e = j * 256 + t, col = e / 864, w = e - col * 864;  // 864 = 32 ci x 27 taps: one output channel's run
    if (co0 + col < Cout && w < nci * 27) dw[((long)(co0 + col) * Cin_src + ci0) * 27 + w] += st[e];
  }
}

static const LdsAttr kWgradFoLdsAttrs[] = {{(const void*)conv3d_k3_wgrad_fo_kernel, 2 * wf::BUFB}};
static const LdsAttrs kWgradFoLdsReg(kWgradFoLdsAttrs);

// partitions of the tile list: mult workgroups per CU over the launch (one resident at a time: 12 waves, 143 KB of LDS)
int wgrad_fo_partitions(const dua_conv3_desc* d, int* combos_out) {
  using namespace wf;
  const int combos = ((d->Cout + 63) / 64) * ((d->Cin + 31) / 32);
  const int total = d->N * ((d->D + TD - 1) / TD) * ((d->H + TH - 1) / TH) * ((d->W + TW - 1) / TW);
  const int mv = (d->policy & 31) >> 1;
  const int mult = mv ? mv : 1;
  int P = (256 * mult + combos - 1) / combos;
  if (P > total) P = total;
  if (P >= 8) P &= ~7;                                          // whole groups of 8 (block ids 8 apart share an XCD)
  if (combos_out) *combos_out = combos;
  return P < 1 ? 1 : P;
}

long wgrad_fo_workspace(const dua_conv3_desc* d) {
  int combos;
  const int P = wgrad_fo_partitions(d, &combos);
  return (long)P * combos * 27 * 2048 * 4;
}

int launch_wgrad_fo(const dua_conv3_desc* d, const void* x, const void* dy, float* dw, int Cin_src, const int* perm, float* ws,
                    long ws_bytes, hipStream_t s) {
  using namespace wf;
  if (int e = ensure_prepared()) return e;
  Args a;
  a.x = x; a.dy = dy; a.part = ws;
  a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W;
  a.Cin = d->Cin; a.Cin_stride = d->Cin_stride; a.Cin_off = d->Cin_off;
  a.Cout = d->Cout; a.Cout_stride = d->Cout_stride; a.Cout_off = d->Cout_off;
  a.ncs = (d->Cin + 31) / 32;
  a.tiles_d = (d->D + TD - 1) / TD; a.tiles_h = (d->H + TH - 1) / TH; a.tiles_w = (d->W + TW - 1) / TW;
  a.total_tiles = d->N * a.tiles_d * a.tiles_h * a.tiles_w;
  a.P = wgrad_fo_partitions(d, &a.ncombo);
  if (!ws || wgrad_fo_workspace(d) > ws_bytes) return DUA_ERR_ARG;
  const long voxs = (long)d->D * d->H * d->W;
  if (voxs * d->Cin_stride >= 0x7fffffffL || voxs * d->Cout_stride >= 0x7fffffffL) return DUA_ERR_ARG;   // 32-bit offsets inside a sample
  hipLaunchKernelGGL(conv3d_k3_wgrad_fo_kernel, dim3(a.P * a.ncombo), dim3(NT), 2 * BUFB, s, a);
  const long per_p = (long)a.ncombo * 27 * 2048;
  if (a.ncombo >= 32 && !perm) {                                 // many slabs, few partitions: the output side decides (see the kernel)
    hipLaunchKernelGGL(wgrad_fo_reduce_taps_kernel, dim3(a.ncombo * 8), dim3(256), 0, s, ws, a.P, a.ncs, a.ncombo, d->Cin, Cin_src,
                       d->Cout, dw);
    return (int)hipGetLastError();
  }
  long nb = (per_p + 255) / 256;
  hipLaunchKernelGGL(wgrad_fo_reduce_kernel, dim3((unsigned)(nb > 8192 ? 8192 : nb)), dim3(256), 0, s, ws, a.P, a.ncs, a.ncombo,
                     d->Cin, Cin_src, d->Cout, perm, dw);
  return (int)hipGetLastError();
}

}  // namespace dua
