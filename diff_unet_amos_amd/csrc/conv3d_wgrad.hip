// Weight gradient of the 3x3x3 (pad 1, stride 1) convolution on the gfx950 matrix cores.
//
//   dW[co][ci][kd][kh][kw] += sum_{n,d,h,w} dy[n,d,h,w,co] * x[n,d+kd-1,h+kh-1,w+kw-1,ci]
//
// Backward-by-weights of every Conv3d the training step reaches (train.py:258-268 -> loss.backward() through
// models/basic_unet/denoiser.py:56-59 and pretrained/basic_unet.py:60-63).
//
// GEMM view: M = Cout, N = 27 taps x Cin, K = voxels -- the contraction runs over the axis that is SLOW in the
// channels-last activations, so both MFMA operands are k-strided in memory.  f16: the tiles are staged in LDS
// exactly as they lie in HBM ([voxel][32 channels], 64-byte rows) and read with ds_read_b64_tr_b16, gfx950's
// transposing LDS read (a 4-voxel x 16-channel block per 16 lanes, delivered channel-per-lane); a tap shift is an
// address offset on the x image.  f32 (parity mode): MFMA 32x32x2 takes one k per lane, plain ds_read_b32.
//
// Workgroup = one kd plane of taps x 64 co x 64 ci, persistent over a strided list of 2x8x8-voxel tiles.  f16: 768
// threads = 12 waves, wave (kh, ci half, co half) keeps 3 (kw) 32x32 fp32 accumulators (three waves per SIMD, every
// SIMD the same MFMA load); f32 / the 6-wave form: wave (kh, ci half) keeps 3 (kw) x 2 (co half) accumulators for its
// whole tile list and adds them to dW (reference layout, fp32 atomics) once at the end.  The next tile's x and
// dy are prefetched into registers while the current one is multiplied.
#include "common.hpp"
#include "../../include/dua_hip.h"

namespace dua {
#ifdef DUA_ABLATE
int g_wgrad_abl = 0;       // diagnostic builds: dua_set_option(3, mask)
#endif
// launch form of a call = dua_conv3_desc.policy: bit 0 = plain (partition-major) block order, bits 1-4 = workgroups per CU over the launch (0 = policy), bit 5 = 6 waves + plain k loop, bit 6 = 6 waves + pipelined k loop (default: 12 waves)
namespace wg {
constexpr int TD = 2, TH = 8, TW = 8, TV = TD * TH * TW;      // 128 output voxels per tile
constexpr int XH = TH + 2, XW = TW + 2, XV = TD * XH * XW;    // 200 input voxels per tile and kd
constexpr int NT = 384;
typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

struct Args {
  const void* x; const void* dy; float* dw; const int* perm; float* part;
  int N, D, H, W;
  int Cin, Cin_stride, Cin_off, Cin_src;
  int Cout, Cout_stride, Cout_off;
  int ncc, ncombo, tiles_d, tiles_h, tiles_w, total_tiles, P;
  int plain_order;
  int abl;     // diagnostic builds (-DDUA_ABLATE): 1 = no MFMA, 2 = no fragment reads, 4 = no global loads, 8 = no LDS stores
};

}  // namespace wg

// 16 bytes per lane from global memory straight into LDS: lane i of the wave lands at lds_wave_base + 16 i
// Written as inline assembly on purpose: behind the builtin hipcc (ROCm 7.2) cannot tell which LDS bytes a pending transfer
// will write and puts an s_waitcnt vmcnt(0) in front of the next LDS read -- the k loop would wait for the transfer it is
// meant to run under.  The loop below waits for its transfers itself (s_waitcnt vmcnt(0) + barrier before a buffer is read).
__device__ __forceinline__ void wg_glds16(const void* g, void* lds_wave_base) {
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base;
  // M0 (the LDS base of the transfer) is compiler-reserved and not preserved around an asm statement, so it is written in
  // the SAME statement that reads it (cdna_hip_programming.md 5.7).  Nothing else in this kernel uses M0 (no movrel, no
  // LDS-DMA builtin, no sendmsg), so it is not saved and restored: that form (kept under -DDUA_GLDS_SAVE_M0, and used by the
  // conv kernel's weight ring) costs this kernel 1-4 % (646 vs 622 us on 96^3 64->64, same box).  M0 holds a full LDS
  // byte address; the second buffer of this kernel starts above 64 KB.
#ifdef DUA_GLDS_SAVE_M0
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(__builtin_amdgcn_readfirstlane(base)) : "memory");
#else
  // (s_nop 0: the one wait state the ReadM0 -> LDS-DMA hazard asks for; hipcc does not look inside an asm statement)
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(__builtin_amdgcn_readfirstlane(base)) : "memory");
#endif
}

// DMA (f16, 12 waves): the tiles go from global memory straight into one of TWO LDS buffers (global_load_lds), issued one
// tile ahead -- no staging registers, no store phase, one barrier per tile; pieces outside the volume are overwritten
// with zeros once the transfer has landed.
template <typename T, bool PIPE, int NCOH, bool DMA = false>
__device__ __forceinline__ void wgrad_body(const wg::Args& a) {
  constexpr int NT = NCOH == 2 ? 384 : 768;          // 6 waves (two co halves each) or 12 waves (one co half each)
  using namespace wg;
  using Frag = typename Elem<T>::Frag;
  constexpr int EPG = Elem<T>::EPG;
  constexpr int G = 64 / EPG;                      // 16-byte groups per voxel per 64-channel chunk
  constexpr int RSB = 32 * (int)sizeof(T);         // LDS row: 32 channels of one voxel
  // one 32-channel half image; +64 B so that the two halves of a voxel, written by the 8 lanes of one ds_write_b128
  // group, fall into different halves of the 32 write banks (stores bank on (a/4) mod 32, not mod 64: a first +128 B
  // pad changed nothing) -- unpadded, SQ_LDS_BANK_CONFLICT was 14 % of the LDS cycles, all on the staging writes
  constexpr int XIMG = XV * RSB + 64, YIMG = TV * RSB + 64;
  constexpr int NX = (XV * G + NT - 1) / NT, NY = (TV * G + NT - 1) / NT;
  constexpr int KV = sizeof(T) == 2 ? 16 : 8;      // voxels per mma32 call
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xs = smem;                 // [2][XV][32]
  char* Ys = smem + 2 * XIMG;      // [2][TV][32]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kh = wave % 3, cih = (wave / 3) & 1, coh0 = NCOH == 2 ? 0 : wave / 6;
  // 1-D grid, XCD-aware: block b runs on XCD b % 8; the three kd workgroups of one (partition, co tile, ci chunk) get
  // ids 24 j + 8 kd + x, i.e. the same XCD, dispatched together -- they read the same x / dy tiles through one L2.
  const int xcd = blockIdx.x & 7, rr = blockIdx.x >> 3;
  int kd = rr % 3, pc = (rr / 3) * 8 + xcd;
  if (a.plain_order) { pc = blockIdx.x % (a.P * a.ncombo); kd = blockIdx.x / (a.P * a.ncombo); if (kd > 2) return; }
  if (pc >= a.P * a.ncombo) return;
  const int part_id = pc % a.P, combo = pc / a.P;
  const int ct = combo / a.ncc, cc = combo % a.ncc;
  const int hl = lane >> 5;

  // per-thread staging items: x item -> (pd, hy, hx, g), dy item -> (voxel, g)
  int xl[NX], xc[NX], yl[NY], yc[NY];      // LDS byte offset (-1: no item) / packed source coordinates (-1: zero fill)
#pragma unroll
  for (int j = 0; j < NX; ++j) {
    const int it = tid + NT * j, v = it / G, g = it % G;
    const int pd = v / (XH * XW), rem = v % (XH * XW), hy = rem / XW, hx = rem % XW;
    const bool inl = it < XV * G;
    xc[j] = inl && cc * 64 + g * EPG < a.Cin ? (pd << 16) | (hy << 8) | hx : -1;
    xl[j] = inl ? ((g * EPG) >> 5) * XIMG + v * RSB + ((g * EPG) & 31) * (int)sizeof(T) : -1;
  }
#pragma unroll
  for (int j = 0; j < NY; ++j) {
    const int it = tid + NT * j, v = it / G, g = it % G;
    const bool inl = it < TV * G;
    yc[j] = inl && ct * 64 + g * EPG < a.Cout ? v : -1;
    yl[j] = inl ? ((g * EPG) >> 5) * YIMG + v * RSB + ((g * EPG) & 31) * (int)sizeof(T) : -1;
  }
  const int xg_off = cc * 64, yg_off = ct * 64;
  // DMA slots: one wave instruction moves 16 voxels x 32 channels (1 KB) of ONE half image: x = 2 halves x 13 voxel blocks
  // (the last one 8 voxels), dy = 2 x 8; wave w takes instructions w, w + 12, (w + 24)
  constexpr int XB = (XV + 15) / 16, NXD = (2 * XB + 11) / 12, NYD = (2 * (TV / 16) + 11) / 12;
  int dxc[NXD], dxg[NXD], dxl[NXD], dyc[NYD], dyg[NYD], dyl[NYD];   // packed coords (-1: lane has no piece) / channel / LDS base
  bool dxch[NXD], dych[NYD];                                        // the piece's channels exist
  if constexpr (DMA) {
    const int vloc = lane >> 2, g4 = lane & 3;
#pragma unroll
    for (int k = 0; k < NXD; ++k) {
      const int q = wave + 12 * k, h = q / XB, b = q % XB, v = 16 * b + vloc;
      const int pd = v / (XH * XW), rem = v % (XH * XW), hy = rem / XW, hx = rem % XW;
      dxc[k] = (q < 2 * XB && v < XV) ? (pd << 16) | (hy << 8) | hx : -1;
      dxg[k] = 32 * h + 8 * g4;
      dxch[k] = cc * 64 + dxg[k] < a.Cin;
      dxl[k] = h * XIMG + 16 * b * RSB;
    }
#pragma unroll
    for (int k = 0; k < NYD; ++k) {
      const int q = wave + 12 * k, h = q / (TV / 16), b = q % (TV / 16);
      dyc[k] = q < 2 * (TV / 16) ? 16 * b + vloc : -1;
      dyg[k] = 32 * h + 8 * g4;
      dych[k] = ct * 64 + dyg[k] < a.Cout;
      dyl[k] = h * YIMG + 16 * b * RSB;
    }
  }
  // issue_tile returns one bit per slot: the piece lies inside the volume and its channels exist (fix_tile zeroes the others)
  auto issue_tile = [&](int tile, char* buf) -> unsigned {
    unsigned okbits = 0;
    int t = tile;
    const int tw_ = t % a.tiles_w; t /= a.tiles_w;
    const int th_ = t % a.tiles_h; t /= a.tiles_h;
    const int td_ = t % a.tiles_d; const int n = t / a.tiles_d;
    const int d0 = td_ * TD, h0 = th_ * TH, w0 = tw_ * TW;
    const T* xb = (const T*)a.x + (long)n * a.D * a.H * a.W * a.Cin_stride + a.Cin_off + xg_off;
    const T* yb = (const T*)a.dy + (long)n * a.D * a.H * a.W * a.Cout_stride + a.Cout_off + yg_off;
#pragma unroll
    for (int k = 0; k < NXD; ++k) {
      const int gd = d0 + kd - 1 + (dxc[k] >> 16), gh = h0 - 1 + ((dxc[k] >> 8) & 255), gw = w0 - 1 + (dxc[k] & 255);
      const bool ok = dxch[k] && (unsigned)gd < (unsigned)a.D && (unsigned)gh < (unsigned)a.H && (unsigned)gw < (unsigned)a.W;
      okbits |= ok ? 1u << k : 0u;
      const long off = ok ? (((long)gd * a.H + gh) * a.W + gw) * a.Cin_stride + dxg[k] : 0;
      if (dxc[k] >= 0) wg_glds16(xb + off, buf + dxl[k]);
    }
#pragma unroll
    for (int k = 0; k < NYD; ++k) {
      const int v = dyc[k], gd = d0 + (v >> 6), gh = h0 + ((v >> 3) & 7), gw = w0 + (v & 7);
      const bool ok = dych[k] && gd < a.D && gh < a.H && gw < a.W;
      okbits |= ok ? 1u << (NXD + k) : 0u;
      const long off = ok ? (((long)gd * a.H + gh) * a.W + gw) * a.Cout_stride + dyg[k] : 0;
      if (dyc[k] >= 0) wg_glds16(yb + off, buf + 2 * XIMG + dyl[k]);
    }
    return okbits;
  };
  auto fix_tile = [&](char* buf, unsigned okbits) {   // after the transfer has landed: zeros over the pieces that are not data
    Frag z;
#pragma unroll
    for (int e = 0; e < EPG; ++e) z[e] = (T)0.f;
#pragma unroll
    for (int k = 0; k < NXD; ++k)
      if (dxc[k] >= 0 && !(okbits >> k & 1)) *(Frag*)(buf + dxl[k] + lane * 16) = z;
#pragma unroll
    for (int k = 0; k < NYD; ++k)
      if (dyc[k] >= 0 && !(okbits >> (NXD + k) & 1)) *(Frag*)(buf + 2 * XIMG + dyl[k] + lane * 16) = z;
  };

  Frag xr[NX], yr[NY];
  auto zero = [](Frag& f) {
#pragma unroll
    for (int e = 0; e < EPG; ++e) f[e] = (T)0.f;
  };
  auto load_tile = [&](int tile) {
#ifdef DUA_ABLATE
    if (a.abl & 4) return;
#endif
    int t = tile;
    const int tw_ = t % a.tiles_w; t /= a.tiles_w;
    const int th_ = t % a.tiles_h; t /= a.tiles_h;
    const int td_ = t % a.tiles_d; const int n = t / a.tiles_d;
    const int d0 = td_ * TD, h0 = th_ * TH, w0 = tw_ * TW;
    const T* xb = (const T*)a.x + (long)n * a.D * a.H * a.W * a.Cin_stride + a.Cin_off + xg_off;
    const T* yb = (const T*)a.dy + (long)n * a.D * a.H * a.W * a.Cout_stride + a.Cout_off + yg_off;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      zero(xr[j]);
      if (xc[j] >= 0) {
        const int gd = d0 + kd - 1 + (xc[j] >> 16), gh = h0 - 1 + ((xc[j] >> 8) & 255), gw = w0 - 1 + (xc[j] & 255);
        const int g = (tid + NT * j) % G;
        if (gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W)
          xr[j] = *(const Frag*)(xb + (((long)gd * a.H + gh) * a.W + gw) * a.Cin_stride + g * EPG);
      }
    }
#pragma unroll
    for (int j = 0; j < NY; ++j) {
      zero(yr[j]);
      if (yc[j] >= 0) {
        const int v = yc[j], gd = d0 + (v >> 6), gh = h0 + ((v >> 3) & 7), gw = w0 + (v & 7);
        const int g = (tid + NT * j) % G;
        if (gd < a.D && gh < a.H && gw < a.W)
          yr[j] = *(const Frag*)(yb + (((long)gd * a.H + gh) * a.W + gw) * a.Cout_stride + g * EPG);
      }
    }
  };
  auto store_tile = [&]() {
#ifdef DUA_ABLATE
    if (a.abl & 8) return;
#endif
#pragma unroll
    for (int j = 0; j < NX; ++j)
      if (xl[j] >= 0) *(Frag*)(Xs + xl[j]) = xr[j];
#pragma unroll
    for (int j = 0; j < NY; ++j)
      if (yl[j] >= 0) *(Frag*)(Ys + yl[j]) = yr[j];
  };

  f32x16 acc[3][NCOH];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < NCOH; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // lane constants of the operand fetch
  int a_off[2], b_row[2], b_col;       // f16: two transposed reads (r = 0, 1); f32: unused slots
  if constexpr (sizeof(T) == 2) {
    const int i16 = lane & 15, q = i16 >> 2, p = i16 & 3, g1 = (lane >> 4) & 1;
    b_col = (16 * g1 + 4 * p) * 2;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      a_off[r] = (8 * hl + 4 * r + q) * RSB + b_col;     // + step * 16 * RSB + coh * YIMG
      b_row[r] = hl * XW + 4 * r + q;                     // x row (within plane) = (hrow + kh) * XW + w + kw
    }
  } else {
    b_col = (lane & 31) * 4;
    a_off[0] = a_off[1] = 0; b_row[0] = b_row[1] = 0;
  }

  auto kloop = [&]() {
    if constexpr (PIPE && sizeof(T) == 2) {
      // explicit two-stage pipeline: the 10 transposed reads of k-step s+1 are issued before the 6 MFMAs of step s
      const char* ya = Ys + a_off[0];
      const char* yb = Ys + a_off[1];
      const char* xa = Xs + cih * XIMG + (kh * XW + b_row[0]) * RSB + b_col;
      const char* xb = Xs + cih * XIMG + (kh * XW + b_row[1]) * RSB + b_col;
      auto rd = [&](int s, Frag* pa, Frag* pb) {
        const int yo = s * 16 * RSB, xo = ((s >> 2) * XH + 2 * (s & 3)) * XW * RSB;
#pragma unroll
        for (int coh = 0; coh < NCOH; ++coh) {
          h4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(ya + (coh0 + coh) * YIMG + yo));
          h4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(yb + (coh0 + coh) * YIMG + yo));
          pa[coh] = __builtin_shufflevector(__builtin_bit_cast(f16x4, lo), __builtin_bit_cast(f16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          h4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(xa + xo + kw * RSB));
          h4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(xb + xo + kw * RSB));
          pb[kw] = __builtin_shufflevector(__builtin_bit_cast(f16x4, lo), __builtin_bit_cast(f16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
        }
      };
      auto mm = [&](const Frag* pa, const Frag* pb) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int coh = 0; coh < NCOH; ++coh) mma32(acc[kw][coh], pa[coh], pb[kw]);
      };
      Frag A0[NCOH], B0[3], A1[NCOH], B1[3];
      rd(0, A0, B0);
#pragma unroll
      for (int s = 0; s < TV / KV; s += 2) {
        rd(s + 1, A1, B1);
        __builtin_amdgcn_sched_barrier(0);
        mm(A0, B0);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < TV / KV) rd(s + 2, A0, B0);
        __builtin_amdgcn_sched_barrier(0);
        mm(A1, B1);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else
#pragma unroll 2
    for (int s = 0; s < TV / KV; ++s) {
      Frag fa[NCOH], fb[3];
#ifdef DUA_ABLATE
      if (a.abl & 2) {
        for (int e = 0; e < EPG; ++e) { fa[0][e] = fa[NCOH - 1][e] = (T)(float)s; fb[0][e] = fb[1][e] = fb[2][e] = (T)(float)lane; }
      } else
#endif
      if constexpr (sizeof(T) == 2) {
        // voxels 16 s + 8 hl + 4 r + q: d = s >> 2, hrow = 2 (s & 3) + hl, w = 4 r + q
        const int d = s >> 2, hr0 = 2 * (s & 3);
#pragma unroll
        for (int coh = 0; coh < NCOH; ++coh) {
          h4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
              (__attribute__((address_space(3))) h4*)(Ys + (coh0 + coh) * YIMG + s * 16 * RSB + a_off[0]));
          h4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
              (__attribute__((address_space(3))) h4*)(Ys + (coh0 + coh) * YIMG + s * 16 * RSB + a_off[1]));
          f16x4 l4 = __builtin_bit_cast(f16x4, lo), h4v = __builtin_bit_cast(f16x4, hi);
          fa[coh] = __builtin_shufflevector(l4, h4v, 0, 1, 2, 3, 4, 5, 6, 7);
        }
        const int rowbase = (d * XH + hr0 + kh) * XW;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          h4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
              (__attribute__((address_space(3))) h4*)(Xs + cih * XIMG + (rowbase + b_row[0] + kw) * RSB + b_col));
          h4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
              (__attribute__((address_space(3))) h4*)(Xs + cih * XIMG + (rowbase + b_row[1] + kw) * RSB + b_col));
          f16x4 l4 = __builtin_bit_cast(f16x4, lo), h4v = __builtin_bit_cast(f16x4, hi);
          fb[kw] = __builtin_shufflevector(l4, h4v, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      } else {
        // voxels 8 s + 2 e + hl (e = 0..3): one w row; d = s >> 3, hrow = s & 7, w = 2 e + hl
        const int d = s >> 3, hrow = s & 7;
#pragma unroll
        for (int coh = 0; coh < NCOH; ++coh)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            fa[coh][e] = *(const float*)(Ys + (coh0 + coh) * YIMG + (s * 8 + 2 * e + hl) * RSB + b_col);
        const int rowbase = (d * XH + hrow + kh) * XW;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            fb[kw][e] = *(const float*)(Xs + cih * XIMG + (rowbase + 2 * e + hl + kw) * RSB + b_col);
      }
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int coh = 0; coh < NCOH; ++coh) {
#ifdef DUA_ABLATE
          if (a.abl & 1) { acc[kw][coh][0] += (float)fa[coh][0] * (float)fb[kw][0]; continue; }
#endif
          mma32(acc[kw][coh], fa[coh], fb[kw]);
        }
    }
  };
  int tile = part_id;
  if constexpr (DMA) {
    // Two buffers, ONE tile in flight.  (Three buffers with two tiles in flight -- the transfers cost no registers -- were
    // measured: 5-8 % slower on every layer shape, like the two-register-set form of the other path: more of this kernel's
    // traffic in flight makes the memory side slower, not faster.)
    constexpr int BUFB = 2 * XIMG + 2 * YIMG;
    int cur = 0;
    unsigned okbits = 0;
    if (tile < a.total_tiles) okbits = issue_tile(tile, smem);
    for (; tile < a.total_tiles; tile += a.P) {
      char* buf = smem + cur * BUFB;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this tile has landed ...
      fix_tile(buf, okbits);
      __syncthreads();                                     // ... for every wave, and the other buffer's readers are through
      Xs = buf; Ys = buf + 2 * XIMG;
      if (tile + a.P < a.total_tiles) okbits = issue_tile(tile + a.P, smem + (cur ^ 1) * BUFB);
      kloop();
      cur ^= 1;
    }
  } else {
  if (tile < a.total_tiles) load_tile(tile);
  for (; tile < a.total_tiles; tile += a.P) {
    __syncthreads();                       // previous tile's fragment reads are done
    store_tile();
    __syncthreads();
    if (tile + a.P < a.total_tiles) load_tile(tile + a.P);
    kloop();
  }
  }

  // acc[kw][coh]: lane column = ci (lane & 31), register i -> co row acc_row(i, hl)
  if (a.part) {
    // partial sums [P][combo][kd][kh][kw][co 64][ci 64], summed by wgrad_reduce_kernel (no atomics: every workgroup
    // would otherwise hit the same Cout*Cin*27 addresses)
    float* pp = a.part + ((((long)part_id * a.ncombo + combo) * 3 + kd) * 9 + kh * 3) * 4096 + cih * 32 + (lane & 31);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int coh = 0; coh < NCOH; ++coh)
#pragma unroll
        for (int i = 0; i < 16; ++i) pp[kw * 4096 + ((coh0 + coh) * 32 + acc_row(i, hl)) * 64] = acc[kw][coh][i];
    return;
  }
  const int cip = cc * 64 + cih * 32 + (lane & 31);
  int ci = -1;
  if (cip < a.Cin) ci = a.perm ? a.perm[cip] : cip;
  if (ci >= 0 && ci < a.Cin_src) {
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int coh = 0; coh < NCOH; ++coh)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int co = ct * 64 + (coh0 + coh) * 32 + acc_row(i, hl);
          if (co < a.Cout)
            unsafeAtomicAdd(a.dw + ((long)co * a.Cin_src + ci) * 27 + kd * 9 + kh * 3 + kw, acc[kw][coh][i]);
        }
  }
}

// ~200 VGPRs: one workgroup (6 waves) per CU.  Forcing 168 VGPRs for two per CU measured 45 % SLOWER (spills in
// the k loop), so the occupancy is left to the register allocator.
template <typename T, bool PIPE>
__global__ __launch_bounds__(wg::NT) void conv3d_k3_wgrad_kernel(wg::Args a) {
  wgrad_body<T, PIPE, 2>(a);
}
// 12 waves, three per SIMD (<= 168 VGPRs): wave = (kh, ci half, co half) with 3 accumulators -- every SIMD carries the
// same MFMA load; the 6-wave form leaves two SIMDs with one wave
template <typename T, bool DMA = false>
__global__ __launch_bounds__(768) void conv3d_k3_wgrad12_kernel(wg::Args a) {
  wgrad_body<T, true, 1, DMA>(a);
}

// dw[co][ci_src][tap] += sum_p part[p][combo][tap][co][ci]; one thread per (combo, tap, co, ci)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int P, int ncc, int ncombo,
                                                           int Cin, int Cin_src, int Cout, const int* __restrict__ perm,
                                                           float* __restrict__ dw) {
  const long per_p = (long)ncombo * 27 * 4096;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < per_p; i += (long)gridDim.x * 256) {
    const int cil = (int)(i & 63), col = (int)((i >> 6) & 63);
    const int tap = (int)((i >> 12) % 27), combo = (int)((i >> 12) / 27);
    const int co = (combo / ncc) * 64 + col, cip = (combo % ncc) * 64 + cil;
    if (co >= Cout || cip >= Cin) continue;
    const int ci = perm ? perm[cip] : cip;
    if (ci < 0 || ci >= Cin_src) continue;
    // The 96^3 layers have 128-256 partitions: sixteen partial sums are requested per round trip (four at a time made this
    // kernel 32-64 dependent round trips per element: 1.1 ms per training step for ~1.5 GB of traffic).  Fixed summation order.
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

static inline int wgrad_partitions(const dua_conv3_desc* d, int* combos_out) {
  using namespace wg;
  const int combos = ((d->Cout + 63) / 64) * ((d->Cin + 63) / 64) * 3;
  const int total = d->N * ((d->D + TD - 1) / TD) * ((d->H + TH - 1) / TH) * ((d->W + TW - 1) / TW);
  // ~3 workgroups per CU over the launch (one resident at a time): measured 1.8x faster than exactly one persistent
  // workgroup per CU on the 96^3 layers (580 vs 1035 us), 2 and 4+ per CU in between
  const int mv = (d->policy & 31) >> 1;        // bits 5, 6 select kernel forms, not a launch shape
  const int mult = mv ? mv : (total < 32 ? 1 : 3);   // tiny levels: fewer partial sums
  int P = (256 * mult + combos - 1) / combos;
  if (P > total) P = total;
  if (P >= 8) P = (P + 7) & ~7;                      // whole groups of 8 (one partition per XCD); fewer: plain order
  if (combos_out) *combos_out = combos;
  return P < 1 ? 1 : P;
}

// dynamic-LDS limits (raised once per device by ensure_prepared(), common.hpp): two (x, dy) tile images; the LDS-DMA form
// of the 12-wave kernel keeps two such buffers
template <typename T> constexpr int wgrad_lds() { return 2 * (wg::XV + wg::TV) * 32 * (int)sizeof(T) + 4 * 64; }
static const LdsAttr kWgradLdsAttrs[] = {
    {(const void*)conv3d_k3_wgrad_kernel<f16, false>, wgrad_lds<f16>()},
    {(const void*)conv3d_k3_wgrad_kernel<f16, true>, wgrad_lds<f16>()},
    {(const void*)conv3d_k3_wgrad_kernel<float, false>, wgrad_lds<float>()},
    {(const void*)conv3d_k3_wgrad_kernel<float, true>, wgrad_lds<float>()},
    {(const void*)conv3d_k3_wgrad12_kernel<f16, false>, wgrad_lds<f16>()},
    {(const void*)conv3d_k3_wgrad12_kernel<f16, true>, 2 * wgrad_lds<f16>()},
};
static const LdsAttrs kWgradLdsReg(kWgradLdsAttrs);

// conv3d_wgrad_fo.hip: the fetch-once form (fp16): one workgroup for all 27 taps of (64 co x 32 ci)
int launch_wgrad_fo(const dua_conv3_desc* d, const void* x, const void* dy, float* dw, int Cin_src, const int* perm, float* ws,
                    long ws_bytes, hipStream_t s);
long wgrad_fo_workspace(const dua_conv3_desc* d);
constexpr int WGRAD_POLICY_THREE_KD = 256;      // dua_conv3_desc.policy bit 8: the first form (a workgroup per kd plane of taps)

template <typename T>
static int launch_wgrad(const dua_conv3_desc* d, const void* x, const void* dy, float* dw, int Cin_src,
                        const int* perm, float* ws, long ws_bytes, hipStream_t s) {
  using namespace wg;
  if (d->policy < 0 || d->policy > 511) return DUA_ERR_ARG;
  if constexpr (sizeof(T) == 2) {
    if (!(d->policy & WGRAD_POLICY_THREE_KD) && ws != nullptr) return launch_wgrad_fo(d, x, dy, dw, Cin_src, perm, ws, ws_bytes, s);
  }
  Args a;
  a.x = x; a.dy = dy; a.dw = dw; a.perm = perm;
  a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W;
  a.Cin = d->Cin; a.Cin_stride = d->Cin_stride; a.Cin_off = d->Cin_off; a.Cin_src = Cin_src;
  a.Cout = d->Cout; a.Cout_stride = d->Cout_stride; a.Cout_off = d->Cout_off;
  a.ncc = (d->Cin + 63) / 64;
  const int nct = (d->Cout + 63) / 64;
  a.tiles_d = (d->D + TD - 1) / TD; a.tiles_h = (d->H + TH - 1) / TH; a.tiles_w = (d->W + TW - 1) / TW;
  a.total_tiles = d->N * a.tiles_d * a.tiles_h * a.tiles_w;
  int combos;
  const int P = wgrad_partitions(d, &combos);
  a.P = P;
  a.ncombo = nct * a.ncc;
#ifdef DUA_ABLATE
  a.abl = g_wgrad_abl;
#else
  a.abl = 0;
#endif
  const int g_wgrad_variant = d->policy & 255;
  a.plain_order = (g_wgrad_variant & 1) || P < 8;
  a.part = (ws && (long)P * combos * 9 * 4096 * 4 <= ws_bytes && P > 1) ? ws : nullptr;
  constexpr int lds = wgrad_lds<T>();
  if (int e = ensure_prepared()) return e;
  const int groups = (P * a.ncombo + 7) / 8;        // groups of 8 (partition, combo) pairs, one per XCD
  const dim3 grid(a.plain_order ? P * a.ncombo * 3 : groups * 24);
  // f16 default: the 12-wave form (-5...-14 % against the 6-wave one on every layer shape, same-process A/B);
  // dua_set_option(4, 64) selects the 6-wave pipelined form, (4, 32) its compiler-scheduled k loop.  f32: 6 waves.
  if (sizeof(T) == 2 && !(g_wgrad_variant & (64 | 32))) {
    if constexpr (sizeof(T) == 2) {
      if (g_wgrad_variant & 128) hipLaunchKernelGGL((conv3d_k3_wgrad12_kernel<T, false>), grid, dim3(768), lds, s, a);   // register prefetch + one LDS buffer
      else hipLaunchKernelGGL((conv3d_k3_wgrad12_kernel<T, true>), grid, dim3(768), 2 * lds, s, a);
    }
  } else
  // f16 default: explicit two-stage k loop (reads of step s+1 before the MFMAs of step s): 3-7 % over hipcc's own
  // schedule on every layer shape in a same-process A/B; dua_set_option(4, 32) selects the plain loop.  f32: plain.
  if (sizeof(T) == 2 && !(g_wgrad_variant & 32)) hipLaunchKernelGGL((conv3d_k3_wgrad_kernel<T, true>), grid, dim3(NT), lds, s, a);
  else hipLaunchKernelGGL((conv3d_k3_wgrad_kernel<T, false>), grid, dim3(NT), lds, s, a);
  if (a.part) {
    const long per_p = (long)nct * a.ncc * 27 * 4096;
    long nb = (per_p + 255) / 256;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(nb > 8192 ? 8192 : nb)), dim3(256), 0, s, a.part, P, a.ncc,
                       nct * a.ncc, d->Cin, Cin_src, d->Cout, perm, dw);
  }
  return (int)hipGetLastError();
}

}  // namespace dua

extern "C" long dua_conv3d_k3_wgrad_workspace(const dua_conv3_desc* d) {
  if (!d) return DUA_ERR_ARG;
  if (d->dtype == DUA_F16 && !(d->policy & dua::WGRAD_POLICY_THREE_KD)) return dua::wgrad_fo_workspace(d);
  int combos;
  const int P = dua::wgrad_partitions(d, &combos);
  return P > 1 ? (long)P * combos * 9 * 4096 * 4 : 0;
}

extern "C" int dua_conv3d_k3_wgrad(const dua_conv3_desc* d, const void* x, const void* dy, float* dw, int Cin_src,
                                   const int* in_perm, void* workspace, long workspace_bytes, void* stream) {
  if (!d || !x || !dy || !dw || Cin_src <= 0) return DUA_ERR_ARG;
  if (d->Cin % 8 || d->Cout % 8 || d->Cin_stride % 8 || d->Cout_stride % 8 || d->Cin_off % 8 || d->Cout_off % 8)
    return DUA_ERR_ARG;
  if (!in_perm && Cin_src > d->Cin) return DUA_ERR_ARG;
  if (d->dtype == DUA_F16) return dua::launch_wgrad<dua::f16>(d, x, dy, dw, Cin_src, in_perm, (float*)workspace, workspace_bytes, (hipStream_t)stream);
  if (d->dtype == DUA_F32) return dua::launch_wgrad<float>(d, x, dy, dw, Cin_src, in_perm, (float*)workspace, workspace_bytes, (hipStream_t)stream);
  return DUA_ERR_ARG;
}
