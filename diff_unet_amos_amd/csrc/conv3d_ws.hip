// conv3d 3x3x3, wave-specialised persistent form (v4) -- the production kernel for layers that fill the chip.
//
// What the ablations of the 4x8x8 kernel showed: its MFMA stream (~200 us for the
// 128->64 @96^3 launch) and its non-MFMA stream (global->LDS staging 160 us, epilogue, per-workgroup
// prologue) are about equally long and overlap badly, because every wave does both.  Here the roles are
// split and the workgroup is persistent:
//   * 512 threads, one workgroup per CU, looping over 8x8x8-voxel x 64-channel output tiles.
//   * waves 0-3 (consumers, one per SIMD) only read operand fragments from LDS and issue MFMAs: each owns two
//     depth slices = 128 voxels x 64 channels = 8 accumulators of 32x32 (128 VGPRs); 6 ds_read_b128 per
//     8 MFMAs (25 % fewer LDS bytes per FLOP than 64x64 wave tiles).
//   * waves 4-7 (producers) move everything: weight slabs (12 KB per (kd,kh), global -> registers two slabs
//     ahead -> LDS double buffer) and the 10x10x10 halo tile of the NEXT Cin chunk / NEXT tile (global ->
//     registers -> InstanceNorm+LeakyReLU+temb transform -> the other halo buffer), so a consumer never
//     waits on a global load, and a tile's prologue overlaps the previous tile's last chunk and epilogue.
//   * one s_barrier per slab.  The consumer places it before the MFMAs of the slab's last k-step, when all
//     its reads of the current slab are already in registers: right after it the next slab is guaranteed
//     complete, so the fragment prefetch runs across the barrier without a bubble.
//   * the weight traffic per FLOP halves against the 256-voxel tile (the L2->LDS bound of v2), halo
//     amplification drops from 2.34x to 1.95x, the InstanceNorm scale/shift preamble runs once per
//     workgroup instead of once per tile.
// LDS: 2 halo buffers x 65,600 B + 2 weight slabs x 12,288 B + 12 B per input channel = 157.3 KB at Cin 128.
// The epilogue staging tile aliases the halo buffer the finished chunk used.  fp16 only (fp32 parity mode
// stays on the v2 kernel), Cin <= 128.
#include "common.hpp"
#include "../../include/dua_hip.h"
#include "conv3_args.hpp"

namespace dua {

namespace c4 {
constexpr int NT = 512, TD = 8, TH = 8, TW = 8, HD = 10, HH = 10, HW = 10, KG = 4, BN = 64;
constexpr int VS = 64, RS = HW * VS + 16, PS = HH * RS;   // 64, 656, 6560
constexpr int HALO = HD * PS;                              // 65600
constexpr int SLAB = 3 * KG * BN * 16;                     // 12288
constexpr int OFF_W = 2 * HALO, OFF_X = OFF_W + 2 * SLAB;  // 131200, 155776
constexpr int CK = 32;                                     // fp16 channels per chunk
constexpr int OS = 32 * 2 + 16;                            // epilogue staging row: 32 channels fp16 + pad
constexpr int STAGE_WAVE = 128 * OS;                       // 10240 B per consumer wave
}  // namespace c4

__device__ __forceinline__ void wg_barrier() { __builtin_amdgcn_s_barrier(); }
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int ABL>
__global__ __launch_bounds__(512, 2) void conv3d_k3_v4_kernel(Conv3Args a) {
  using namespace c4;
  using T = f16;
  using Frag = f16x8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xsc = (float*)(smem + OFF_X);
  float* xsh = xsc + a.nchunks * CK;
  float* xad = xsh + a.nchunks * CK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.y, ct = blockIdx.z;
  const bool fused = a.xf.stats != nullptr;
  const int ntl = ((int)blockIdx.x < a.ntiles) ? (a.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  if (ntl == 0) return;
  const int nphase_chunks = ntl * a.nchunks;     // (tile, chunk) pairs this workgroup walks
  const int Gt = nphase_chunks * 9;              // slabs in its stream

  if (fused) xform_preamble(a.xf, n, a.Cin, xsc, xsh, xad);
  __syncthreads();

  auto tile_origin = [&](int i, int& d0, int& h0, int& w0) {
    const int tile = xcd_remap((int)blockIdx.x + i * (int)gridDim.x, a.ntiles);
    const int tw_ = tile % a.tiles_w, th_ = (tile / a.tiles_w) % a.tiles_h, td_ = tile / (a.tiles_w * a.tiles_h);
    d0 = td_ * TD; h0 = th_ * TH; w0 = tw_ * TW;
  };

  if (wave >= 4) {
    // =========================== producers ===========================
    const int ptid = tid - 256;
    const T* xin = (const T*)a.x + (long)n * a.D * a.H * a.W * a.Cin_stride + a.Cin_off;
    const char* wsrc = (const char*)a.w + (long)ct * a.nchunks * 9 * SLAB + ptid * 16;
    f32x4 wreg[3][3];
    auto load_slab = [&](int g, int set) {
      if (ABL & 2) return;
      const char* src = wsrc + (long)(((g / 9) % a.nchunks) * 9 + g % 9) * SLAB;
#pragma unroll
      for (int j = 0; j < 3; ++j) wreg[set][j] = *(const f32x4*)(src + j * 4096);
    };
    auto store_slab = [&](int g, int set) {
      if (ABL & 2) return;
      char* dst = smem + OFF_W + (g & 1) * SLAB + ptid * 16;
#pragma unroll
      for (int j = 0; j < 3; ++j) *(f32x4*)(dst + j * 4096) = wreg[set][j];
    };
    // Halo staging: a thread owns up to two (row hy, column hx, k-group) "pairs" of the 10x10 face (400 pairs
    // over 256 threads) and walks the 10 depth planes of each: one 64-bit base, constant plane stride, LDS
    // offsets as immediates.  Pair 0 and pair 1 are moved in different phases (10 fragments live at a time).
    int lofs[2], hy_[2], hx_[2], kgp[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int p = ptid + 256 * q, col = p >> 2;
      kgp[q] = p & 3; hy_[q] = col / HW; hx_[q] = col - hy_[q] * HW;
      lofs[q] = hy_[q] * RS + hx_[q] * VS + kgp[q] * 16;
    }
    Frag hv[HD];
    bool hw_ok = false;
    auto load_pair = [&](int pc, int q) {
      if (ABL & 1) return;
      const int i = pc / a.nchunks, ch = pc % a.nchunks;
      int d0, h0, w0;
      tile_origin(i, d0, h0, w0);
      const int gh = h0 + hy_[q] - 1, gw = w0 + hx_[q] - 1;
      hw_ok = (ptid + 256 * q < HH * HW * KG) && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W && (ch * CK + kgp[q] * 8 < a.Cin);
      const long plane = (long)a.H * a.W * a.Cin_stride;
      const T* base = xin + (((long)(d0 - 1) * a.H + gh) * a.W + gw) * a.Cin_stride + ch * CK + kgp[q] * 8;
#pragma unroll
      for (int hd = 0; hd < HD; ++hd) {
        const int gd = d0 - 1 + hd;
        const bool ok = hw_ok && gd >= 0 && gd < a.D;
        const Frag f = *(const Frag*)(ok ? base + hd * plane : xin);    // branch-free: padding reads a valid voxel, then a select
#pragma unroll
        for (int e = 0; e < 8; ++e) hv[hd][e] = ok ? f[e] : (T)0.f;
      }
    };
    auto store_pair = [&](int pc, int q) {
      if (ABL & 1) return;
      if (ptid + 256 * q >= HH * HW * KG) return;
      const int i = pc / a.nchunks, ch = pc % a.nchunks;
      int d0, h0, w0;
      tile_origin(i, d0, h0, w0);
      char* hb = smem + (pc & 1) * HALO + lofs[q];
      const int c0 = ch * CK + kgp[q] * 8;
      float sc[8], sh[8], ad[8];
      const bool xf = fused && hw_ok;
      if (xf) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = xsc[c0 + e]; sh[e] = xsh[c0 + e]; ad[e] = xad[c0 + e]; }
      }
#pragma unroll
      for (int hd = 0; hd < HD; ++hd) {
        const int gd = d0 - 1 + hd;
        Frag f = hv[hd];
        if (xf && gd >= 0 && gd < a.D) f = xform_frag<T>(f, sc, sh, ad, a.xf.slope);   // padding stays literal zero
        *(Frag*)(hb + hd * PS) = f;
      }
    };

    load_slab(0, 0);
    load_slab(1, 1);
    load_pair(0, 0);
    store_slab(0, 0);
    store_pair(0, 0);
    load_pair(0, 1);
    store_pair(0, 1);
    wait_lds();
    wg_barrier();                                  // #0: slab 0 and halo 0 are in LDS
    for (int pc = 0; pc < nphase_chunks; ++pc) {
      const bool has_next = pc + 1 < nphase_chunks;
#pragma unroll
      for (int sl = 0; sl < 9; ++sl) {
        const int g = pc * 9 + sl;
        if (g + 2 < Gt) load_slab(g + 2, ((sl + 2) % 9) % 3);
        if (sl == 0 && has_next) load_pair(pc + 1, 0);
        if (sl == 2 && has_next) store_pair(pc + 1, 0);
        if (sl == 3 && has_next) load_pair(pc + 1, 1);
        if (sl == 5 && has_next) store_pair(pc + 1, 1);
        if (g + 1 < Gt) store_slab(g + 1, ((sl + 1) % 9) % 3);
        wait_lds();
        wg_barrier();                              // #(g+1): slab g+1 (and by sl 5 the next halo) are in LDS
      }
    }
    return;
  }

  // =========================== consumers ===========================
  if (!(ABL & 8)) __builtin_amdgcn_s_setprio(1);
  const int r = lane & 31, hh = lane >> 5;
  const int a_base = (2 * wave) * PS + (r >> 3) * RS + (r & 7) * VS + hh * 16;
  const int b_base = OFF_W + (hh * BN + r) * 16;
  f32x16 acc[2][2][2];     // [depth slice][h half][cout half]
  Frag fa[2][4], fb[2][2]; // double-buffered fragments: A = (ds, m), B = q
  auto zero_acc = [&]() {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[s][m][q][i] = 0.f;
  };
  // fragments of k-step t (kw = t>>1, ks = t&1) from the slab bases (one VGPR each; the rest are immediates)
  auto ld = [&](const char* ap, const char* wb, int t, int b) {
    const int kw = t >> 1, ks = t & 1;
    const int ao = kw * VS + ks * 32, bo = (kw * KG + 2 * ks) * BN * 16;
    fa[b][0] = *(const Frag*)(ap + ao);
    fb[b][0] = *(const Frag*)(wb + bo);
    fb[b][1] = *(const Frag*)(wb + bo + 32 * 16);
    fa[b][1] = *(const Frag*)(ap + ao + 4 * RS);
    fa[b][2] = *(const Frag*)(ap + ao + PS);
    fa[b][3] = *(const Frag*)(ap + ao + PS + 4 * RS);
  };
  auto a_ptr = [&](int pc, int sl) { return (const char*)smem + (pc & 1) * HALO + a_base + (sl / 3) * PS + (sl % 3) * RS; };
  auto b_ptr = [&](int g) { return (const char*)smem + b_base + (g & 1) * SLAB; };
  auto mm = [&](int b) {
    if (ABL & 4) { asm volatile("" ::"v"(fa[b][0]), "v"(fa[b][1]), "v"(fa[b][2]), "v"(fa[b][3]), "v"(fb[b][0]), "v"(fb[b][1])); return; }
    mma32(acc[0][0][0], fa[b][0], fb[b][0]);
    mma32(acc[0][0][1], fa[b][0], fb[b][1]);
    mma32(acc[0][1][0], fa[b][1], fb[b][0]);
    mma32(acc[0][1][1], fa[b][1], fb[b][1]);
    mma32(acc[1][0][0], fa[b][2], fb[b][0]);
    mma32(acc[1][0][1], fa[b][2], fb[b][1]);
    mma32(acc[1][1][0], fa[b][3], fb[b][0]);
    mma32(acc[1][1][1], fa[b][3], fb[b][1]);
  };

  zero_acc();
  unsigned long long t_bar = 0, t_epi = 0, t_start = 0;
  if (ABL & 16) t_start = __builtin_amdgcn_s_memtime();
  wg_barrier();                                    // #0
  const char* ap = a_ptr(0, 0);
  const char* wb = b_ptr(0);
  ld(ap, wb, 0, 0);
  for (int pc = 0; pc < nphase_chunks; ++pc) {
#pragma unroll
    for (int sl = 0; sl < 9; ++sl) {
      const int g = pc * 9 + sl;
      // steps 0..4: prefetch t+1, multiply t.  6 steps per slab => buffer of step t is t & 1.
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        ld(ap, wb, t + 1, (t + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        mm(t & 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      // step 5: its fragments are the last reads of this slab.  Once they have landed, meet the producers:
      // slab g+1 is complete after this barrier, and they may start overwriting slab g's buffer.
      unsigned long long tb0 = 0;
      if (ABL & 16) tb0 = __builtin_amdgcn_s_memtime();
      wait_lds();
      wg_barrier();                                // #(g+1)
      if (ABL & 16) t_bar += __builtin_amdgcn_s_memtime() - tb0;
      if (g + 1 < Gt) {
        ap = sl < 8 ? a_ptr(pc, sl + 1) : a_ptr(pc + 1, 0);
        wb = b_ptr(g + 1);
        ld(ap, wb, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      mm(1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if ((pc + 1) % a.nchunks != 0) continue;

    // ---------------- tile epilogue (this wave's 128 voxels x 64 channels) ----------------
    unsigned long long te0 = 0;
    if (ABL & 16) te0 = __builtin_amdgcn_s_memtime();
    const int i = pc / a.nchunks;
    int d0, h0, w0;
    tile_origin(i, d0, h0, w0);
    char* ot = smem + (pc & 1) * HALO + wave * STAGE_WAVE;     // every halo read of this chunk happened before the last barrier
    T* yout = (T*)a.y + (long)n * a.D * a.H * a.W * a.Cout_stride + a.Cout_off + ct * BN;
    double Sd[2] = {0, 0}, Qd[2] = {0, 0};                     // per lane: channel q*32 + r (lanes with hh == 0 publish)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int co = q * 32 + r;
      const float bq = a.bias[ct * BN + co];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int gd = d0 + 2 * wave + s;
        const bool dok = gd < a.D;
        const float cnt = dok ? (float)(min(TH, a.H - h0) * min(TW, a.W - w0)) : 0.f;
        float sum = 0.f, vals[2][16];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int hl = 4 * m + (k >> 2), wl = (k & 3) + 4 * hh;
            const bool ok = dok && (h0 + hl < a.H) && (w0 + wl < a.W);
            const T tv = (T)(acc[s][m][q][k] + bq);
            const float fv = ok ? (float)tv : 0.f;
            vals[m][k] = fv;
            sum += fv;
            *(T*)(ot + (s * 64 + m * 32 + acc_row(k, hh)) * OS + r * 2) = tv;
          }
        sum += __shfl_xor(sum, 32);
        const float mean = cnt > 0.f ? sum / cnt : 0.f;
        float m2 = 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int hl = 4 * m + (k >> 2), wl = (k & 3) + 4 * hh;
            const bool ok = dok && (h0 + hl < a.H) && (w0 + wl < a.W);
            const float dl = vals[m][k] - mean;
            m2 += ok ? dl * dl : 0.f;
          }
        m2 += __shfl_xor(m2, 32);
        if (cnt > 0.f) { Sd[q] += (double)sum; Qd[q] += (double)m2 + (double)sum * (double)sum / (double)cnt; }
      }
      __builtin_amdgcn_wave_barrier();
      // read this wave's 128 x 32 staging tile back as 16-byte rows and store: 4 lanes cover a voxel's 64 B
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int v = it * 16 + (lane >> 2), cg = lane & 3;     // v: 0..127 = slice*64 + h*8 + w
        const int s = v >> 6, gh = h0 + ((v >> 3) & 7), gw = w0 + (v & 7), gd = d0 + 2 * wave + s;
        if (gd < a.D && gh < a.H && gw < a.W && ct * BN + q * 32 + cg * 8 < a.Cout)
          *(Frag*)(yout + (((long)gd * a.H + gh) * a.W + gw) * a.Cout_stride + q * 32 + cg * 8) =
              *(const Frag*)(ot + v * OS + cg * 16);
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (hh == 0) {
      const int rep = (blockIdx.x + wave) & (STAT_REPLICAS - 1);
#pragma unroll
      for (int q = 0; q < 2; ++q)
        if (ct * BN + q * 32 + r < a.Cout) stats_add(a.stats, n, a.cout_pad, rep, ct * BN + q * 32 + r, Sd[q], Qd[q]);
    }
    zero_acc();
    if (ABL & 16) t_epi += __builtin_amdgcn_s_memtime() - te0;
  }
  if ((ABL & 16) && lane == 0 && a.part != nullptr) {
    unsigned long long* o = (unsigned long long*)a.part + ((long)blockIdx.x * 4 + wave) * 4;
    o[0] = __builtin_amdgcn_s_memtime() - t_start; o[1] = t_bar; o[2] = t_epi; o[3] = (unsigned long long)Gt;
  }
}

int launch_conv3_v4(Conv3Args& a, int N, int nct, int xf_bytes, hipStream_t s) {
  using namespace c4;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return DUA_ERR_ARG;
    cus = p.multiProcessorCount;
    hipError_t e = hipFuncSetAttribute((const void*)conv3d_k3_v4_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, OFF_X + 3 * 4 * 128);
    if (e != hipSuccess) { cus = 0; return (int)e; }
#ifdef DUA_ABLATE
    hipFuncSetAttribute((const void*)conv3d_k3_v4_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, OFF_X + 3 * 4 * 128);
    hipFuncSetAttribute((const void*)conv3d_k3_v4_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, OFF_X + 3 * 4 * 128);
    hipFuncSetAttribute((const void*)conv3d_k3_v4_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, OFF_X + 3 * 4 * 128);
    hipFuncSetAttribute((const void*)conv3d_k3_v4_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, OFF_X + 3 * 4 * 128);
    hipFuncSetAttribute((const void*)conv3d_k3_v4_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, OFF_X + 3 * 4 * 128);
    hipFuncSetAttribute((const void*)conv3d_k3_v4_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, OFF_X + 3 * 4 * 128);
    hipFuncSetAttribute((const void*)conv3d_k3_v4_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, OFF_X + 3 * 4 * 128);
    hipFuncSetAttribute((const void*)conv3d_k3_v4_kernel<19>, hipFuncAttributeMaxDynamicSharedMemorySize, OFF_X + 3 * 4 * 128);
#endif
  }
  const int td8 = (a.D + TD - 1) / TD;
  a.ntiles = td8 * a.tiles_h * a.tiles_w;
  const int gx = a.ntiles < cus ? a.ntiles : cus;
  extern int g_conv_variant;
  const dim3 grid(gx, N, nct);
  switch (g_conv_variant) {
#ifdef DUA_ABLATE
    case 201: hipLaunchKernelGGL(conv3d_k3_v4_kernel<1>, grid, dim3(NT), OFF_X + xf_bytes, s, a); break;
    case 202: hipLaunchKernelGGL(conv3d_k3_v4_kernel<2>, grid, dim3(NT), OFF_X + xf_bytes, s, a); break;
    case 203: hipLaunchKernelGGL(conv3d_k3_v4_kernel<3>, grid, dim3(NT), OFF_X + xf_bytes, s, a); break;
    case 204: hipLaunchKernelGGL(conv3d_k3_v4_kernel<4>, grid, dim3(NT), OFF_X + xf_bytes, s, a); break;
    case 208: hipLaunchKernelGGL(conv3d_k3_v4_kernel<8>, grid, dim3(NT), OFF_X + xf_bytes, s, a); break;
    case 207: hipLaunchKernelGGL(conv3d_k3_v4_kernel<7>, grid, dim3(NT), OFF_X + xf_bytes, s, a); break;
    case 210: hipLaunchKernelGGL(conv3d_k3_v4_kernel<16>, grid, dim3(NT), OFF_X + xf_bytes, s, a); break;
    case 211: hipLaunchKernelGGL(conv3d_k3_v4_kernel<19>, grid, dim3(NT), OFF_X + xf_bytes, s, a); break;
#endif
    default: hipLaunchKernelGGL(conv3d_k3_v4_kernel<0>, grid, dim3(NT), OFF_X + xf_bytes, s, a);
  }
  return (int)hipGetLastError();
}

}  // namespace dua
