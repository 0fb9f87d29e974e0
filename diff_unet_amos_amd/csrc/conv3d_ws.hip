// conv3d 3x3x3, wave-specialised persistent form (variant 4) -- for fp16 layers that fill the chip.
//
// What the stamps and ablations of the 4x8x8 kernel showed: its MFMA stream and its
// non-MFMA stream (global->LDS staging, per-workgroup prologue, epilogue) are about equally long and overlap
// badly because every wave does both; and a first wave-specialised attempt with ONE 128x64 MFMA wave per SIMD
// lost more on the consumer side (a lone wave issues a 48-MFMA slab in ~2,000 cycles instead of 1,536, and its
// VALU-heavy epilogue runs at 4 cycles per instruction) than it won on staging.  This form keeps TWO MFMA
// waves per SIMD:
//   * 768 threads, one persistent workgroup per CU, looping over 8x8x8-voxel x 64-channel output tiles.
//   * waves 0-7 (consumers, two per SIMD): wave w owns depth slice w = 64 voxels x 64 channels (2x2
//     accumulators of 32x32), reads operand fragments from LDS one k-step ahead and issues MFMAs; nothing else.
//   * waves 8-11 (producers, one per SIMD) move everything: weight slabs (12 KB per (kd,kh), global -> registers
//     two slabs ahead -> LDS double buffer) and the 10x10x10 halo tile of the NEXT Cin chunk / NEXT tile
//     (global -> registers -> InstanceNorm+LeakyReLU+temb transform -> the other halo buffer).  A consumer never
//     waits on a global load, a tile's prologue overlaps the previous tile's last chunk and epilogue, and the
//     scale/shift preamble runs once per workgroup instead of once per tile.
//   * one s_barrier per slab.  A consumer places it before the MFMAs of the slab's last k-step, when all its
//     reads of the current slab are already in registers: right after it the next slab is guaranteed
//     complete, so the fragment prefetch runs across the barrier without a bubble.
//   * InstanceNorm sums stay in fp64 registers across all tiles of the workgroup and are published once
//     (7x fewer atomics than one publication per tile).
//   * weight traffic per FLOP halves against the 256-voxel tile, halo amplification 2.34x -> 1.95x.
// LDS: 2 halo buffers x 65,600 B + 2 weight slabs x 12,288 B + 12 B per input channel = 157.3 KB at Cin 128; the
// epilogue staging tile aliases the halo buffer the finished chunk used.  fp16 only (fp32 parity mode stays on
// the v2 kernel), Cin <= 128.  12 waves per CU => 170 VGPRs per wave.
#include "common.hpp"
#include "../../include/dua_hip.h"
#include "conv3_args.hpp"

namespace dua {

namespace c4 {
constexpr int NT = 768, NCONS = 512, TD = 8, TH = 8, TW = 8, HD = 10, HH = 10, HW = 10, KG = 4, BN = 64;
constexpr int VS = 64, RS = HW * VS + 16, PS = HH * RS;   // 64, 656, 6560
constexpr int HALO = HD * PS;                              // 65600
constexpr int SLAB = 3 * KG * BN * 16;                     // 12288
constexpr int OFF_W = 2 * HALO, OFF_X = OFF_W + 2 * SLAB;  // 131200, 155776
constexpr int OFF_F = OFF_X + 3 * 4 * 128;                 // 157312: hand-off counters ready[2], done[2]
constexpr int LDS_TOTAL = OFF_F + 64;
constexpr int CK = 32;                                     // fp16 channels per chunk
constexpr int OS = 32 * 2 + 16;                            // epilogue staging row: 32 channels fp16 + pad
constexpr int STAGE_WAVE = 64 * OS;                        // 5120 B per consumer wave
}  // namespace c4

__device__ __forceinline__ void wg_barrier() { __builtin_amdgcn_s_barrier(); }
// Producer-side loads are inline asm so that THIS code, not hipcc, decides where the wave waits for them: the
// compiler's own placement put s_waitcnt vmcnt(0) after every load of the halo walk and at every loop edge.
// Loads complete in issue order; vm_wait<N>(regs...) returns when all but the N youngest are done and ties the
// registers to the wait so that no use is scheduled above it.
template <typename V>
__device__ __forceinline__ void vm_load16(V& dst, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
template <int N, typename V>
__device__ __forceinline__ void vm_wait3(V& r0, V& r1, V& r2) {
  asm volatile("s_waitcnt vmcnt(%3)" : "+v"(r0), "+v"(r1), "+v"(r2) : "n"(N) : "memory");
}
template <int N, typename V>
__device__ __forceinline__ void vm_wait10(V (&r)[10]) {
  asm volatile("s_waitcnt vmcnt(%10)"
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9])
               : "n"(N) : "memory");
}
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// Workgroup-local hand-offs through LDS counters (no s_barrier in the steady state: a barrier drains the MFMA pipe of
// every SIMD at once, ~430 cycles per slab here).  A wave publishes after its own LDS traffic has completed; a
// waiter spins on one ds_read with s_sleep.  Counters only grow, so "seen >= target" can never be missed.
__device__ __forceinline__ void lds_signal(unsigned* cnt) {
  wait_lds();
  if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_wait_ge(unsigned* cnt, unsigned target) {
  for (int spin = 0; spin < (1 << 22); ++spin) {     // bounded: a protocol bug must not hang the GPU (results would be wrong instead)
    const unsigned v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    if ((int)(v - target) >= 0) break;
    __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
}

template <int ABL, int SYNC = 0>     // SYNC 0: one s_barrier per slab (default); 1: LDS counters (measured 8 % slower)
__global__ __launch_bounds__(768, 3) void conv3d_k3_v4_kernel(Conv3Args a) {
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

  unsigned* f_ready = (unsigned*)(smem + OFF_F);     // [2]: +1 per producer wave per slab written into that weight buffer
  unsigned* f_done = f_ready + 2;                     // [2]: +1 per consumer wave per slab fully read from that buffer
  if (tid < 4) f_ready[tid] = 0;
  if (fused) xform_preamble(a.xf, n, a.Cin, xsc, xsh, xad);
  __syncthreads();

  auto tile_origin = [&](int i, int& d0, int& h0, int& w0) {
    const int tile = xcd_remap((int)blockIdx.x + i * (int)gridDim.x, a.ntiles);
    const int tw_ = tile % a.tiles_w, th_ = (tile / a.tiles_w) % a.tiles_h, td_ = tile / (a.tiles_w * a.tiles_h);
    d0 = td_ * TD; h0 = th_ * TH; w0 = tw_ * TW;
  };

  if (wave >= 8) {
    // =========================== producers ===========================
    const int ptid = tid - NCONS;
    const T* xin = (const T*)a.x + (long)n * a.D * a.H * a.W * a.Cin_stride + a.Cin_off;
    const char* wsrc = (const char*)a.w + (long)ct * a.nchunks * 9 * SLAB + ptid * 16;
    f32x4 wreg[3][3];
    auto load_slab = [&](int g, int set) {
      if (ABL & 2) return;
      const char* src = wsrc + (long)(((g / 9) % a.nchunks) * 9 + g % 9) * SLAB;
#pragma unroll
      for (int j = 0; j < 3; ++j) vm_load16(wreg[set][j], src + j * 4096);
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
      const int p = min(ptid + 256 * q, HH * HW * KG - 1), col = p >> 2;   // threads past the last pair duplicate it
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
      hw_ok = gh >= 0 && gh < a.H && gw >= 0 && gw < a.W && (ch * CK + kgp[q] * 8 < a.Cin);
      const long plane = (long)a.H * a.W * a.Cin_stride;
      const T* base = xin + (((long)(d0 - 1) * a.H + gh) * a.W + gw) * a.Cin_stride + ch * CK + kgp[q] * 8;
#pragma unroll
      for (int hd = 0; hd < HD; ++hd) {
        const int gd = d0 - 1 + hd;
        const bool ok = hw_ok && gd >= 0 && gd < a.D;
        // no use of the value here: ten loads go out back to back; out-of-volume items read a valid voxel and are
        // zeroed when they are stored (a select next to the load makes hipcc wait for each load in turn)
        vm_load16(hv[hd], ok ? base + hd * plane : xin);
      }
    };
    auto store_pair = [&](int pbuf, int pc, int q) {      // pbuf: halo buffer parity; pc: (tile, chunk) the data belongs to
      if (ABL & 1) return;
      const int i = pc / a.nchunks, ch = pc % a.nchunks;
      int d0, h0, w0;
      tile_origin(i, d0, h0, w0);
      char* hb = smem + (pbuf & 1) * HALO + lofs[q];
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
        const bool ok = hw_ok && gd >= 0 && gd < a.D;
        Frag f = hv[hd];
        if (xf) f = xform_frag<T>(f, sc, sh, ad, a.xf.slope);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = ok ? f[e] : (T)0.f;                          // padding stays literal zero
        *(Frag*)(hb + hd * PS) = f;
      }
    };

    load_slab(0, 0);
    load_slab(min(1, Gt - 1), 1);
    load_slab(min(2, Gt - 1), 2);
    load_pair(0, 0);
    vm_wait10<0>(hv);                              // prologue: simply wait for everything
    vm_wait3<0>(wreg[0][0], wreg[0][1], wreg[0][2]);
    vm_wait3<0>(wreg[1][0], wreg[1][1], wreg[1][2]);
    vm_wait3<0>(wreg[2][0], wreg[2][1], wreg[2][2]);
    store_slab(0, 0);
    store_pair(0, 0, 0);
    load_pair(0, 1);
    vm_wait10<0>(hv);
    store_pair(0, 0, 1);
    if (SYNC) lds_signal(&f_ready[0]);
    else { wait_lds(); wg_barrier(); }             // #0: slab 0 and halo 0 are in LDS
    // Steady state, one phase per slab g (sl = g % 9), every phase the same straight-line code:
    //   issue the 3 loads of slab g+3; (sl 0 / 3: issue the 10 loads of halo pair 0 / 1 of the next chunk);
    //   wait for slab g+1 (issued two phases ago) and write it to the other weight buffer;
    //   (sl 2 / 5: wait for the halo pair issued two phases ago, transform, write to the other halo buffer).
    // Loads return in order, so the waits are counted: behind slab g+1's loads there are the 6 slab loads of the
    // last two phases plus 10 halo loads when a pair was issued in phases sl-2..sl.  Past the end of the stream the
    // clamped indices re-load the last slab / halo into buffers nobody reads any more (no branches in the loop).
    unsigned long long p_bar = 0, p_start = 0, p_halo = 0;
    if (ABL & 16) p_start = __builtin_amdgcn_s_memtime();
    for (int pc = 0; pc < nphase_chunks; ++pc) {
      const int pcn = min(pc + 1, nphase_chunks - 1);
#pragma unroll
      for (int sl = 0; sl < 9; ++sl) {
        const int g = pc * 9 + sl;
        load_slab(min(g + 3, Gt - 1), sl % 3);              // (sl + 3) % 9 % 3 == sl % 3
        if (sl == 0) load_pair(pcn, 0);
        if (sl == 3) load_pair(pcn, 1);
        constexpr int S1 = 0;  (void)S1;
        // the buffer slab g+1 goes into was last read as slab g-1: all eight consumer waves must be done with it
        if (SYNC && g >= 1) lds_wait_ge(&f_done[(g + 1) & 1], 8u * (unsigned)((g - 1) / 2 + 1));
        if (!(ABL & 2)) {
          if (sl <= 5) vm_wait3<16>(wreg[(sl + 1) % 3][0], wreg[(sl + 1) % 3][1], wreg[(sl + 1) % 3][2]);
          else         vm_wait3<6>(wreg[(sl + 1) % 3][0], wreg[(sl + 1) % 3][1], wreg[(sl + 1) % 3][2]);
        }
        store_slab(g + 1, (sl + 1) % 3);
        if (sl == 2 || sl == 5) {
          if (!(ABL & 1)) vm_wait10<6>(hv);
          store_pair(pc + 1, pcn, sl == 2 ? 0 : 1);
        }
        if (SYNC) lds_signal(&f_ready[(g + 1) & 1]);  // slab g+1 (and by sl 5 the next halo) are in LDS
        else {
          wait_lds();
          unsigned long long tb = 0;
          if (ABL & 16) tb = __builtin_amdgcn_s_memtime();
          wg_barrier();
          if (ABL & 16) p_bar += __builtin_amdgcn_s_memtime() - tb;
        }
      }
    }
    if ((ABL & 16) && lane == 0 && a.part != nullptr) {
      unsigned long long* o = (unsigned long long*)a.part + ((long)blockIdx.x * 12 + wave) * 4;
      o[0] = __builtin_amdgcn_s_memtime() - p_start; o[1] = p_bar; o[2] = p_halo; o[3] = (unsigned long long)Gt;
    }
    return;
  }

  // =========================== consumers ===========================
  if (!(ABL & 8)) __builtin_amdgcn_s_setprio(1);
  const int r = lane & 31, hh = lane >> 5;
  const int a_base = wave * PS + (r >> 3) * RS + (r & 7) * VS + hh * 16;
  const int b_base = OFF_W + (hh * BN + r) * 16;
  f32x16 acc[2][2];        // [h half][cout half]
  Frag fa[2][2], fb[2][2]; // double-buffered fragments
  float bias2[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) bias2[q] = a.bias[ct * BN + q * 32 + r];
  auto init_acc = [&]() {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][q][i] = bias2[q];
  };
  // fragments of k-step t (kw = t>>1, ks = t&1) from the slab bases (one VGPR each; the rest are immediates)
  auto ld = [&](const char* ap, const char* wb, int t, int b) {
    const int kw = t >> 1, ks = t & 1;
    const int ao = kw * VS + ks * 32, bo = (kw * KG + 2 * ks) * BN * 16;
    fa[b][0] = *(const Frag*)(ap + ao);
    fb[b][0] = *(const Frag*)(wb + bo);
    fb[b][1] = *(const Frag*)(wb + bo + 32 * 16);
    fa[b][1] = *(const Frag*)(ap + ao + 4 * RS);
  };
  auto a_ptr = [&](int pc, int sl) { return (const char*)smem + (pc & 1) * HALO + a_base + (sl / 3) * PS + (sl % 3) * RS; };
  auto b_ptr = [&](int g) { return (const char*)smem + b_base + (g & 1) * SLAB; };
  auto mm = [&](int b) {
    if (ABL & 4) { asm volatile("" ::"v"(fa[b][0]), "v"(fa[b][1]), "v"(fb[b][0]), "v"(fb[b][1])); return; }
    mma32(acc[0][0], fa[b][0], fb[b][0]);
    mma32(acc[0][1], fa[b][0], fb[b][1]);
    mma32(acc[1][0], fa[b][1], fb[b][0]);
    mma32(acc[1][1], fa[b][1], fb[b][1]);
  };

  init_acc();
  double Sd[2] = {0, 0}, Qd[2] = {0, 0};           // InstanceNorm sums of channel q*32 + r, kept across tiles
  unsigned long long t_bar = 0, t_epi = 0, t_start = 0;
  if (ABL & 16) t_start = __builtin_amdgcn_s_memtime();
  if (SYNC) lds_wait_ge(&f_ready[0], 4u);
  else wg_barrier();                               // #0
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
      if (SYNC) {
        lds_signal(&f_done[g & 1]);                                         // this wave is done with slab g
        if (g + 1 < Gt) lds_wait_ge(&f_ready[(g + 1) & 1], 4u * (unsigned)((g + 1) / 2 + 1));   // slab g+1 is in LDS
      } else {
        wait_lds();
        wg_barrier();                              // #(g+1)
      }
      if (ABL & 16) t_bar += __builtin_amdgcn_s_memtime() - tb0;
      ap = sl < 8 ? a_ptr(pc, sl + 1) : a_ptr(pc + 1, 0);      // past the end: a harmless read of stale LDS
      wb = b_ptr(g + 1);
      ld(ap, wb, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      mm(1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if ((pc + 1) % a.nchunks != 0) continue;

    // ---------------- tile epilogue (this wave's 64 voxels x 64 channels) ----------------
    unsigned long long te0 = 0;
    if (ABL & 16) te0 = __builtin_amdgcn_s_memtime();
    const int i = pc / a.nchunks;
    int d0, h0, w0;
    tile_origin(i, d0, h0, w0);
    // the staging tile aliases this chunk's halo buffer: every consumer wave must have finished its last slab
    if (SYNC) lds_wait_ge(&f_done[(pc * 9 + 8) & 1], 8u * (unsigned)((pc * 9 + 8) / 2 + 1));
    char* ot = smem + (pc & 1) * HALO + wave * STAGE_WAVE;
    T* yout = (T*)a.y + (long)n * a.D * a.H * a.W * a.Cout_stride + a.Cout_off + ct * BN;
    const int gd = d0 + wave;
    const bool dok = gd < a.D;
    const bool full = d0 + TD <= a.D && h0 + TH <= a.H && w0 + TW <= a.W;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const int hl = 4 * m + (k >> 2), wl = (k & 3) + 4 * hh;
          const bool ok = full || (dok && (h0 + hl < a.H) && (w0 + wl < a.W));
          const float v = ok ? acc[m][q][k] : 0.f;
          s1 += v;
          s2 = fmaf(v, v, s2);
          *(T*)(ot + (m * 32 + acc_row(k, hh)) * OS + r * 2) = (T)v;
        }
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      Sd[q] += (double)s1; Qd[q] += (double)s2;
      __builtin_amdgcn_wave_barrier();
      // read this wave's 64 x 32 staging tile back as 16-byte rows and store: 4 lanes cover a voxel's 64 B
      if (dok) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int v = it * 16 + (lane >> 2), cg = lane & 3;     // v: 0..63 = h*8 + w
          const int gh = h0 + (v >> 3), gw = w0 + (v & 7);
          if ((full || (gh < a.H && gw < a.W)) && ct * BN + q * 32 + cg * 8 < a.Cout)
            *(Frag*)(yout + (((long)gd * a.H + gh) * a.W + gw) * a.Cout_stride + q * 32 + cg * 8) =
                *(const Frag*)(ot + v * OS + cg * 16);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    init_acc();
    if (ABL & 16) t_epi += __builtin_amdgcn_s_memtime() - te0;
  }
  if (hh == 0) {
    const int rep = (blockIdx.x + wave) & (STAT_REPLICAS - 1);
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (ct * BN + q * 32 + r < a.Cout) stats_add(a.stats, n, a.cout_pad, rep, ct * BN + q * 32 + r, Sd[q], Qd[q]);
  }
  if ((ABL & 16) && lane == 0 && a.part != nullptr) {
    unsigned long long* o = (unsigned long long*)a.part + ((long)blockIdx.x * 12 + wave) * 4;
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
    hipError_t e = hipFuncSetAttribute((const void*)conv3d_k3_v4_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    if (e != hipSuccess) { cus = 0; return (int)e; }
#ifdef DUA_ABLATE
    hipFuncSetAttribute((const void*)conv3d_k3_v4_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    hipFuncSetAttribute((const void*)conv3d_k3_v4_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    hipFuncSetAttribute((const void*)conv3d_k3_v4_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    hipFuncSetAttribute((const void*)conv3d_k3_v4_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    hipFuncSetAttribute((const void*)conv3d_k3_v4_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    hipFuncSetAttribute((const void*)conv3d_k3_v4_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    hipFuncSetAttribute((const void*)conv3d_k3_v4_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    hipFuncSetAttribute((const void*)conv3d_k3_v4_kernel<19>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
#endif
  }
  const int td8 = (a.D + TD - 1) / TD;
  a.ntiles = td8 * a.tiles_h * a.tiles_w;
  extern int g_conv_variant;
  extern int g_v4_grid_quarters;                    // dua_set_option(5, q): workgroups = q/4 per CU (default 4 = one per CU)
  const int want = cus * g_v4_grid_quarters / 4;
  const int gx = a.ntiles < want ? a.ntiles : want;
  const dim3 grid(gx, N, nct);
  switch (g_conv_variant) {
#ifdef DUA_ABLATE
    case 201: hipLaunchKernelGGL(conv3d_k3_v4_kernel<1>, grid, dim3(NT), LDS_TOTAL, s, a); break;
    case 202: hipLaunchKernelGGL(conv3d_k3_v4_kernel<2>, grid, dim3(NT), LDS_TOTAL, s, a); break;
    case 203: hipLaunchKernelGGL(conv3d_k3_v4_kernel<3>, grid, dim3(NT), LDS_TOTAL, s, a); break;
    case 204: hipLaunchKernelGGL(conv3d_k3_v4_kernel<4>, grid, dim3(NT), LDS_TOTAL, s, a); break;
    case 208: hipLaunchKernelGGL(conv3d_k3_v4_kernel<8>, grid, dim3(NT), LDS_TOTAL, s, a); break;
    case 207: hipLaunchKernelGGL(conv3d_k3_v4_kernel<7>, grid, dim3(NT), LDS_TOTAL, s, a); break;
    case 210: hipLaunchKernelGGL(conv3d_k3_v4_kernel<16>, grid, dim3(NT), LDS_TOTAL, s, a); break;
    case 211: hipLaunchKernelGGL(conv3d_k3_v4_kernel<19>, grid, dim3(NT), LDS_TOTAL, s, a); break;
#endif
    default: hipLaunchKernelGGL(conv3d_k3_v4_kernel<0>, grid, dim3(NT), LDS_TOTAL, s, a);
  }
  return (int)hipGetLastError();
}

}  // namespace dua
