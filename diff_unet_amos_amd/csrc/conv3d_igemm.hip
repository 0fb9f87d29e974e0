// 3x3x3 (pad 1, stride 1) 3-D convolution as implicit GEMM on the gfx950 matrix cores.
//
// Replaces every `Conv3d k3 p1` the reference reaches through MONAI's Convolution block
// (models/basic_unet/denoiser.py:56-59, models/basic_unet/pretrained/basic_unet.py:60-63) and
// fuses around it:
//   prologue  : InstanceNorm3d(affine) + LeakyReLU(0.1) (+ timestep-embedding bias) of the
//               PRODUCER layer, applied while the halo tile is staged (denoiser.py:63-67);
//               out-of-volume halo voxels stay literal zeros (padding follows the activation)
//   epilogue  : + bias, per-(n, c) InstanceNorm statistics of THIS layer's output (sum x and sum x^2 from the
//               fp32 accumulators, combined per workgroup, then two fp64 atomics per channel into one of 8
//               replica rows), coalesced 16-byte stores of the raw output.
//
// GEMM view: M = output voxels, N = Cout, K = 27 taps x Cin.
// Workgroup (256 threads = 4 waves, two workgroups per CU): a 4x8x8 output tile x 64 output channels; wave w
// owns depth slice w (64 voxels) as 4 x 4 accumulator blocks of MFMA 16x16 (v_mfma_f32_16x16x32_f16: one
// instruction consumes a whole 64-byte Cin chunk of a tap; fp32 parity mode: 4 x v_mfma_f32_16x16x4_f32).
// The 16x16 shape is a measured choice (tools/ubench/mfma_lds_ubench.hip): on random
// data the chip holds a higher clock on it than on 32x32x16, +10..19 % FLOP/s for the same LDS traffic.
// Per Cin chunk of 64 bytes/voxel the 6x10x10 halo tile is staged once in LDS (registers in between: the
// producer's normalisation is applied there); the 9 taps of one kd plane of the packed weights follow it slab
// by slab, copied global -> LDS by LDS-DMA (global_load_lds, no registers) one slab ahead of the MFMAs.
// A and B fragments are 16-byte ds_read_b128; the halo row stride (672 B: stride / 16 == 2 mod 4) and the
// 2x8 row -> voxel map make every fragment read bank-conflict free under the ds_read_b128 lane groups.
#include <type_traits>
#include "common.hpp"
#include "../../include/dua_hip.h"
#include "conv3_args.hpp"

namespace dua {

namespace c3 {
constexpr int TH = 8, TW = 8;
constexpr int HH = TH + 2, HW = TW + 2;
constexpr int KG = 4;                      // k-groups (16 B) per chunk
constexpr int VS = KG * 16;                // 64 B per halo voxel per chunk
constexpr int RS = HW * VS + 32;           // 672: halo row stride, padded (bank-conflict free for the 16-row A blocks)
constexpr int PS = HH * RS;                // 6720: halo plane stride
constexpr int BN = 64;                     // output channels per workgroup
constexpr int SLAB = 3 * KG * BN * 16;     // 12288: packed weights of one (kd, kh): [kw][k-group][64 couts][16 B]
constexpr int XF_MAX = 3 * 4 * 1024;       // scale/shift/add tables of the fused input transform (Cin <= 1024)
constexpr int NSLAB = 3;                   // weight slab ring: the DMA runs two slabs ahead of the MFMAs
constexpr int lds_main(int td) { return (td + 2) * PS + NSLAB * SLAB; }
}  // namespace c3

#ifdef DUA_ABLATE      // diagnostic builds only (outputs are wrong): -DDUA_ABLATE=<mask>, 1 = no MFMA, 2 = no fragment
constexpr int ABL = DUA_ABLATE;   // reads, 4 = no weight staging, 8 = no halo staging, 16 = no epilogue
#else
constexpr int ABL = 0;
#endif

// 0 = auto (split-K for small layers, 2x8x8 tiles for mid-size ones); 2 forces 4x8x8 tiles without split-K, 3 forces 2x8x8
int g_conv_variant = 0;
int g_skip_splitk_finish = 0;   // diagnostics (dua_set_option(2, 1)): time the split-K main kernel alone; outputs are not finished
extern int g_wgrad_abl;
extern int g_wgrad_variant;

__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// One K = 32-channel (fp16) / 16-channel (fp32) step of a 16x16 output block.  Lane l = (row/col c = l & 15,
// k-group kq = l >> 4) holds the 16 bytes of its row (A: voxel) or column (B: output channel) for k-group kq.
__device__ __forceinline__ void mma16(f32x4& acc, const f16x8& a, const f16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(f32x4& acc, const f32x4& a, const f32x4& b) {
  // exact fp32: MFMA j multiplies element j of every lane's k-group, so the four cover the 16 channels once
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc, 0, 0, 0);
}

// TDP = tile depth: 4 (4x8x8 voxels, wave = depth slice = four 16-voxel blocks) or 2 (2x8x8 voxels, wave =
// (depth slice, h half) = two blocks) -- the small tile doubles the workgroup count of the 24^3 layers.
// Accumulator block (mb, nb) of a wave: voxels = h rows {2 mb, 2 mb + 1} of its slice (16 = 2 x 8), channels
// nb * 16 .. + 15; lane (c = lane & 15, rq = lane >> 4) holds channel nb * 16 + c of voxels rq * 4 + j, j = 0..3,
// i.e. h row 2 mb + (rq >> 1), w = (rq & 1) * 4 + j.
template <typename T, int TDP = 4>
__global__ __launch_bounds__(256, 2) void conv3d_k3_kernel(Conv3Args a) {
  using namespace c3;
  constexpr int TD = TDP, HD = TDP + 2, MB = TDP;       // MB: 16-voxel blocks per wave
  constexpr int HALO_BYTES = HD * PS, LDS_MAIN = HALO_BYTES + NSLAB * SLAB;
  constexpr int NITEMS = HD * HH * HW * KG, NIT = (NITEMS + 255) / 256;
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
  const int c16 = lane & 15, rq = lane >> 4;
  const int dwave = TDP == 4 ? wave : wave >> 1;          // depth slice of this wave
  const int hbase = TDP == 4 ? 0 : (wave & 1) * 4;        // first h row of this wave's blocks
  const int tile = xcd_remap(blockIdx.x, a.ntiles);
  const int ct = blockIdx.y, n = blockIdx.z % a.N;
  const int tw_ = tile % a.tiles_w, th_ = (tile / a.tiles_w) % a.tiles_h, td_ = tile / (a.tiles_w * a.tiles_h);
  const int d0 = td_ * TD, h0 = th_ * TH, w0 = tw_ * TW;
  const bool fused = a.xf.stats != nullptr;

  const int kg_t = tid & (KG - 1);
  int goff[NIT]; int loff[NIT];           // element offset inside the sample (< 2^31, checked by the launcher) / LDS byte offset
  const T* xin = (const T*)a.x + (long)n * a.D * a.H * a.W * a.Cin_stride + a.Cin_off;
#pragma unroll
  for (int j = 0; j < NIT; ++j) {
    int it = tid + 256 * j;
    int hv = it >> 2;
    int hd = hv / (HH * HW), rem = hv - hd * (HH * HW), hy = rem / HW, hx = rem - hy * HW;
    int gd = d0 + hd - 1, gh = h0 + hy - 1, gw = w0 + hx - 1;
    bool ok = it < NITEMS && gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
    goff[j] = ok ? ((gd * a.H + gh) * a.W + gw) * a.Cin_stride + kg_t * EPG : -1;
    loff[j] = it < NITEMS ? hd * PS + hy * RS + hx * VS + kg_t * 16 : -1;
  }
  // Weight slabs go global -> LDS by LDS-DMA: 12 pieces of 1 KB per slab, three per wave, destination = wave-uniform
  // base + lane * 16 (the packed layout is exactly the LDS image).  The ring holds three slabs: slab g+2 is requested at
  // the start of phase g into the buffer phase g-1 read (every wave is past that phase's barrier), so an L2 round trip
  // (~1 us under load, about one phase) has two phases to complete.  Vector-memory operations retire in issue order,
  // so "slab g+1 has landed" is a COUNTED wait at the end of phase g: everything but the youngest operations (the 3
  // pieces of slab g+2, plus this phase's halo prefetch when it was issued after them) must be done.
  const char* wsrc = (const char*)a.w + (long)ct * a.nchunks * 9 * SLAB + (wave * 3) * 1024 + lane * 16;
  auto dma_slab = [&](int g, int gl) {       // g = slab index within the cout tile (chunk * 9 + kd * 3 + kh); gl = g - g0
    if (ABL & 4) return;
    const char* src = wsrc + (long)g * SLAB;
    char* dst = wbuf + (gl % NSLAB) * SLAB + (wave * 3) * 1024;
#pragma unroll
    for (int j = 0; j < 3; ++j) glds16(src + j * 1024, dst + j * 1024);
  };
  Frag hv_[NIT];
  // Loads are unconditional (an item outside the volume or past Cin reads the sample's first k-group and is zeroed when
  // it is stored): a load under a lane-dependent branch gets its own basic block and its own wait from hipcc.
  auto load_halo = [&](int ch) {
    if (ABL & 8) return;
    const bool cok = ch * CK + kg_t * EPG < a.Cin;
#pragma unroll
    for (int j = 0; j < NIT; ++j) hv_[j] = *(const Frag*)(xin + ((goff[j] >= 0 && cok) ? goff[j] + ch * CK : 0));
  };
  auto store_halo = [&](int ch) {
    if (ABL & 8) return;
    const int c0 = ch * CK + kg_t * EPG;
    const bool cok = c0 < a.Cin;
    float sc[EPG], sh[EPG], ad[EPG];
    if (fused && cok) {
#pragma unroll
      for (int e = 0; e < EPG; ++e) { sc[e] = xsc[c0 + e]; sh[e] = xsh[c0 + e]; ad[e] = xad[c0 + e]; }
    }
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      Frag f = hv_[j];
      if (fused && cok) f = xform_frag<T>(f, sc, sh, ad, a.xf.slope);
      const bool ok = goff[j] >= 0 && cok;
#pragma unroll
      for (int e = 0; e < EPG; ++e) f[e] = ok ? f[e] : (T)0.f;             // padding stays literal zero
      if (loff[j] >= 0) *(Frag*)(halo + loff[j]) = f;
    }
  };

  // accumulators start at the bias of the lane's output channel (split-K adds it in the finish kernel instead)
  f32x4 acc[MB][4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const float b0 = a.ksplit > 1 ? 0.f : a.bias[ct * BN + nb * 16 + c16];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[m][nb][i] = b0;
  }

  // ---- work range: units u = chunk * 3 + kd, three (kd, kh) slabs each ----
  const int ks_id = blockIdx.z / a.N;
  const int u0 = ks_id * a.units_per_split;
  const int u1 = min(a.nchunks * 3, u0 + a.units_per_split);
  const int g0 = u0 * 3, g1 = u1 * 3;

  // ---- prologue ----
  dma_slab(g0, 0);
  if (g0 + 1 < g1) dma_slab(g0 + 1, 1);
  load_halo(u0 / 3);
  if (fused) {     // only the Cin chunks this workgroup walks (all of them unless split-K)
    xform_preamble(a.xf, n, min(a.Cin, ((u1 + 2) / 3) * CK), xsc, xsh, xad, (u0 / 3) * CK);
    __syncthreads();
  }
  store_halo(u0 / 3);
  asm volatile("s_waitcnt vmcnt(3)" ::: "memory");      // slab g0 has landed; slab g0+1 may still be in flight
  if (g0 + 1 >= g1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  const int a_base = dwave * PS + (hbase + (c16 >> 3)) * RS + (c16 & 7) * VS + rq * 16;
  const int b_base = (rq * BN + c16) * 16;
  int ring = 0;                                          // (g - g0) % NSLAB without a division
  for (int u = u0; u < u1; ++u) {
    const int kd = u % 3;
    const bool next_chunk = kd == 2 && u + 1 < u1;     // the unit after this one starts a new Cin chunk
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int g = u * 3 + kh;
      if (g + 2 < g1) dma_slab(g + 2, g + 2 - g0);
      const bool halo_now = kh == 0 && next_chunk;
      if (halo_now) load_halo(u / 3 + 1);
      const char* ap = halo + a_base + kd * PS + kh * RS;
      const char* wb = wbuf + ring * SLAB + b_base;
      ring = ring == NSLAB - 1 ? 0 : ring + 1;
      // three k-steps (kw); fragments of step t+1 are in flight while the MFMAs of step t issue
      Frag fa[2][MB], fb[2][4];
      auto ld = [&](int kw, int b) {
        if (ABL & 2) return;
#pragma unroll
        for (int m = 0; m < MB; ++m) fa[b][m] = *(const Frag*)(ap + 2 * m * RS + kw * VS);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) fb[b][nb] = *(const Frag*)(wb + (kw * KG * BN + nb * 16) * 16);
      };
      ld(0, 0);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        if (t + 1 < 3) ld(t + 1, (t + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);       // keep the prefetch ahead of the MFMAs (hipcc sinks it otherwise)
        if (ABL & 1) {
#pragma unroll
          for (int m = 0; m < MB; ++m) asm volatile("" ::"v"(fa[t & 1][m]));
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) asm volatile("" ::"v"(fb[t & 1][nb]));
        } else {
#pragma unroll
          for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) mma16(acc[m][nb], fa[t & 1][m], fb[t & 1][nb]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // slab g+1 must have landed before anyone passes the barrier.  Younger than its pieces: the 3 pieces of slab g+2
      // (when requested) and, in the phase that prefetches the next halo, those NIT loads -- or fewer, if they have
      // already been consumed; a count that is too small only waits longer.
      if (g + 2 < g1) {
        if (halo_now) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 + NIT) : "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // next slab visible to every wave, and everyone is done with this one
    }
    if (next_chunk) {
      store_halo(u / 3 + 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  __syncthreads();     // the epilogue reuses the halo and weight buffers

  if (ABL & 16) {
    if (acc[0][0][0] == 123.f && acc[0][1][1] == 5.f) ((float*)a.y)[0] = 1.f;
    return;
  }
  // ---- epilogue, one 32-channel half (nb pair) at a time: the wave's 16*MB voxels x 32 channels go through an LDS
  // staging tile so that global stores are 16 bytes per lane and whole 64-byte (fp16) runs per voxel ----
  const int gd = d0 + dwave;
  const bool dok = gd < a.D;
  const bool full = d0 + TD <= a.D && h0 + TH <= a.H && w0 + TW <= a.W;
  const bool partial_out = a.ksplit > 1;        // split-K: fp32 partial tile to part[ks][n][voxel][cout_pad], no statistics
  constexpr int NV = 16 * MB;                   // voxels per wave
  auto stage_and_store = [&](auto* out_base, int out_stride, int q, bool want_stats, float* ex) {
    using OT = typename std::remove_pointer<decltype(out_base)>::type;
    constexpr int OEPG = 16 / (int)sizeof(OT);
    constexpr int OS = 32 * (int)sizeof(OT) + 16;
    char* ot = smem + wave * NV * OS;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int nb = 2 * q + h2;
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int vw = m * 16 + rq * 4 + i;            // voxel within the wave: h = hbase + (vw >> 3), w = vw & 7
          float v = acc[m][nb][i];
          if (!full) {
            const bool ok = dok && (h0 + hbase + (vw >> 3) < a.H) && (w0 + (vw & 7) < a.W);
            v = ok ? v : 0.f;
          }
          s += v;
          ss = fmaf(v, v, ss);
          *(OT*)(ot + vw * OS + (h2 * 16 + c16) * (int)sizeof(OT)) = (OT)v;
        }
      if (want_stats) {
        s += __shfl_xor(s, 16); ss += __shfl_xor(ss, 16);
        s += __shfl_xor(s, 32); ss += __shfl_xor(ss, 32);
        if (rq == 0) { ex[(wave * BN + nb * 16 + c16) * 2] = s; ex[(wave * BN + nb * 16 + c16) * 2 + 1] = ss; }
      }
    }
    __syncthreads();
    if (dok) {
      constexpr int GPV = 32 / OEPG;            // 16-byte groups per voxel in this half
      constexpr int VPI = 64 / GPV;
      typedef OT OFrag __attribute__((ext_vector_type(OEPG)));
#pragma unroll
      for (int it = 0; it < NV / VPI; ++it) {
        const int v = it * VPI + lane / GPV, cg = lane % GPV;
        const int gh = h0 + hbase + (v >> 3), gw = w0 + (v & 7);
        if ((full || (gh < a.H && gw < a.W)) && (partial_out || ct * BN + q * 32 + cg * OEPG < a.Cout))
          *(OFrag*)(out_base + (((long)gd * a.H + gh) * a.W + gw) * out_stride + q * 32 + cg * OEPG) =
              *(const OFrag*)(ot + v * OS + cg * 16);
      }
    }
  };
  if (partial_out) {
    float* pout = a.part + ((long)(ks_id * a.N + n) * a.D * a.H * a.W) * a.cout_pad + ct * BN;
    stage_and_store(pout, a.cout_pad, 0, false, (float*)nullptr);
    __syncthreads();
    stage_and_store(pout, a.cout_pad, 1, false, (float*)nullptr);
    return;
  }
  float* ex = (float*)(smem + 4 * NV * (32 * 4 + 16));   // [4 waves][64 couts][2], behind the largest staging tile
  T* yout = (T*)a.y + (long)n * a.D * a.H * a.W * a.Cout_stride + a.Cout_off + ct * BN;
  stage_and_store(yout, a.Cout_stride, 0, true, ex);
  __syncthreads();               // staging tile is reused by the second half
  stage_and_store(yout, a.Cout_stride, 1, true, ex);
  if (wave == 0) {
    double S = 0, Q = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { S += (double)ex[(w * BN + lane) * 2]; Q += (double)ex[(w * BN + lane) * 2 + 1]; }
    if (ct * BN + lane < a.Cout) stats_add(a.stats, n, a.cout_pad, blockIdx.x & (STAT_REPLICAS - 1), ct * BN + lane, S, Q);
  }
}

// ---- split-K finish: y = sum_k part[k] + bias (stored as T), and this layer's InstanceNorm sums ----
// block = 256 threads = VL voxel lanes x G channel groups of 4; each thread walks ITER voxels.
template <typename T>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ part, int ksplit, int N, long vox,
                                                            int cout_pad, int Cout, const float* __restrict__ bias,
                                                            T* __restrict__ y, int Cout_stride, int Cout_off,
                                                            double* stats, int G, int VL, int ITER) {
  __shared__ float red[256][8];
  const int n = blockIdx.y;
  const int cg = threadIdx.x % G, vl = threadIdx.x / G;
  const int c = cg * 4;
  float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  f32x4 b4 = *(const f32x4*)(bias + c);
  if (vl < VL) {
    for (int i = 0; i < ITER; ++i) {
      const long v = ((long)blockIdx.x * ITER + i) * VL + vl;
      if (v >= vox) break;
      f32x4 acc = b4;
      const float* pp = part + ((long)n * vox + v) * cout_pad + c;
      const long kstride = (long)N * vox * cout_pad;
      int k = 0;
      for (; k + 4 <= ksplit; k += 4) {          // four independent loads in flight
        const f32x4 p0 = *(const f32x4*)(pp + (k + 0) * kstride), p1 = *(const f32x4*)(pp + (k + 1) * kstride);
        const f32x4 p2 = *(const f32x4*)(pp + (k + 2) * kstride), p3 = *(const f32x4*)(pp + (k + 3) * kstride);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += (p0[e] + p1[e]) + (p2[e] + p3[e]);
      }
      for (; k < ksplit; ++k) {
        const f32x4 p = *(const f32x4*)(pp + k * kstride);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += p[e];
      }
      T o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (T)acc[e];
        const float f = (float)o[e];
        s[e] += f; q[e] += f * f;
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
  if (vl == 0 && c < Cout) {
    double S[4] = {0, 0, 0, 0}, Q[4] = {0, 0, 0, 0};
    for (int j = 0; j < VL; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) { S[e] += (double)red[j * G + cg][e]; Q[e] += (double)red[j * G + cg][4 + e]; }
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c + e < Cout) stats_add(stats, n, cout_pad, blockIdx.x & (STAT_REPLICAS - 1), c + e, S[e], Q[e]);
  }
}


static inline void choose_split(int base_wgs, int units, int* ksplit, int* ups) {
  *ksplit = 1; *ups = units;
  if (base_wgs > 64 || units <= 1) return;        // 24^3 and up: the partial-tile round trip costs more than it buys
  int want = (320 + base_wgs - 1) / base_wgs;      // ~one workgroup per CU, each keeping >= a few units of work
  if (want > units) want = units;
  *ups = (units + want - 1) / want;
  *ksplit = (units + *ups - 1) / *ups;
}

// hipFuncSetAttribute is per device: remember which devices have the dynamic-LDS limit raised
template <typename T>
static int ensure_lds_attr() {
  static bool done[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return DUA_ERR_ARG;
  if (done[dev]) return 0;
  hipError_t e = hipFuncSetAttribute((const void*)conv3d_k3_kernel<T, 4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     c3::lds_main(4) + c3::XF_MAX);
  if (e == hipSuccess)
    e = hipFuncSetAttribute((const void*)conv3d_k3_kernel<T, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            c3::lds_main(2) + c3::XF_MAX);
  if (e != hipSuccess) return (int)e;
  done[dev] = true;
  return 0;
}

template <typename T>
static int launch_conv3(const dua_conv3_desc* d, const void* x, const void* w, const float* bias,
                        const dua_in_norm* in, void* y, double* stats, float* ws, long ws_bytes, hipStream_t s) {
  using namespace c3;
  constexpr int CK = KG * Elem<T>::EPG;
  constexpr int TD = 4;
  Conv3Args a;
  a.x = x; a.w = w; a.bias = bias; a.y = y; a.stats = stats;
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
  a.ksplit = 1; a.units_per_split = a.nchunks * 3; a.part = nullptr;
  if (a.nchunks * CK > 1024) return DUA_ERR_ARG;
  const long vox = (long)d->D * d->H * d->W;
  if (vox * d->Cin_stride >= (1L << 31)) return DUA_ERR_ARG;       // per-sample element offsets are 32-bit in the kernel
  const int xf_bytes = in && in->stats ? 3 * 4 * a.nchunks * CK : 0;
  if (int e = ensure_lds_attr<T>()) return e;
  if (ws != nullptr && g_conv_variant == 0) {
    int ks, ups;
    choose_split(a.ntiles * nct * d->N, a.nchunks * 3, &ks, &ups);
    if (ks > 1 && (long)ks * d->N * vox * a.cout_pad * 4 <= ws_bytes) { a.ksplit = ks; a.units_per_split = ups; a.part = ws; }
  }
  // 24^3-sized layers (too few 4x8x8 tiles for 256 CUs, too big for split-K to pay): 2x8x8 tiles, twice the workgroups
  if (a.ksplit == 1 && ((g_conv_variant == 0 && a.ntiles * nct * d->N < 200 && a.ntiles * nct * d->N > 64) || g_conv_variant == 3)) {
    const int td2 = (d->D + 1) / 2;
    a.ntiles = td2 * a.tiles_h * a.tiles_w;
    dim3 grid2(a.ntiles, nct, d->N);
    hipLaunchKernelGGL((conv3d_k3_kernel<T, 2>), grid2, dim3(256), lds_main(2) + xf_bytes, s, a);
    return (int)hipGetLastError();
  }
  dim3 grid(a.ntiles, nct, d->N * a.ksplit);
  hipLaunchKernelGGL((conv3d_k3_kernel<T, 4>), grid, dim3(256), lds_main(4) + xf_bytes, s, a);
  if (a.ksplit > 1) {
    if (g_skip_splitk_finish) return (int)hipGetLastError();
    const int G = a.cout_pad / 4 > 256 ? 256 : a.cout_pad / 4;       // channel groups handled per block pass
    if (a.cout_pad / 4 > 256) return DUA_ERR_ARG;
    const int VL = 256 / G;
    int ITER = 8;
    while (ITER > 1 && (vox + (long)VL * ITER - 1) / ((long)VL * ITER) < 128) ITER >>= 1;   // >= ~128 blocks; fewer blocks = fewer fp64 atomics
    dim3 fgrid((unsigned)((vox + (long)VL * ITER - 1) / ((long)VL * ITER)), d->N);
    hipLaunchKernelGGL(splitk_finish_kernel<T>, fgrid, dim3(256), 0, s, (const float*)ws, a.ksplit, d->N, vox, a.cout_pad,
                       d->Cout, bias, (T*)y, d->Cout_stride, d->Cout_off, stats, G, VL, ITER);
  }
  return (int)hipGetLastError();
}

}  // namespace dua

extern "C" {

int dua_set_option(int key, int value) {
  if (key == 1 && (value == 0 || value == 2 || value == 3)) { dua::g_conv_variant = value; return 0; }
  if (key == 2 && (value == 0 || value == 1)) { dua::g_skip_splitk_finish = value; return 0; }
#ifdef DUA_ABLATE
  if (key == 3 && value >= 0 && value < 16) { dua::g_wgrad_abl = value; return 0; }   // diagnostic builds only
#endif
  if (key == 4 && value >= 0 && value < 128) { dua::g_wgrad_variant = value; return 0; }
  return DUA_ERR_ARG;
}

long dua_conv3d_k3_workspace(const dua_conv3_desc* d) {
  using namespace dua::c3;
  if (!d || (d->dtype != DUA_F16 && d->dtype != DUA_F32)) return DUA_ERR_ARG;
  const int ck = 4 * (d->dtype == DUA_F16 ? 8 : 4);
  const int nch = (d->Cin + ck - 1) / ck, nct = (d->Cout + BN - 1) / BN;
  const int tiles = ((d->D + 3) / 4) * ((d->H + TH - 1) / TH) * ((d->W + TW - 1) / TW);
  int ks, ups;
  dua::choose_split(tiles * nct * d->N, nch * 3, &ks, &ups);
  return ks > 1 ? (long)ks * d->N * d->D * d->H * d->W * nct * BN * 4 : 0;
}

int dua_conv3d_k3_fwd(const dua_conv3_desc* d, const void* x, const void* w_packed, const float* bias_padded,
                      const dua_in_norm* in, void* y, double* out_stats, void* workspace, long workspace_bytes,
                      void* stream) {
  if (!d || !x || !w_packed || !bias_padded || !y || !out_stats) return DUA_ERR_ARG;
  if (d->Cin % 8 || d->Cout % 8 || d->Cin_stride % 8 || d->Cout_stride % 8 || d->Cin_off % 8 || d->Cout_off % 8)
    return DUA_ERR_ARG;
  if (in && in->stats && (!in->gamma || !in->beta || in->c_pad < d->Cin)) return DUA_ERR_ARG;
  if (d->dtype == DUA_F16)
    return dua::launch_conv3<dua::f16>(d, x, w_packed, bias_padded, in, y, out_stats, (float*)workspace, workspace_bytes, (hipStream_t)stream);
  if (d->dtype == DUA_F32)
    return dua::launch_conv3<float>(d, x, w_packed, bias_padded, in, y, out_stats, (float*)workspace, workspace_bytes, (hipStream_t)stream);
  return DUA_ERR_ARG;
}

}  // extern "C"
