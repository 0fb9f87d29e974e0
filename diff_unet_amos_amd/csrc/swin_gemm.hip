// Token GEMMs of the Swin stages with tall activations and small weights (BASELINE config 5):
//   out[token][n] = sum_k A[token][k] * W[n][k] (+ bias[n])        M = 1e4 .. 1e6 tokens, K <= 384, N <= 192 per launch
// = the nn.Linear layers of WindowAttention (qkv, proj: models/swin_unetr/attention.py:91-94,99,118), of the MLP
// (MONAI MLPBlock linear1 / linear2, transformer.py:376,434) and of PatchMerging (reduction, patch.py:89-92) at the two
// fine stages (48^3 x 48 and 24^3 x 96 tokens), and the 1x1x1 conv3 of channel-changing UnetResBlocks (blocks.py:286-296).
// These are HBM-bound (a few FLOP per byte); what a library GEMM cannot do is fuse what follows, so each launch carries
// one of the epilogues the reference applies next:
//   PLAIN     + bias
//   GELU      + bias, exact GELU                                      (MLPBlock act, between linear1 and linear2)
//   STATS     raw output + per-(sample, channel) sum / sum of squares (norm3 statistics of conv3)
//   RESIDUAL  x[token] += out + bias on the fp32 stream               (x + mlp(norm2(x)), transformer.py:477-480)
//   SCATTER   window token -> voxel (window_reverse, roll back, crop), x += out + bias, ln_out = norm2(x)
//                                                                    (transformer.py:417-434, 475-476)
//
// Mapping: the weights are the A operand of MFMA 32x32x16 (rows = output channels, resident in LDS for the whole launch),
// a wave's 32 tokens are the B operand.  A tile of 128 token rows comes in with coalesced 16-byte loads (consecutive lanes =
// consecutive bytes of a row; one row per lane, the direct operand load, costs one address per lane: 97 us instead of 54 for
// the library GEMM on the 96^3 conv3, profiles/r2_swin_*), is read back as operand fragments from padded LDS rows, and the
// result leaves the same way: accumulators (per LANE = TOKEN, four consecutive output channels per register quad -- bias,
// GELU and the statistics are register arithmetic) -> LDS tile -> coalesced rows.  The scatter epilogue walks the tile
// with 16 lanes per token (LayerNorm = a 16-lane butterfly), like window_scatter_add_norm.
#include <mutex>
#include "common.hpp"
#include "../../include/dua_hip.h"
#include "swin_geom.hpp"

namespace dua {

namespace tg {
constexpr int MAXNB_ALL = 6;    // 32-row blocks of output channels per launch (N <= 192)
constexpr int KSTEPS = 12;      // k-steps (16 channels) resident per pass (192 channels)
}  // namespace tg

struct TokLinArgs {
  const f16* A; int lda; long M; int K, N;
  const f16* W; const float* bias;
  int mode;
  f16* out; int ldc, out_off;
  float* x;
  stat_t* stats; int c_pad;
  WinGeom g; const float* gamma; const float* beta; float eps; f16* ln_out;
  int w_row, a_row, o_row, w_bytes;   // LDS row strides (bytes) of W, the A tile, the output tile; size of the W region
};

// LDS row stride for rows of `bytes` payload read 16 bytes per lane with one row per lane: an odd number of 16-byte units
// keeps the 16 lanes of a read phase on distinct bank groups.
static inline int padded_row(int bytes) { return (bytes / 16) & 1 ? bytes : bytes + 16; }

template <int MODE, int NB>
__global__ __launch_bounds__(256) void token_linear_kernel(TokLinArgs a) {
  using namespace tg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Wl = smem;                       // [NB*32][w_row]   weights, resident
  char* At = smem + a.w_bytes;           // [128][a_row]     activation tile (one K chunk), then the output tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int K = a.K, N = a.N, kg = K >> 3;
  for (int i0 = 0; i0 < NB * 32 * kg; i0 += 4 * 256) {           // weights: four 16-byte pieces per thread in flight (a load per
    f16x8 wv[4];                                                  // iteration behind a row test went to memory one at a time)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 256 + tid;
      const int n = i / kg, g8 = i - n * kg;
      wv[u] = *(const f16x8*)(a.W + (n < N ? (long)n * K + g8 * 8 : 0));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 256 + tid;
      const int n = i / kg, g8 = i - n * kg;
      f16x8 v = wv[u];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = n < N ? v[e] : (f16)0.f;
      if (i < NB * 32 * kg) *(f16x8*)(Wl + n * a.w_row + g8 * 16) = v;
    }
  }
  const int sample = blockIdx.y;                                  // STATS: one grid row per sample
  const f16* A = a.A + (long)sample * a.M * a.lda;
  constexpr int SNB = MODE == DUA_TOKLIN_STATS ? (NB < 2 ? NB : 2) : 1;
  float ssum[SNB][16], ssq[SNB][16];                              // STATS (N <= 64): per-lane partial sums over this workgroup's tiles
  if (MODE == DUA_TOKLIN_STATS) {
#pragma unroll
    for (int nb = 0; nb < SNB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) { ssum[nb][i] = 0.f; ssq[nb][i] = 0.f; }
  }
  const long tiles = (a.M + 127) / 128;
  // conv3 with K <= 96 (one pass per tile): the NEXT tile's rows are requested as soon as this
  // tile's are in LDS and travel under its MFMAs, epilogue and copy-out -- a tile used to be load latency + compute + store
  // latency in a row (8.9 us per tile and workgroup on the 96^3 conv3, 3.1 TB/s with three workgroups per CU).
  // (the other epilogues lose more to the 55 extra registers -- a workgroup less per CU -- than the overlap gains: measured; for
  // the qkv form, which has two workgroups per CU by its LDS tile whatever its registers, it changes nothing: 20.1 vs 20.4 us)
  const bool single = MODE == DUA_TOKLIN_STATS && K <= 96;
  constexpr int PF = 128 * 12 / 256;                              // 16-byte pieces per thread of a 128 x 96 tile
  f16x8 pf[PF];
  const int cpr1 = K >> 3;
  auto fetch = [&](long tok0) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int c = tid + 256 * j;
      const int row = c / cpr1, col = c - row * cpr1;
      const bool ok = c < 128 * cpr1 && tok0 + row < a.M;
      pf[j] = *(const f16x8*)(A + (ok ? (tok0 + row) * a.lda + col * 8 : 0));
      if (!ok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[j][e] = (f16)0.f;
      }
    }
  };
  if (single && (long)blockIdx.x < tiles) fetch(blockIdx.x * 128L);
  for (long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const long tok0 = tile * 128;
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16 * KSTEPS) {
      const int kc = K - k0 < 16 * KSTEPS ? K - k0 : 16 * KSTEPS, cpr = kc >> 3;
      __syncthreads();                                            // W staged / the previous tile's copy-out is done
      if (single) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
          const int c = tid + 256 * j;
          const int row = c / cpr1, col = c - row * cpr1;
          if (c < 128 * cpr1) *(f16x8*)(At + row * a.a_row + col * 16) = pf[j];
        }
        if (tile + gridDim.x < tiles) fetch((tile + gridDim.x) * 128);
      } else {
        for (int c = tid; c < 128 * cpr; c += 256) {              // coalesced: consecutive lanes, consecutive 16 bytes of a row
          const int row = c / cpr, col = c - row * cpr;
          f16x8 v;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (f16)0.f;
          if (tok0 + row < a.M) v = *(const f16x8*)(A + (tok0 + row) * a.lda + k0 + col * 8);
          *(f16x8*)(At + row * a.a_row + col * 16) = v;
        }
      }
      __syncthreads();
      const char* arow = At + (wave * 32 + r) * a.a_row;
      for (int ks = 0; ks * 16 < kc; ++ks) {
        const int kk = 16 * ks + 8 * hh;
        f16x8 bf;
#pragma unroll
        for (int e = 0; e < 8; ++e) bf[e] = (f16)0.f;
        if (kk < kc) bf = *(const f16x8*)(arow + kk * 2);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          // K % 16 == 8: the upper lane half of the last k-step lies behind the row (next row's weights, or the slack
          // behind the last row: 0 x NaN would poison the output) -- it contributes zero weights
          f16x8 af = bf;                                           // zeros when kk >= kc
          if (kk < kc) af = *(const f16x8*)(Wl + (nb * 32 + r) * a.w_row + (k0 + kk) * 2);
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[nb], 0, 0, 0);
        }
      }
    }
    __syncthreads();                                              // every wave is done reading the A tile: reuse it for the output
    // ---- registers -> output tile in LDS.  Register quad j of block nb = channels nb*32 + 8j + 4hh + (0..3) of token r ----
    const long tok = tok0 + wave * 32 + r;
    const bool valid = tok < a.M;
    char* orow = At + (wave * 32 + r) * a.o_row;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c0 = nb * 32 + 8 * j + 4 * hh;
        if (c0 < N) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[nb][4 * j + e] + (a.bias ? a.bias[c0 + e] : 0.f);
            if (MODE == DUA_TOKLIN_GELU) v[e] = gelu_erf(v[e]);
          }
          if (MODE == DUA_TOKLIN_RESIDUAL || MODE == DUA_TOKLIN_SCATTER) {
            *(f32x4*)(orow + c0 * 4) = f32x4{v[0], v[1], v[2], v[3]};
          } else {
            f16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (f16)v[e];
            *(f16x4*)(orow + c0 * 2) = o;
            if (MODE == DUA_TOKLIN_STATS && valid && nb < SNB) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float q = (float)o[e];                     // statistics of what the consumer will read
                ssum[nb < SNB ? nb : 0][4 * j + e] += q;
                ssq[nb < SNB ? nb : 0][4 * j + e] = fmaf(q, q, ssq[nb < SNB ? nb : 0][4 * j + e]);
              }
            }
          }
        }
      }
    }
    __syncthreads();
    // ---- output tile -> global memory, coalesced ----
    if (MODE == DUA_TOKLIN_PLAIN || MODE == DUA_TOKLIN_GELU || MODE == DUA_TOKLIN_STATS) {
      const int cpr = N >> 3;
      for (int c = tid; c < 128 * cpr; c += 256) {
        const int row = c / cpr, col = c - row * cpr;
        if (tok0 + row < a.M)
          *(f16x8*)(a.out + ((long)sample * a.M + tok0 + row) * a.ldc + a.out_off + col * 8) = *(const f16x8*)(At + row * a.o_row + col * 16);
      }
    } else if (MODE == DUA_TOKLIN_RESIDUAL) {
      // x += out as 16-byte read-modify-writes of the tile's rows (contiguous in x).  Every piece is REQUESTED before the first
      // is stored: as a load - add - store loop the store of one piece and the load of the next may alias as far as the
      // compiler knows, so the pieces went to memory one round trip after the other (4 NB of them per tile).
      const int cpr = N >> 2;
      constexpr int MAXIT = 4 * NB;
      float* xt = a.x + tok0 * N;
      const long left = (a.M - tok0) * cpr;                        // pieces of this tile that are rows of x
      f32x4 xv[MAXIT];
#pragma unroll
      for (int j = 0; j < MAXIT; ++j) {
        const int c = tid + 256 * j;
        xv[j] = *(const f32x4*)(xt + (c < 128 * cpr && c < left ? c * 4 : 0));
      }
#pragma unroll
      for (int j = 0; j < MAXIT; ++j) {
        const int c = tid + 256 * j;
        const int row = c / cpr, col = c - row * cpr;
        if (c < 128 * cpr && c < left) {
          const f32x4 dv = *(const f32x4*)(At + row * a.o_row + col * 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) xv[j][e] += dv[e];
          *(f32x4*)(xt + c * 4) = xv[j];
        }
      }
    } else {                                                       // DUA_TOKLIN_SCATTER
      // G = N / 12 lanes per token (4 / 8 / 16 for N = 48 / 96 / 192), each with three pieces of four consecutive channels
      // (piece = i * G + lane-in-group): the shortcut, the stream and the LayerNorm output move as 16- and 8-byte accesses
      // (one channel per lane and access took three times the memory instructions).  The shortcut pieces of two passes are
      // requested before the first row is written back: a store to x and the next row's load from x may alias for the compiler.
      const WinGeom& g = a.g;
      const int G = N / 12, lg = G == 4 ? 2 : (G == 8 ? 3 : 4);
      const int jl = tid & (G - 1), rpp = 256 >> lg, npass = 128 / rpp;
      for (int p0 = 0; p0 < npass; p0 += 2) {                     // two passes at a time: 128 registers = four workgroups per CU
        long dstv[2];
        f32x4 xs[2][3];
#pragma unroll
        for (int q4 = 0; q4 < 2; ++q4) {
          const long tk = tok0 + (tid >> lg) + rpp * (p0 + q4);
          const bool inb = p0 + q4 < npass && tk < a.M;
          const long tkc = inb ? tk : 0;
          const int t = (int)(tkc % g.n), wi = (int)((tkc / g.n) % g.nw), b = (int)(tkc / ((long)g.n * g.nw));
          int d, h, w;
          const bool real = window_to_voxel(g, wi, t, d, h, w) && inb;
          dstv[q4] = real ? ((((long)b * g.D + d) * g.H + h) * g.W + w) * N : -1;
#pragma unroll
          for (int i = 0; i < 3; ++i) xs[q4][i] = *(const f32x4*)(a.x + (real ? dstv[q4] + (i * G + jl) * 4 : 0));
        }
#pragma unroll
        for (int q4 = 0; q4 < 2; ++q4) {
          if (p0 + q4 >= npass) break;
          const int row = (tid >> lg) + rpp * (p0 + q4);
          const bool real = dstv[q4] >= 0;
          float v[12];
          float s = 0.f;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const f32x4 dv = *(const f32x4*)(At + row * a.o_row + (i * G + jl) * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[4 * i + e] = dv[e] + (real ? xs[q4][i][e] : 0.f); s += v[4 * i + e]; }
          }
          for (int o2 = G >> 1; o2 > 0; o2 >>= 1) s += __shfl_xor(s, o2);
          const float mean = s / (float)N;
          float q = 0.f;
#pragma unroll
          for (int i = 0; i < 12; ++i) { const float dl = v[i] - mean; q = fmaf(dl, dl, q); }
          for (int o2 = G >> 1; o2 > 0; o2 >>= 1) q += __shfl_xor(q, o2);
          const float rstd = rsqrtf(q / (float)N + a.eps);
          if (real) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              const int c = (i * G + jl) * 4;
              *(f32x4*)(a.x + dstv[q4] + c) = f32x4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
              const f32x4 gmv = *(const f32x4*)(a.gamma + c), btv = *(const f32x4*)(a.beta + c);    // L1 hits
              f16x4 lo;
#pragma unroll
              for (int e = 0; e < 4; ++e) lo[e] = (f16)((v[4 * i + e] - mean) * rstd * gmv[e] + btv[e]);
              *(f16x4*)(a.ln_out + dstv[q4] + c) = lo;
            }
          }
        }
      }
    }
  }
  if (MODE == DUA_TOKLIN_STATS) {
    // Lanes -> waves -> one pair of WIDE atomic instructions per workgroup (consecutive lanes = consecutive channels).
    // Two active lanes per atomic instruction, 32 instructions per wave, cost 100 us at 768 workgroups: the memory side
    // handles one instruction's lanes together, so the number of atomic INSTRUCTIONS is what counts.
    __syncthreads();
    float* red = (float*)At;                                       // [4 waves][64 channels][2]
#pragma unroll
    for (int nb = 0; nb < SNB; ++nb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float s = ssum[nb][i], ss = ssq[nb][i];
#pragma unroll
        for (int o2 = 16; o2 > 0; o2 >>= 1) { s += __shfl_xor(s, o2); ss += __shfl_xor(ss, o2); }
        const int c = nb * 32 + 8 * (i >> 2) + 4 * hh + (i & 3);
        if (r == 0) { red[(wave * 64 + c) * 2] = s; red[(wave * 64 + c) * 2 + 1] = ss; }
      }
    }
    __syncthreads();
    if (tid < N) {
      double S = 0, Q = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) { S += red[(w * 64 + tid) * 2]; Q += red[(w * 64 + tid) * 2 + 1]; }
      stats_add(a.stats, sample, a.c_pad, blockIdx.x % STAT_REPLICAS, tid, S, Q);
    }
  }
}

// ---- the whole MLP of a Swin block in one kernel (fp16 operands) -----------------------------------------------------------
//   x += linear2(GELU(linear1(ln2)))          (MONAI MLPBlock; transformer.py:376,433-434,477-480), hidden = 4 C
// linear1's accumulator (lane = token, registers = hidden units) IS linear2's B operand after GELU and a conversion to fp16:
// the 16 hidden units a lane pair holds in registers [8s, 8s+8) of block nb are one k-step of the second product, provided
// the rows of W2 are stored in the same permuted order (k = nb*32 + 16 s + (e & 3) + 8 (e >> 2) + 4 hh) -- the hidden
// activation (42 MB per block at 48^3 tokens) never exists in memory.  Hidden units are walked 192 at a time (6 accumulator
// blocks); with hidden = 192 (C = 48) both weight matrices stay resident in LDS for the whole launch, with hidden = 384
// (C = 96) each pass reloads its halves.
struct MlpArgs {
  const f16* ln; long M; int C, hidden;
  const f16* W1; const float* b1; const f16* W2; const float* b2;
  float* x;
  int w1_row, w2_row, a_row, o_row;       // LDS row strides in bytes
};

template <int C>
__global__ __launch_bounds__(256) void swin_mlp_kernel(MlpArgs a) {
  constexpr int HB = 6, HP = 32 * HB;             // hidden units per pass
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NB2 = (C + 31) >> 5, npass = 4 * C / HP;
  char* W1l = smem;                               // [HP][w1_row]            rows = hidden units of the pass, k = C
  char* W2l = W1l + HP * a.w1_row;                // [NB2*32][w2_row]        rows = output channels, k = permuted hidden units of the pass
  char* At = W2l + NB2 * 32 * a.w2_row;           // [128][a_row] ln2 tile, later [128][o_row] fp32 output tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  constexpr int kg = C >> 3;
  auto load_weights = [&](int p) {
    for (int i = tid; i < HP * kg; i += 256) {                    // W1 rows [p*HP, (p+1)*HP)
      const int n = i / kg, g8 = i - n * kg;
      *(f16x8*)(W1l + n * a.w1_row + g8 * 16) = *(const f16x8*)(a.W1 + (long)(p * HP + n) * C + g8 * 8);
    }
    // W2p[c][nb][s][hh][8] <- W2[c][p*HP + nb*32 + 16 s + perm(hh, e)], perm = (e & 3) + 8 (e >> 2) + 4 hh: a 16-byte piece of a
    // W2 row (hidden units 8 t .. 8 t + 7 of a 16-unit half block, t = 0 / 1) is elements 4 t .. 4 t + 3 of BOTH hh fragments --
    // one 16-byte load and two 8-byte LDS writes (gathered per element it was 48 two-byte global loads per thread and launch)
    for (int i0 = 0; i0 < NB2 * 32 * (HP / 8); i0 += 4 * 256) {
      f16x8 wv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 256 + tid, c = i / (HP / 8), hc = i - c * (HP / 8);
        wv[u] = *(const f16x8*)(a.W2 + (c < C ? (long)c * a.hidden + p * HP + hc * 8 : 0));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 256 + tid, c = i / (HP / 8), hc = i - c * (HP / 8);
        if (i < NB2 * 32 * (HP / 8)) {
          const int blk = hc >> 1, t = hc & 1;                      // blk = nb * 2 + s
          f16x4 lo, hi;
#pragma unroll
          for (int e = 0; e < 4; ++e) { lo[e] = c < C ? wv[u][e] : (f16)0.f; hi[e] = c < C ? wv[u][4 + e] : (f16)0.f; }
          *(f16x4*)(W2l + c * a.w2_row + (blk * 2 + 0) * 16 + t * 8) = lo;
          *(f16x4*)(W2l + c * a.w2_row + (blk * 2 + 1) * 16 + t * 8) = hi;
        }
      }
    }
  };
  if constexpr (npass == 1) load_weights(0);
  const long tiles = (a.M + 127) / 128;
  for (long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const long tok0 = tile * 128;
    __syncthreads();
    for (int c8 = tid; c8 < 128 * kg; c8 += 256) {                // ln2 tile, coalesced
      const int row = c8 / kg, col = c8 - row * kg;
      f16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (f16)0.f;
      if (tok0 + row < a.M) v = *(const f16x8*)(a.ln + (tok0 + row) * C + col * 8);
      *(f16x8*)(At + row * a.a_row + col * 16) = v;
    }
    f32x16 acc2[NB2];
#pragma unroll
    for (int nb = 0; nb < NB2; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc2[nb][i] = 0.f;
    for (int p = 0; p < npass; ++p) {
      if constexpr (npass > 1) { __syncthreads(); load_weights(p); }
      __syncthreads();
      // ---- linear1: 192 hidden units x 32 tokens per wave ----
      f32x16 acc1[HB];
#pragma unroll
      for (int nb = 0; nb < HB; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc1[nb][i] = 0.f;
      const char* arow = At + (wave * 32 + r) * a.a_row;
      for (int ks = 0; ks * 16 < C; ++ks) {
        const f16x8 bf = *(const f16x8*)(arow + (16 * ks + 8 * hh) * 2);
#pragma unroll
        for (int nb = 0; nb < HB; ++nb) {
          const f16x8 af = *(const f16x8*)(W1l + (nb * 32 + r) * a.w1_row + (16 * ks + 8 * hh) * 2);
          acc1[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc1[nb], 0, 0, 0);
        }
      }
      // ---- + bias, GELU, to fp16: the B operand of linear2; linear2 accumulates over the passes ----
#pragma unroll
      for (int nb = 0; nb < HB; ++nb) {
        f16x8 g[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int hu = p * HP + nb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
          g[i >> 3][i & 7] = (f16)gelu_erf(acc1[nb][i] + a.b1[hu]);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int cb = 0; cb < NB2; ++cb) {
            const f16x8 af = *(const f16x8*)(W2l + (cb * 32 + r) * a.w2_row + ((nb * 2 + s2) * 2 + hh) * 16);
            acc2[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, g[s2], acc2[cb], 0, 0, 0);
          }
      }
    }
    __syncthreads();                                              // the ln2 tile is dead: reuse it for the output
    char* orow = At + (wave * 32 + r) * a.o_row;
#pragma unroll
    for (int cb = 0; cb < NB2; ++cb) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c0 = cb * 32 + 8 * j + 4 * hh;
        if (c0 < C)
          *(f32x4*)(orow + c0 * 4) = f32x4{acc2[cb][4 * j] + a.b2[c0], acc2[cb][4 * j + 1] + a.b2[c0 + 1],
                                           acc2[cb][4 * j + 2] + a.b2[c0 + 2], acc2[cb][4 * j + 3] + a.b2[c0 + 3]};
      }
    }
    __syncthreads();
    constexpr int cpr = C >> 2, MAXIT = 128 * cpr / 256;            // x += out, coalesced 16-byte read-modify-writes; every piece is
    float* xt = a.x + tok0 * C;                                     // requested before the first is stored (see token_linear RESIDUAL)
    const long left = (a.M - tok0) * cpr;
    f32x4 xv[MAXIT];
#pragma unroll
    for (int j = 0; j < MAXIT; ++j) {
      const int c4 = tid + 256 * j;
      xv[j] = *(const f32x4*)(xt + (c4 < left ? c4 * 4 : 0));
    }
#pragma unroll
    for (int j = 0; j < MAXIT; ++j) {
      const int c4 = tid + 256 * j;
      const int row = c4 / cpr, col = c4 - row * cpr;
      if (c4 < left) {
        const f32x4 dv = *(const f32x4*)(At + row * a.o_row + col * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[j][e] += dv[e];
        *(f32x4*)(xt + c4 * 4) = xv[j];
      }
    }
  }
}

// ---- PatchEmbed on the matrix cores (fp16) --------------------------------------------------------------------------
// Conv3d(k = s = 2) = a [tokens x 8 Cp] x [8 Cp -> 48] GEMM whose A row is gathered from the 2x2x2 voxels of the token (each
// Cp channels = Cp * 2 contiguous bytes).  Same tile scheme as token_linear; the epilogue is stage_out's: + bias + t_proj row,
// the fp32 stream, layer_norm without affine + the encoder's map into a channels-last slice (transformer.py:270-275).
struct PatchArgs {
  const f16* in; int B, D, H, W, Cs, Cp;
  const float* wk; const float* bias; const float* tadd; int tadd_stride; float eps;
  const f16* emb; float* x; f16* out; int out_stride, out_off;
  int w_row, a_row, o_row, w_bytes;
};

__global__ __launch_bounds__(256) void patch_embed_mfma_kernel(PatchArgs a) {
  constexpr int E = 48, NB = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Wl = smem;
  char* At = smem + a.w_bytes;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int K = 8 * a.Cp, cpt = a.Cp >> 3;                          // 16-byte chunks per tap
  // W_l[n][k] = w_packed[k][n] as fp16 (rows >= 48: zero).  16-byte loads of four output channels, four in flight per thread: as
  // one 4-byte load per element (48 dependent-latency iterations per thread) this staging was half of the workgroup's lifetime.
  for (int i = tid; i < (NB * 32 - E) * K; i += 256) {
    const int n = E + i / K, k = i % K;
    *(f16*)(Wl + n * a.w_row + k * 2) = (f16)0.f;
  }
  {
    const int pieces = K * (E / 4);                                 // piece i = (k, four channels 4q..4q+3) = wk + 4 i
    for (int i0 = 0; i0 < pieces; i0 += 4 * 256) {
      f32x4 wv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 256 + tid;
        wv[u] = *(const f32x4*)(a.wk + (i < pieces ? (long)i * 4 : 0));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 256 + tid;
        const int k = i / (E / 4), q = i - k * (E / 4);
        if (i < pieces) {
#pragma unroll
          for (int e = 0; e < 4; ++e) *(f16*)(Wl + (4 * q + e) * a.w_row + k * 2) = (f16)wv[u][e];
        }
      }
    }
  }
  const int D2 = a.D / 2, H2 = a.H / 2, W2 = a.W / 2;
  const long per = (long)D2 * H2 * W2, total = a.B * per;
  const long tiles = (total + 127) / 128;
  for (long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const long tok0 = tile * 128;
    __syncthreads();
    {                                                               // thread = (row, half of the taps)
      const int row = tid >> 1, th = tid & 1;
      const long tok = tok0 + row;
      const bool ok = tok < total;
      const long tc = ok ? tok : 0;
      const int w2 = (int)(tc % W2), h2 = (int)((tc / W2) % H2), d2 = (int)((tc / ((long)W2 * H2)) % D2);
      const int b = (int)(tc / per);
      // every piece of the token's 2x2x2 gather is requested (from a clamped address) before the first is stored
      f16x8 gv[4][4];                                               // [tap of this half][16-byte chunk], Cp <= 32
#pragma unroll
      for (int tp = 0; tp < 4; ++tp) {
        const int tap = th * 4 + tp;
        const int d = 2 * d2 + (tap >> 2), h = 2 * h2 + ((tap >> 1) & 1), w = 2 * w2 + (tap & 1);
        const f16* p = a.in + ((((long)b * a.D + d) * a.H + h) * a.W + w) * a.Cs;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c < cpt) gv[tp][c] = *(const f16x8*)(p + c * 8);
      }
#pragma unroll
      for (int tp = 0; tp < 4; ++tp) {
        const int tap = th * 4 + tp;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c < cpt) {
            f16x8 v = gv[tp][c];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = ok ? v[e] : (f16)0.f;
            *(f16x8*)(At + row * a.a_row + (tap * cpt + c) * 16) = v;
          }
      }
    }
    __syncthreads();
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
    const char* arow = At + (wave * 32 + r) * a.a_row;
    for (int ks = 0; ks * 16 < K; ++ks) {
      const int kk = 16 * ks + 8 * hh;
      f16x8 bf;
#pragma unroll
      for (int e = 0; e < 8; ++e) bf[e] = (f16)0.f;
      if (kk < K) bf = *(const f16x8*)(arow + kk * 2);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const f16x8 af = *(const f16x8*)(Wl + (nb * 32 + r) * a.w_row + kk * 2);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[nb], 0, 0, 0);
      }
    }
    __syncthreads();
    char* orow = At + (wave * 32 + r) * a.o_row;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c0 = nb * 32 + 8 * j + 4 * hh;
        if (c0 < E) *(f32x4*)(orow + c0 * 4) = f32x4{acc[nb][4 * j], acc[nb][4 * j + 1], acc[nb][4 * j + 2], acc[nb][4 * j + 3]};
      }
    __syncthreads();
    // four lanes per token, three pieces of four channels each (piece = i * 4 + lane-in-group): 16-byte stream stores, 8-byte
    // output stores; the embedding pieces and the per-sample adds of both passes are requested before the first row is reduced
    const int jl = tid & 3;
    f32x4 cadd[2][3];
    f16x4 ev[2][3];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const long tok = tok0 + (tid >> 2) + 64 * it;
      const bool ok = tok < total;
      const long tc = ok ? tok : 0;
      const int b = (int)(tc / per);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int c = (i * 4 + jl) * 4;
        cadd[it][i] = *(const f32x4*)(a.bias + c);
        if (a.tadd) {
          const f32x4 ta = *(const f32x4*)(a.tadd + (long)b * a.tadd_stride + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) cadd[it][i][e] += ta[e];
        }
        ev[it][i] = f16x4{(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
        if (a.emb) ev[it][i] = *(const f16x4*)(a.emb + tc * E + c);
      }
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = (tid >> 2) + 64 * it;
      const long tok = tok0 + row;
      const bool ok = tok < total;
      float v[12];
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const f32x4 dv = *(const f32x4*)(At + row * a.o_row + (i * 4 + jl) * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[4 * i + e] = dv[e] + cadd[it][i][e]; s += v[4 * i + e]; }
      }
#pragma unroll
      for (int o2 = 2; o2 > 0; o2 >>= 1) s += __shfl_xor(s, o2);
      const float mean = s / (float)E;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 12; ++i) { const float dl = v[i] - mean; q = fmaf(dl, dl, q); }
#pragma unroll
      for (int o2 = 2; o2 > 0; o2 >>= 1) q += __shfl_xor(q, o2);
      const float rstd = rsqrtf(q / (float)E + a.eps);
      if (ok) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int c = (i * 4 + jl) * 4;
          if (a.x) *(f32x4*)(a.x + tok * E + c) = f32x4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
          f16x4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov[e] = (f16)((v[4 * i + e] - mean) * rstd + (float)ev[it][i][e]);
          *(f16x4*)(a.out + tok * a.out_stride + a.out_off + c) = ov;
        }
      }
    }
  }
}

int launch_patch_embed_mfma(int B, int D, int H, int W, int Cs, int Cp, const void* in, const float* wk, const float* bias,
                            const float* tadd, int tadd_stride, float eps, const void* emb, float* x, void* out, int out_stride,
                            int out_off, hipStream_t stream) {
  PatchArgs a;
  a.in = (const f16*)in; a.B = B; a.D = D; a.H = H; a.W = W; a.Cs = Cs; a.Cp = Cp; a.wk = wk; a.bias = bias; a.tadd = tadd;
  a.tadd_stride = tadd_stride; a.eps = eps; a.emb = (const f16*)emb; a.x = x; a.out = (f16*)out; a.out_stride = out_stride;
  a.out_off = out_off;
  const int K = 8 * Cp;
  a.w_row = padded_row(K * 2); a.a_row = padded_row(((K + 15) / 16) * 32); a.o_row = padded_row(48 * 4);
  a.w_bytes = 64 * a.w_row + 16;
  const int lds = a.w_bytes + 128 * (a.a_row > a.o_row ? a.a_row : a.o_row);
  if (lds > 160 * 1024) return DUA_ERR_ARG;
  if (int e = ensure_prepared()) return e;
  const long tiles = ((long)B * (D / 2) * (H / 2) * (W / 2) + 127) / 128;
  dim3 grid((unsigned)(tiles < 512 ? tiles : 512));
  hipLaunchKernelGGL(patch_embed_mfma_kernel, grid, dim3(256), lds, stream, a);
  return (int)hipGetLastError();
}

}  // namespace dua

namespace dua {
using Kern = void (*)(TokLinArgs);
#define ROW(M_) {token_linear_kernel<M_, 1>, token_linear_kernel<M_, 2>, token_linear_kernel<M_, 3>, token_linear_kernel<M_, 4>, \
                 token_linear_kernel<M_, 5>, token_linear_kernel<M_, 6>}
static const Kern kTokLinTable[5][6] = {ROW(DUA_TOKLIN_PLAIN), ROW(DUA_TOKLIN_GELU), ROW(DUA_TOKLIN_STATS), ROW(DUA_TOKLIN_RESIDUAL),
                                        ROW(DUA_TOKLIN_SCATTER)};
#undef ROW
// dynamic-LDS limits (raised once per device by ensure_prepared(), common.hpp): every kernel here may be asked for > 64 KB
#define ATTR_ROW(M_) {(const void*)token_linear_kernel<M_, 1>, 160 * 1024}, {(const void*)token_linear_kernel<M_, 2>, 160 * 1024}, \
                     {(const void*)token_linear_kernel<M_, 3>, 160 * 1024}, {(const void*)token_linear_kernel<M_, 4>, 160 * 1024}, \
                     {(const void*)token_linear_kernel<M_, 5>, 160 * 1024}, {(const void*)token_linear_kernel<M_, 6>, 160 * 1024}
static const LdsAttr kSwinGemmLdsAttrs[] = {
    {(const void*)patch_embed_mfma_kernel, 160 * 1024},
    {(const void*)swin_mlp_kernel<48>, 160 * 1024},
    {(const void*)swin_mlp_kernel<96>, 160 * 1024},
    ATTR_ROW(DUA_TOKLIN_PLAIN), ATTR_ROW(DUA_TOKLIN_GELU), ATTR_ROW(DUA_TOKLIN_STATS), ATTR_ROW(DUA_TOKLIN_RESIDUAL),
    ATTR_ROW(DUA_TOKLIN_SCATTER),
};
#undef ATTR_ROW
static const LdsAttrs kSwinGemmLdsReg(kSwinGemmLdsAttrs);
}  // namespace dua

extern "C" int dua_swin_mlp(long tokens, int C, const void* ln2, const void* W1, const float* b1, const void* W2, const float* b2,
                            float* x, void* stream) {
  using namespace dua;
  if (tokens <= 0 || (C != 48 && C != 96) || !ln2 || !W1 || !b1 || !W2 || !b2 || !x) return DUA_ERR_ARG;
  MlpArgs a;
  a.ln = (const f16*)ln2; a.M = tokens; a.C = C; a.hidden = 4 * C; a.W1 = (const f16*)W1; a.b1 = b1; a.W2 = (const f16*)W2; a.b2 = b2;
  a.x = x;
  a.w1_row = padded_row(C * 2); a.w2_row = padded_row(192 * 2); a.a_row = padded_row(C * 2); a.o_row = padded_row(C * 4);
  const int nb2 = (C + 31) / 32;
  const int lds = 192 * a.w1_row + nb2 * 32 * a.w2_row + 128 * (a.a_row > a.o_row ? a.a_row : a.o_row);
  if (lds > 160 * 1024) return DUA_ERR_ARG;
  if (int e = ensure_prepared()) return e;
  const long tiles = (tokens + 127) / 128;
  dim3 grid((unsigned)(tiles < 512 ? tiles : 512));
  if (C == 48) hipLaunchKernelGGL(swin_mlp_kernel<48>, grid, dim3(256), lds, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(swin_mlp_kernel<96>, grid, dim3(256), lds, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

extern "C" int dua_token_linear(const dua_token_linear_desc* d, void* stream) {
  using namespace dua;
  if (!d || !d->A || !d->W || d->M <= 0 || d->K <= 0 || d->K % 8 || d->K > 384 || d->N <= 0 || d->N % 8 || d->N > 32 * tg::MAXNB_ALL ||
      d->lda < d->K || d->lda % 8 || d->samples <= 0)
    return DUA_ERR_ARG;
  TokLinArgs a{};
  a.A = (const f16*)d->A; a.lda = d->lda; a.M = d->M; a.K = d->K; a.N = d->N; a.W = (const f16*)d->W; a.bias = d->bias;
  a.mode = d->mode; a.out = (f16*)d->out; a.ldc = d->ldc; a.out_off = d->out_off; a.x = d->x; a.stats = d->stats;
  a.c_pad = d->c_pad; a.gamma = d->gamma; a.beta = d->beta; a.eps = d->eps; a.ln_out = (f16*)d->ln_out;
  int samples = 1;
  switch (d->mode) {
    case DUA_TOKLIN_PLAIN: case DUA_TOKLIN_GELU:
      if (!d->out || d->ldc % 8 || d->out_off % 8 || d->ldc < d->out_off + d->N || d->samples != 1) return DUA_ERR_ARG;
      break;
    case DUA_TOKLIN_STATS:
      if (!d->out || !d->stats || d->N > 64 || d->c_pad < d->N || d->ldc % 8 || d->out_off % 8 || d->ldc < d->out_off + d->N) return DUA_ERR_ARG;
      samples = d->samples;
      break;
    case DUA_TOKLIN_RESIDUAL:
      if (!d->x || d->samples != 1) return DUA_ERR_ARG;
      break;
    case DUA_TOKLIN_SCATTER:
      if (!d->x || !d->ln_out || !d->gamma || !d->beta || !geom_ok(&d->geom) || d->geom.C != d->N || (d->N != 48 && d->N != 96 && d->N != 192) || d->samples != 1) return DUA_ERR_ARG;
      if (((size_t)d->gamma & 15) || ((size_t)d->beta & 15) || ((size_t)d->x & 15) || ((size_t)d->ln_out & 7)) return DUA_ERR_ARG;
      a.g = make_geom(&d->geom);
      if ((long)a.g.B * a.g.nw * a.g.n != d->M) return DUA_ERR_ARG;
      break;
    default: return DUA_ERR_ARG;
  }
  const int NB = (d->N + 31) / 32;
  const int kc = d->K < 16 * tg::KSTEPS ? d->K : 16 * tg::KSTEPS;
  const bool f32_tile = d->mode == DUA_TOKLIN_RESIDUAL || d->mode == DUA_TOKLIN_SCATTER;
  a.w_row = padded_row(d->K * 2);
  a.a_row = padded_row(((kc + 15) / 16) * 32);
  a.o_row = padded_row(d->N * (f32_tile ? 4 : 2));
  a.w_bytes = NB * 32 * a.w_row + 16;
  const int tile_bytes = 128 * (a.a_row > a.o_row ? a.a_row : a.o_row);
  const int lds = a.w_bytes + tile_bytes;
  if (lds > 160 * 1024) return DUA_ERR_ARG;
  const Kern kern = kTokLinTable[d->mode][NB - 1];
  if (int e = ensure_prepared()) return e;
  // Enough workgroups to fill every CU to its occupancy limit: a wave works through load -> MFMA -> epilogue of one tile
  // at a time, so the latency of its loads is hidden by the OTHER waves of the SIMD, not inside the wave.
  // (a pure query, cached per kernel and LDS size; the cache is shared by every thread that launches)
  static std::mutex occ_mu;
  static int occ[5][6] = {};
  static int occ_lds[5][6] = {};
  int oc;
  {
    std::lock_guard<std::mutex> lock(occ_mu);
    int& o = occ[d->mode][NB - 1];
    if (o == 0 || occ_lds[d->mode][NB - 1] != lds) {
      int nblk = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)kern, 256, lds) != hipSuccess || nblk < 1) nblk = 1;
      o = nblk > 8 ? 8 : nblk;
      occ_lds[d->mode][NB - 1] = lds;
    }
    oc = o;
  }
  const long tiles = (d->M + 127) / 128;
  long cap = 256L * (d->background > 0 && d->background < oc ? d->background : oc) / samples;      // background: that many workgroups per CU
  if (cap < 1) cap = 1;
  dim3 grid((unsigned)(tiles < cap ? tiles : cap), samples);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
