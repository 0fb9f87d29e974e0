// Token GEMMs of the COARSE Swin stages and of the channel-changing UnetResBlocks below 48^3 (BASELINE config 5):
//   out[token][n] = sum_k A[token][k] * W[n][k] (+ bias[n])     M = 343 .. 21 952 tokens, K = 96 .. 3072, N = 96 .. 1536
// = nn.Linear of WindowAttention (qkv, proj: models/swin_unetr/attention.py:91-94,99,118), of the MLP (MONAI MLPBlock
// linear1 / linear2, transformer.py:376,433-434) and of PatchMerging (reduction, patch.py:89-92) at stages 1-3, and the
// 1x1x1 conv3 of UnetResBlock (blocks.py:286-296, 311-314) where it is wider than the fine-stage kernel (swin_gemm.hip:
// all weights resident in LDS, N <= 192, K <= 384) can hold.  Round 2 ran these ~60 launches per step on hipBLASLt.
//
// A plain tiled MFMA GEMM sized for these shapes (a few GFLOP at most, latency bound): workgroup = 64 tokens x 64 output
// channels, 4 waves as 2 x 2 MFMA 32x32x16 accumulators; both operands stream through LDS in 64-deep K steps (rows padded
// to 144 B: an odd number of 16-byte units, so the 16 lanes of a ds_read_b128 phase hit distinct bank groups), the next K
// step's 16-byte pieces are in registers while the current one multiplies.  The WEIGHTS are the MFMA A operand (rows =
// output channels), the tokens the B operand: a lane then holds four consecutive channels of ONE token per register quad,
// so bias, GELU and the residual add are register arithmetic, and the tile leaves through LDS as coalesced 16-byte rows.
//   PLAIN     + bias                      -> fp16 out[token][out_off + n]
//   GELU      exact GELU(acc + bias)      -> fp16 (MLPBlock act between linear1 and linear2: the separate GELU pass is gone)
//   RESIDUAL  x[token][n] += acc + bias   on the fp32 token stream
#include "common.hpp"
#include "../../include/dua_hip.h"

namespace dua {

namespace wg_ {
constexpr int BM = 64, BN = 64, BK = 64;
constexpr int ROW = BK * 2 + 16;                 // 144 B per staged row
constexpr int TILE = 64 * ROW;                   // one operand tile: 9216 B
constexpr int OROW32 = BN * 4 + 16, OROW16 = BN * 2 + 16;
}  // namespace wg_

struct GemmArgs {
  const f16* A; int lda; long M; int K, N;
  const f16* W; const float* bias;
  f16* out; int ldc, out_off;
  float* x; int ldx;
  float* part; int kper, ksplit;        // split K: slice z contracts [z * kper, (z + 1) * kper) into part[z][M][N] (fp32)
};

// SPLIT: the small-token layers (27 .. 343 tokens against K up to 3072: a few dozen tiles, each walking 24 - 48 dependent K
// steps of ~1 us) divide K over gridDim.z; fp32 partial tiles go to a workspace and token_gemm_finish_kernel applies the
// epilogue (52.8 us -> ~8 us for the 27-token x 3072 -> 768 reduction of the last patch merging).
template <int MODE, bool SPLIT = false>
__global__ __launch_bounds__(256) void token_gemm_kernel(GemmArgs a) {
  using namespace wg_;
  __shared__ __attribute__((aligned(16))) char smem[2 * TILE > 64 * OROW32 ? 2 * TILE : 64 * OROW32];
  char* As = smem;                 // [64 tokens][ROW]
  char* Ws = smem + TILE;          // [64 channels][ROW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int wm = wave >> 1, wn = wave & 1;                       // this wave's 32-token block and 32-channel block
  // staging: thread t moves rows (t >> 3) and (t >> 3) + 32 of each operand, 16-byte piece (t & 7) of the 128-byte K step
  const int srow = tid >> 3, spc = tid & 7;
  f16x8 ra[2], rw[2];
  f16x8 z8;
#pragma unroll
  for (int e = 0; e < 8; ++e) z8[e] = (f16)0.f;
  auto load_step = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long m = m0 + srow + 32 * j;
      const int n = n0 + srow + 32 * j, k = k0 + spc * 8;
      // clamped addresses, masked use: branch-free loads
      const f16x8 va = *(const f16x8*)(a.A + (m < a.M ? m : a.M - 1) * a.lda + (k < a.K ? k : 0));
      const f16x8 vw = *(const f16x8*)(a.W + (long)(n < a.N ? n : a.N - 1) * a.K + (k < a.K ? k : 0));
      ra[j] = (m < a.M && k < a.K) ? va : z8;
      rw[j] = (n < a.N && k < a.K) ? vw : z8;
    }
  };
  auto store_step = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      *(f16x8*)(As + (srow + 32 * j) * ROW + spc * 16) = ra[j];
      *(f16x8*)(Ws + (srow + 32 * j) * ROW + spc * 16) = rw[j];
    }
  };
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int kbeg = SPLIT ? blockIdx.z * a.kper : 0;
  const int kend = SPLIT ? (kbeg + a.kper < a.K ? kbeg + a.kper : a.K) : a.K;
  load_step(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();                        // everyone is done reading the previous step's tiles
    store_step();
    __syncthreads();
    if (k0 + BK < kend) load_step(k0 + BK); // in flight while this step multiplies
    const char* wrow = Ws + (wn * 32 + r) * ROW + hh * 16;
    const char* arow = As + (wm * 32 + r) * ROW + hh * 16;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const f16x8 fw = *(const f16x8*)(wrow + ks * 32);
      const f16x8 fa = *(const f16x8*)(arow + ks * 32);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw, fa, acc, 0, 0, 0);      // rows = channels, columns = tokens
    }
  }
  __syncthreads();
  if constexpr (SPLIT) {
    // fp32 partial tile, as it stands, through LDS for 16-byte rows
    char* ot = smem;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *(f32x4*)(ot + (wm * 32 + r) * OROW32 + (wn * 32 + 8 * j + 4 * hh) * 4) = f32x4{acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]};
    __syncthreads();
    float* pz = a.part + (long)blockIdx.z * a.M * a.N;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 16 + (tid >> 4), c4 = tid & 15;
      const long m = m0 + row;
      const int n = n0 + c4 * 4;
      if (m < a.M && n < a.N) *(f32x4*)(pz + m * a.N + n) = *(const f32x4*)(ot + row * OROW32 + c4 * 16);
    }
    return;
  }
  // ---- epilogue: register quad j of a lane = channels n0 + wn*32 + 8j + 4hh + (0..3) of token m0 + wm*32 + r ----
  float v[16];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = n0 + wn * 32 + 8 * j + 4 * hh + e;
      float t = acc[4 * j + e] + ((a.bias && n < a.N) ? a.bias[n] : 0.f);
      if (MODE == DUA_TOKLIN_GELU) t = gelu_erf(t);
      v[4 * j + e] = t;
    }
  if (MODE == DUA_TOKLIN_RESIDUAL) {
    // through LDS as fp32 rows, then x += with 16-byte accesses
    char* ot = smem;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *(f32x4*)(ot + (wm * 32 + r) * OROW32 + (wn * 32 + 8 * j + 4 * hh) * 4) = f32x4{v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]};
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 16 + (tid >> 4), c4 = tid & 15;
      const long m = m0 + row;
      const int n = n0 + c4 * 4;
      if (m < a.M && n < a.N) {
        float* xp = a.x + m * a.ldx + n;
        f32x4 xv = *(const f32x4*)xp;
        const f32x4 o = *(const f32x4*)(ot + row * OROW32 + c4 * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[e] += o[e];
        *(f32x4*)xp = xv;
      }
    }
  } else {
    char* ot = smem;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *(f16x4*)(ot + (wm * 32 + r) * OROW16 + (wn * 32 + 8 * j + 4 * hh) * 2) =
          f16x4{(f16)v[4 * j], (f16)v[4 * j + 1], (f16)v[4 * j + 2], (f16)v[4 * j + 3]};
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = it * 32 + (tid >> 3), c8 = tid & 7;
      const long m = m0 + row;
      const int n = n0 + c8 * 8;
      if (m < a.M && n < a.N) *(f16x8*)(a.out + m * a.ldc + a.out_off + n) = *(const f16x8*)(ot + row * OROW16 + c8 * 16);
    }
  }
}

// sum of the K slices + bias, then the epilogue of MODE; one thread = four consecutive channels of one token
template <int MODE>
__global__ __launch_bounds__(256) void token_gemm_finish_kernel(GemmArgs a) {
  const long q = blockIdx.x * 256L + threadIdx.x, nq = a.N / 4;
  if (q >= a.M * nq) return;
  const long m = q / nq;
  const int n = (int)(q - m * nq) * 4;
  f32x4 pv[16];
#pragma unroll
  for (int z = 0; z < 16; ++z) {
    const int zz = z < a.ksplit ? z : a.ksplit - 1;                  // clamped address, masked use
    pv[z] = *(const f32x4*)(a.part + ((long)zz * a.M + m) * a.N + n);
  }
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = a.bias ? a.bias[n + e] : 0.f;
#pragma unroll
  for (int z = 0; z < 16; ++z)
    if (z < a.ksplit)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += pv[z][e];
  if (MODE == DUA_TOKLIN_RESIDUAL) {
    float* xp = a.x + m * a.ldx + n;
    f32x4 xv = *(const f32x4*)xp;
#pragma unroll
    for (int e = 0; e < 4; ++e) xv[e] += v[e];
    *(f32x4*)xp = xv;
  } else {
    if (MODE == DUA_TOKLIN_GELU)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
    *(f16x4*)(a.out + m * a.ldc + a.out_off + n) = f16x4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
  }
}

}  // namespace dua

// K slices a launch of dua_token_gemm would use for (M, K, N), and the workspace bytes that enables them (0: no split)
extern "C" long dua_token_gemm_workspace(long M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0) return DUA_ERR_ARG;
  const long tiles = ((M + 63) / 64) * ((N + 63) / 64);
  const int ksteps = (K + 63) / 64;
  if (tiles >= 128 || ksteps < 8) return 0;
  long z = 256 / tiles;
  if (z > ksteps / 2) z = ksteps / 2;
  if (z > 16) z = 16;
  if (z < 2) return 0;
  return z * M * N * 4;
}

extern "C" int dua_token_gemm(const dua_token_linear_desc* d, void* workspace, long workspace_bytes, void* stream) {
  using namespace dua;
  if (!d || !d->A || !d->W || d->M <= 0 || d->K <= 0 || d->K % 8 || d->N <= 0 || d->N % 8 || d->lda < d->K || d->lda % 8) return DUA_ERR_ARG;
  if (d->M > 65535L * 64 || d->N > 65535 * 64) return DUA_ERR_ARG;
  GemmArgs a{};
  a.A = (const f16*)d->A; a.lda = d->lda; a.M = d->M; a.K = d->K; a.N = d->N; a.W = (const f16*)d->W; a.bias = d->bias;
  a.out = (f16*)d->out; a.ldc = d->ldc; a.out_off = d->out_off; a.x = d->x; a.ldx = d->N;
  a.part = nullptr; a.kper = d->K; a.ksplit = 1;
  dim3 grid((unsigned)((d->M + 63) / 64), (unsigned)((d->N + 63) / 64));
  const long need = dua_token_gemm_workspace(d->M, d->K, d->N);
  if (need > 0 && workspace && workspace_bytes >= need && d->N % 4 == 0 &&
      (d->mode == DUA_TOKLIN_PLAIN || d->mode == DUA_TOKLIN_GELU || d->mode == DUA_TOKLIN_RESIDUAL)) {
    if (d->mode != DUA_TOKLIN_RESIDUAL && (!d->out || d->ldc % 8 || d->out_off % 8 || d->ldc < d->out_off + d->N)) return DUA_ERR_ARG;
    if (d->mode == DUA_TOKLIN_RESIDUAL && !d->x) return DUA_ERR_ARG;
    const int z = (int)(need / (d->M * d->N * 4));
    const int ksteps = (d->K + 63) / 64;
    a.part = (float*)workspace; a.ksplit = z; a.kper = ((ksteps + z - 1) / z) * 64;
    a.ksplit = (d->K + a.kper - 1) / a.kper;                          // slices that actually hold work
    grid.z = a.ksplit;
    hipLaunchKernelGGL((token_gemm_kernel<DUA_TOKLIN_PLAIN, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
    const long quads = d->M * (d->N / 4);
    dim3 fg((unsigned)((quads + 255) / 256));
    if (d->mode == DUA_TOKLIN_PLAIN) hipLaunchKernelGGL(token_gemm_finish_kernel<DUA_TOKLIN_PLAIN>, fg, dim3(256), 0, (hipStream_t)stream, a);
    else if (d->mode == DUA_TOKLIN_GELU) hipLaunchKernelGGL(token_gemm_finish_kernel<DUA_TOKLIN_GELU>, fg, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(token_gemm_finish_kernel<DUA_TOKLIN_RESIDUAL>, fg, dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
  }
  switch (d->mode) {
    case DUA_TOKLIN_PLAIN:
    case DUA_TOKLIN_GELU:
      if (!d->out || d->ldc % 8 || d->out_off % 8 || d->ldc < d->out_off + d->N) return DUA_ERR_ARG;
      if (d->mode == DUA_TOKLIN_PLAIN) hipLaunchKernelGGL(token_gemm_kernel<DUA_TOKLIN_PLAIN>, grid, dim3(256), 0, (hipStream_t)stream, a);
      else hipLaunchKernelGGL(token_gemm_kernel<DUA_TOKLIN_GELU>, grid, dim3(256), 0, (hipStream_t)stream, a);
      break;
    case DUA_TOKLIN_RESIDUAL:
      if (!d->x || d->N % 4) return DUA_ERR_ARG;
      hipLaunchKernelGGL(token_gemm_kernel<DUA_TOKLIN_RESIDUAL>, grid, dim3(256), 0, (hipStream_t)stream, a);
      break;
    default:
      return DUA_ERR_ARG;
  }
  return (int)hipGetLastError();
}
