// Backward of ConvTranspose3d(kernel 2, stride 2) (MONAI UpSample "deconv", models/basic_unet/denoiser.py:161-170) for
// the training step; dOut is read straight out of the gradient of the concat buffer (channel slice, no split copy).
//
//   data gradient  : dX[v][ci] = sum_{tap, co} dOut[child(v, tap)][co] * W[ci][co][tap]
//                    GEMM M = input voxels, N = Cin, K = 8 taps x Cout; the A tile of a (tap, co chunk) step is the
//                    256 child voxels of the workgroup's 256 input voxels -- a gather of 64-byte rows.
//   weight gradient: dW[ci][co][tap] += sum_v x[v][ci] * dOut[child(v, tap)][co]
//                    GEMM per tap M = Cin, N = Cout, K = voxels (the slow axis of channels-last data): x and the eight
//                    child tiles are staged as they lie in HBM and read with ds_read_b64_tr_b16 (as conv3d_wgrad.hip);
//                    workgroup = 8 waves = 8 taps x 64 ci x 64 co, persistent over 128-voxel tiles, partial sums to a
//                    workspace + reduce kernel.  f32 (parity) form: MFMA 32x32x2, plain reads.
#include "common.hpp"
#include "../../include/dua_hip.h"

namespace dua {

namespace db {
constexpr int TM = 256, BN = 64, KG = 4;
constexpr int VS = KG * 16 + 16;          // 80 B per staged voxel (as the forward kernel)
constexpr int A_BYTES = TM * VS;
constexpr int W_BYTES = KG * BN * 16;
}  // namespace db

struct DeconvBwdArgs {
  const void* x; const void* dy; const void* w; void* dx; float* dw; float* part;
  int N, D, H, W;                  // INPUT spatial extent (dy is 2x)
  int Cin, Cin_stride, Cin_off;    // x / dx channel slice
  int Cout, Cout_stride, Cout_off; // dy channel slice
  int nchunks, nct, P, total_tiles, ncc;
};

// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void deconv_k2s2_dgrad_kernel(DeconvBwdArgs a) {
  using namespace db;
  using Frag = typename Elem<T>::Frag;
  constexpr int EPG = Elem<T>::EPG;
  constexpr int CK = KG * EPG;
  constexpr int OS = BN * (int)sizeof(T) + 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* alds = smem;
  char* wlds = smem + A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const long vox = (long)a.D * a.H * a.W;
  const long v0 = (long)blockIdx.x * TM;
  const int ct = blockIdx.y, n = blockIdx.z;
  const int H2 = 2 * a.H, W2 = 2 * a.W;
  const T* dyb = (const T*)a.dy + (long)n * vox * 8 * a.Cout_stride + a.Cout_off;
  // packed weights: [tap][ct over Cin][chunk over Cout][kg][64 ci][EPG co]
  const char* wsrc = (const char*)a.w + (long)ct * a.nchunks * W_BYTES;
  const long wtap = (long)a.nct * a.nchunks * W_BYTES;

  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][q][i] = 0.f;

  const int kg_t = tid & 3;
  long child0[4];                       // child voxel of tap (0,0,0) for this thread's four staging rows, -1 = out of range
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long v = v0 + (tid >> 2) + 64 * j;
    if (v < vox) {
      const int w = (int)(v % a.W); const long t = v / a.W;
      const int h = (int)(t % a.H), d = (int)(t / a.H);
      child0[j] = ((long)(2 * d) * H2 + 2 * h) * W2 + 2 * w;
    } else child0[j] = -1;
  }
  const int steps = 8 * a.nchunks;
  Frag pa[4];
  f32x4 pw;
  auto load_step = [&](int it) {
    const int tap = it / a.nchunks, ch = it % a.nchunks;
    const long toff = ((long)(tap >> 2) * H2 + ((tap >> 1) & 1)) * W2 + (tap & 1);
    const int c0 = ch * CK + kg_t * EPG;
    const bool cok = c0 < a.Cout;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (child0[j] >= 0 && cok) pa[j] = *(const Frag*)(dyb + (child0[j] + toff) * a.Cout_stride + c0);
      else
#pragma unroll
        for (int e = 0; e < EPG; ++e) pa[j][e] = (T)0.f;
    }
    pw = *(const f32x4*)(wsrc + tap * wtap + (long)ch * W_BYTES + tid * 16);
  };
  load_step(0);
  for (int it = 0; it < steps; ++it) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) *(Frag*)(alds + ((tid >> 2) + 64 * j) * VS + kg_t * 16) = pa[j];
    *(f32x4*)(wlds + tid * 16) = pw;
    __syncthreads();
    if (it + 1 < steps) load_step(it + 1);
#pragma unroll
    for (int ks = 0; ks < KG / 2; ++ks) {
      Frag a0 = *(const Frag*)(alds + (wave * 64 + r) * VS + (2 * ks + hh) * 16);
      Frag a1 = *(const Frag*)(alds + (wave * 64 + 32 + r) * VS + (2 * ks + hh) * 16);
      Frag b0 = *(const Frag*)(wlds + ((2 * ks + hh) * BN + r) * 16);
      Frag b1 = *(const Frag*)(wlds + ((2 * ks + hh) * BN + 32 + r) * 16);
      mma32(acc[0][0], a0, b0);
      mma32(acc[0][1], a0, b1);
      mma32(acc[1][0], a1, b0);
      mma32(acc[1][1], a1, b1);
    }
  }
  __syncthreads();
  char* ot = smem + wave * 64 * OS;
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        *(T*)(ot + (m * 32 + acc_row(i, hh)) * OS + (q * 32 + r) * (int)sizeof(T)) = (T)acc[m][q][i];
  __syncthreads();
  constexpr int GPV = BN / EPG, VPI = 64 / GPV;
  T* xout = (T*)a.dx + (long)n * vox * a.Cin_stride + a.Cin_off + ct * BN;
#pragma unroll
  for (int it = 0; it < 64 / VPI; ++it) {
    const int vl = it * VPI + lane / GPV, cg = lane % GPV;
    const long v = v0 + wave * 64 + vl;
    if (v < vox && ct * BN + cg * EPG < a.Cin)
      *(Frag*)(xout + v * a.Cin_stride + cg * EPG) = *(const Frag*)(ot + vl * OS + cg * 16);
  }
}

// ---------------------------------------------------------------------------------------------------------------
namespace dwg {
constexpr int NT = 256;                 // 4 waves: each contracts a quarter of a tile's voxels into its own 64 x 64 accumulators
template <typename T> constexpr int tile_voxels() { return sizeof(T) == 2 ? 128 : 64; }   // 8 k-steps of 16 (f16) / 8 (f32) voxels
typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
}

// Weight gradient of the transposed convolution: eight independent GEMMs dW_tap[ci][co] = sum_v x[v][ci] * dy[2v + tap][co].
// One workgroup = one tap, one (64 ci x 64 co) tile, one partition of the input voxels; it walks its partition in tiles of
// TV voxels: the x tile and the tap's dy tile are staged in LDS exactly as they lie in HBM (rows of 32 channels) and read
// with the transposing LDS read (conv3d_wgrad.hip); the NEXT tile's 16-byte pieces are already in registers while the current
// one is multiplied.  Wave w contracts k-steps w, w + 4: the four 64 x 64 partial results are summed through LDS at the end and
// written as one [ci][co] tile of the partial-sum buffer, which deconv_wgrad_reduce_kernel adds over the partitions.
// (The first version gave a workgroup all eight taps of a tile -- 147 KB of LDS, one workgroup per CU, no prefetch, 768
// partitions to fill the chip: 920 us + 190 us of reduction for the 48^3 -> 96^3 layer at batch 2, whose operands are 254 MB.)
template <typename T>
__global__ __launch_bounds__(dwg::NT) void deconv_k2s2_wgrad_kernel(DeconvBwdArgs a) {
  using namespace dwg;
  using Frag = typename Elem<T>::Frag;
  constexpr int EPG = Elem<T>::EPG;
  constexpr int TV = tile_voxels<T>();
  constexpr int G = 64 / EPG;                      // 16-byte pieces per voxel
  constexpr int RSB = 32 * (int)sizeof(T);
  constexpr int IMG = TV * RSB + 64;               // one 32-channel half image of one tile; +64 B: the two halves of a voxel
                                                   // land in different halves of the 32 store banks (as conv3d_wgrad.hip)
  constexpr int KV = sizeof(T) == 2 ? 16 : 8;      // voxels per mma32 call
  constexpr int NI = TV * G / NT;                  // pieces per thread and operand (4)
  static_assert(TV * G % NT == 0 && (TV / KV) % 4 == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xs = smem;                                 // [2][TV][32]
  char* Ys = smem + 2 * IMG;                       // [2][TV][32]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hl = lane >> 5;
  const int part_id = blockIdx.x, tap = blockIdx.y & 7, combo = blockIdx.y >> 3;
  const int ci_t = combo / a.ncc, co_t = combo % a.ncc;          // ncc = co tiles
  const int vox = a.D * a.H * a.W;                 // the launcher checks that 8 * vox fits an int
  const int H2 = 2 * a.H, W2 = 2 * a.W;
  const int toff = ((tap >> 2) * H2 + ((tap >> 1) & 1)) * W2 + (tap & 1);
  const int tiles_per_n = (vox + TV - 1) / TV;

  f32x16 acc[2][2];                                // [ci half][co half]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  int a_off[2], b_col;
  if constexpr (sizeof(T) == 2) {
    const int i16 = lane & 15, q = i16 >> 2, p = i16 & 3, g1 = (lane >> 4) & 1;
    b_col = (16 * g1 + 4 * p) * 2;
    a_off[0] = (8 * hl + q) * RSB + b_col;
    a_off[1] = (8 * hl + 4 + q) * RSB + b_col;
  } else {
    b_col = (lane & 31) * 4;
    a_off[0] = a_off[1] = 0;
  }
  // this thread's pieces: voxel vl[j] of the tile, 16-byte group g (the same for all j: NT is a multiple of G)
  const int g = tid % G, vl0 = tid / G;
  constexpr int VSTEP = NT / G;
  const int lds_off = ((g * EPG) >> 5) * IMG + ((g * EPG) & 31) * (int)sizeof(T);
  const bool xg_ok = ci_t * 64 + g * EPG < a.Cin, yg_ok = co_t * 64 + g * EPG < a.Cout;

  Frag xr[NI], yr[NI];
  auto load_tile = [&](int tile) {
    const int n = tile / tiles_per_n, v0 = (tile - n * tiles_per_n) * TV;
    // channel groups behind Cin / Cout are zero-filled below: their loads go to the slice's first group (a real address)
    const T* xb = (const T*)a.x + (long)n * vox * a.Cin_stride + a.Cin_off + (xg_ok ? ci_t * 64 + g * EPG : 0);
    const T* yb = (const T*)a.dy + (long)n * vox * 8 * a.Cout_stride + a.Cout_off + (yg_ok ? co_t * 64 + g * EPG : 0);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int v = v0 + vl0 + VSTEP * j;
      const bool in = v < vox;
      const int vc = in ? v : 0;
      const int w = vc % a.W, t = vc / a.W, h = t % a.H, d = t / a.H;
      const long ov = (long)((2 * d) * H2 + 2 * h) * W2 + 2 * w + toff;
      xr[j] = *(const Frag*)(xb + (long)vc * a.Cin_stride);             // unconditional loads on clamped addresses
      yr[j] = *(const Frag*)(yb + ov * a.Cout_stride);
      if (!(in && xg_ok)) {
#pragma unroll
        for (int e = 0; e < EPG; ++e) xr[j][e] = (T)0.f;
      }
      if (!(in && yg_ok)) {
#pragma unroll
        for (int e = 0; e < EPG; ++e) yr[j][e] = (T)0.f;
      }
    }
  };

  int tile = part_id;
  if (tile < a.total_tiles) load_tile(tile);
  for (; tile < a.total_tiles; tile += a.P) {
    __syncthreads();                               // the previous tile's fragment reads are done
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      *(Frag*)(Xs + lds_off + (vl0 + VSTEP * j) * RSB) = xr[j];
      *(Frag*)(Ys + lds_off + (vl0 + VSTEP * j) * RSB) = yr[j];
    }
    __syncthreads();
    if (tile + a.P < a.total_tiles) load_tile(tile + a.P);
#pragma unroll
    for (int ss = 0; ss < TV / KV / 4; ++ss) {
      const int s = wave + 4 * ss;
      Frag fx[2], fy[2];
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          h4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(Xs + hf * IMG + s * 16 * RSB + a_off[0]));
          h4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(Xs + hf * IMG + s * 16 * RSB + a_off[1]));
          fx[hf] = __builtin_shufflevector(__builtin_bit_cast(f16x4, lo), __builtin_bit_cast(f16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
          h4 lo2 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(Ys + hf * IMG + s * 16 * RSB + a_off[0]));
          h4 hi2 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(Ys + hf * IMG + s * 16 * RSB + a_off[1]));
          fy[hf] = __builtin_shufflevector(__builtin_bit_cast(f16x4, lo2), __builtin_bit_cast(f16x4, hi2), 0, 1, 2, 3, 4, 5, 6, 7);
        }
      } else {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            fx[hf][e] = *(const float*)(Xs + hf * IMG + (s * 8 + 2 * e + hl) * RSB + b_col);
            fy[hf][e] = *(const float*)(Ys + hf * IMG + (s * 8 + 2 * e + hl) * RSB + b_col);
          }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma32(acc[i][j], fx[i], fy[j]);      // rows = ci, columns = co
    }
  }
  // sum the four waves' 64 x 64 results through LDS in two halving rounds (waves 2, 3 -> 0, 1; wave 1 -> 0: 32 KB, inside the
  // tile buffers), then wave 0 writes its tile of part[P][combo][tap][ci 64][co 64]; lane column = co (lane & 31),
  // register e -> ci row
  float* red = (float*)smem + (wave & 1) * 4096 + (lane & 31);
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    const bool writer = round == 0 ? wave >= 2 : wave == 1, reader = round == 0 ? wave < 2 : wave == 0;
    __syncthreads();
    if (writer) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) red[(i * 32 + acc_row(e, hl)) * 64 + j * 32] = acc[i][j][e];
    }
    __syncthreads();
    if (reader) {
      const float* rd = round == 0 ? red : (const float*)smem + 4096 + (lane & 31);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] += rd[(i * 32 + acc_row(e, hl)) * 64 + j * 32];
    }
  }
  if (wave != 0) return;
  float* pp = a.part + (((long)part_id * (gridDim.y >> 3) + combo) * 8 + tap) * 4096 + (lane & 31);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) pp[(i * 32 + acc_row(e, hl)) * 64 + j * 32] = acc[i][j][e];
}

// dw[ci][co][tap] += sum_p part[p][combo][tap][ci][co]
__global__ __launch_bounds__(256) void deconv_wgrad_reduce_kernel(const float* __restrict__ part, int P, int ncc, int ncombo,
                                                                  int Cin, int Cout, float* __restrict__ dw) {
  const long per_p = (long)ncombo * 8 * 4096;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < per_p; i += (long)gridDim.x * 256) {
    const int col = (int)(i & 63), cil = (int)((i >> 6) & 63);
    const int tap = (int)((i >> 12) & 7), combo = (int)(i >> 15);
    const int ci = (combo / ncc) * 64 + cil, co = (combo % ncc) * 64 + col;
    if (ci >= Cin || co >= Cout) continue;
    float s0 = 0.f, s1 = 0.f;
    int p = 0;
    for (; p + 1 < P; p += 2) { s0 += part[(long)p * per_p + i]; s1 += part[(long)(p + 1) * per_p + i]; }
    if (p < P) s0 += part[(long)p * per_p + i];
    dw[((long)ci * Cout + co) * 8 + tap] += s0 + s1;
  }
}

static inline int deconv_wgrad_partitions(const dua_conv3_desc* d, int* ncombo) {
  const int combos = ((d->Cin + 63) / 64) * ((d->Cout + 63) / 64);
  const long vox = (long)d->D * d->H * d->W;
  const int tv = d->dtype == DUA_F16 ? dwg::tile_voxels<f16>() : dwg::tile_voxels<float>();
  const long total = d->N * ((vox + tv - 1) / tv);
  long P = (3 * 256 + 8 * combos - 1) / (8 * combos);      // ~three workgroups per CU over the launch: (partition, tap, combo)
  if (P > total) P = total;
  if (ncombo) *ncombo = combos;
  return (int)(P < 1 ? 1 : P);
}

// dynamic-LDS limits (raised once per device by ensure_prepared(), common.hpp)
template <typename T> constexpr int deconv_dgrad_lds() {
  constexpr int OS = db::BN * (int)sizeof(T) + 16;
  return (db::TM * OS > db::A_BYTES + db::W_BYTES) ? db::TM * OS : db::A_BYTES + db::W_BYTES;
}
template <typename T> constexpr int deconv_wgrad_lds() { return 4 * (dwg::tile_voxels<T>() * 32 * (int)sizeof(T) + 64); }
static const LdsAttr kDeconvBwdLdsAttrs[] = {
    {(const void*)deconv_k2s2_dgrad_kernel<f16>, deconv_dgrad_lds<f16>()},
    {(const void*)deconv_k2s2_dgrad_kernel<float>, deconv_dgrad_lds<float>()},
    {(const void*)deconv_k2s2_wgrad_kernel<f16>, deconv_wgrad_lds<f16>()},
    {(const void*)deconv_k2s2_wgrad_kernel<float>, deconv_wgrad_lds<float>()},
};
static const LdsAttrs kDeconvBwdLdsReg(kDeconvBwdLdsAttrs);

template <typename T>
static int launch_deconv_bwd(const dua_conv3_desc* d, const void* x, const void* dy, const void* w_packed, void* dx,
                             float* dw, float* ws, long ws_bytes, hipStream_t s) {
  constexpr int CK = db::KG * Elem<T>::EPG;
  DeconvBwdArgs a{};
  a.x = x; a.dy = dy; a.w = w_packed; a.dx = dx; a.dw = dw; a.part = ws;
  a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W;
  a.Cin = d->Cin; a.Cin_stride = d->Cin_stride; a.Cin_off = d->Cin_off;
  a.Cout = d->Cout; a.Cout_stride = d->Cout_stride; a.Cout_off = d->Cout_off;
  const long vox = (long)d->D * d->H * d->W;
  if (int e = ensure_prepared()) return e;
  if (dx) {
    a.nchunks = (d->Cout + CK - 1) / CK;             // K chunks run over Cout
    a.nct = (d->Cin + db::BN - 1) / db::BN;          // output tiles over Cin
    constexpr int LDS = deconv_dgrad_lds<T>();
    dim3 grid((unsigned)((vox + db::TM - 1) / db::TM), a.nct, d->N);
    hipLaunchKernelGGL(deconv_k2s2_dgrad_kernel<T>, grid, dim3(256), LDS, s, a);
  }
  if (dw) {
    int ncombo;
    const int P = deconv_wgrad_partitions(d, &ncombo);
    if (!ws || (long)P * ncombo * 8 * 4096 * 4 > ws_bytes) return DUA_ERR_ARG;
    a.P = P; a.ncc = (d->Cout + 63) / 64;
    constexpr int TV = dwg::tile_voxels<T>();
    a.total_tiles = (int)(d->N * ((vox + TV - 1) / TV));
    if (vox * 8 > 0x7fffffffL) return DUA_ERR_ARG;                      // the kernel indexes voxels with ints
    const int lds = 4 * (TV * 32 * (int)sizeof(T) + 64);                // x and dy tile, two half images each (>= 32 KB: the
                                                                        // end-of-kernel reduction reuses it)
    hipLaunchKernelGGL(deconv_k2s2_wgrad_kernel<T>, dim3(P, 8 * ncombo), dim3(dwg::NT), lds, s, a);
    const long per_p = (long)ncombo * 8 * 4096;
    long nb = (per_p + 255) / 256;
    hipLaunchKernelGGL(deconv_wgrad_reduce_kernel, dim3((unsigned)(nb > 4096 ? 4096 : nb)), dim3(256), 0, s, ws, P, a.ncc, ncombo,
                       d->Cin, d->Cout, dw);
  }
  return (int)hipGetLastError();
}

// ---- weights for the data gradient: [tap][ct over Cin][chunk over Cout][kg][64 ci][EPG co] from w[ci][co][tap] ----
template <typename T>
__global__ void pack_deconv_dgrad_kernel(int Cin, int Cout, int nchunks, int nct, const float* __restrict__ w,
                                         T* __restrict__ out, long total) {
  constexpr int EPG = Elem<T>::EPG;
  constexpr int CK = 4 * EPG;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long t = i;
    const int e = (int)(t % EPG); t /= EPG;
    const int ci_l = (int)(t % 64); t /= 64;
    const int kg = (int)(t % 4); t /= 4;
    const int ch = (int)(t % nchunks); t /= nchunks;
    const int ct = (int)(t % nct); const int tap = (int)(t / nct);
    const int ci = ct * 64 + ci_l, co = ch * CK + kg * EPG + e;
    float v = 0.f;
    if (co < Cout && ci < Cin) v = w[((long)ci * Cout + co) * 8 + tap];
    out[i] = (T)v;
  }
}

}  // namespace dua

extern "C" {

long dua_pack_deconv_weights_dgrad(int dtype, int Cin, int Cout, const float* w, void* w_packed, void* stream) {
  const int epg = dtype == DUA_F16 ? 8 : 4, ck = 4 * epg;
  if ((dtype != DUA_F16 && dtype != DUA_F32) || Cout <= 0 || Cin <= 0) return DUA_ERR_ARG;
  const int nchunks = (Cout + ck - 1) / ck, nct = (Cin + 63) / 64;
  const long total = 8L * nct * nchunks * 4 * 64 * epg;
  const long bytes = total * (dtype == DUA_F16 ? 2 : 4);
  if (!w_packed) return bytes;
  if (!w) return DUA_ERR_ARG;
  long b = (total + 255) / 256;
  dim3 grid((unsigned)(b > 16384 ? 16384 : b));
  if (dtype == DUA_F16)
    hipLaunchKernelGGL(dua::pack_deconv_dgrad_kernel<dua::f16>, grid, dim3(256), 0, (hipStream_t)stream, Cin, Cout, nchunks,
                       nct, w, (dua::f16*)w_packed, total);
  else
    hipLaunchKernelGGL(dua::pack_deconv_dgrad_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, Cin, Cout, nchunks, nct,
                       w, (float*)w_packed, total);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? bytes : -(long)e;
}

long dua_deconv_k2s2_bwd_workspace(const dua_conv3_desc* d) {
  if (!d) return DUA_ERR_ARG;
  int ncombo;
  const int P = dua::deconv_wgrad_partitions(d, &ncombo);
  return (long)P * ncombo * 8 * 4096 * 4;
}

int dua_deconv_k2s2_bwd(const dua_conv3_desc* d, const void* x, const void* dy, const void* w_packed_dgrad, void* dx,
                        float* dw, void* workspace, long workspace_bytes, void* stream) {
  if (!d || !dy || (!dx && !dw) || (dx && !w_packed_dgrad) || (dw && !x)) return DUA_ERR_ARG;
  if (d->Cin % 8 || d->Cout % 8 || d->Cin_stride % 8 || d->Cout_stride % 8 || d->Cin_off % 8 || d->Cout_off % 8) return DUA_ERR_ARG;
  if (d->dtype == DUA_F16)
    return dua::launch_deconv_bwd<dua::f16>(d, x, dy, w_packed_dgrad, dx, dw, (float*)workspace, workspace_bytes, (hipStream_t)stream);
  if (d->dtype == DUA_F32)
    return dua::launch_deconv_bwd<float>(d, x, dy, w_packed_dgrad, dx, dw, (float*)workspace, workspace_bytes, (hipStream_t)stream);
  return DUA_ERR_ARG;
}

}  // extern "C"
