// 3x3x3 (pad 1, stride 1) 3-D convolution as implicit GEMM on the gfx950 matrix cores.
//
// Replaces every `Conv3d k3 p1` the reference reaches through MONAI's Convolution block
// (models/basic_unet/denoiser.py:56-59, models/basic_unet/pretrained/basic_unet.py:60-63) and
// fuses around it:
//   prologue  : InstanceNorm3d(affine) + LeakyReLU(0.1) (+ timestep-embedding bias) of the
//               PRODUCER layer, applied while the halo tile is staged (denoiser.py:63-67);
//               out-of-volume halo voxels stay literal zeros (padding follows the activation)
//   epilogue  : + bias, per-(n, c) InstanceNorm statistics of THIS layer's output (per-wave sum and
//               centred second moment, combined per workgroup, then two fp64 atomics per channel
//               into one of 8 replica rows), coalesced 16-byte stores of the raw output.
//
// GEMM view: M = output voxels, N = Cout, K = 27 taps x Cin.
// Workgroup (256 threads = 4 waves): a 4x8x8 output tile x 64 output channels; wave w owns
// depth slice w (64 voxels) as 2x2 MFMA 32x32 accumulators.  Per Cin chunk of 64 bytes/voxel
// the 6x10x10 halo tile is staged once in LDS; the 9 taps of one kd plane of the packed
// weights follow it slab by slab.  A and B fragments are 16-byte ds_read_b128; the halo row
// stride (656 B) and the 4x8 row->voxel map make every fragment read bank-conflict free.
#include "common.hpp"
#include "../../include/dua_hip.h"

namespace dua {

namespace c3 {
constexpr int TD = 4, TH = 8, TW = 8;
constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;
constexpr int KG = 4;                      // k-groups (16 B) per chunk
constexpr int VS = KG * 16;                // 64 B per halo voxel per chunk
constexpr int RS = HW * VS + 16;           // 656: halo row stride, padded (bank-conflict free)
constexpr int PS = HH * RS;                // 6560: halo plane stride
constexpr int HALO_BYTES = HD * PS;        // 39360
constexpr int BN = 64;                     // output channels per workgroup
constexpr int WSLAB = 9 * KG * BN * 16;    // 36864: packed weights of one (chunk, kd)
constexpr int LDS_BYTES = HALO_BYTES + WSLAB;  // 76224 (+ 12 B per input channel when the prologue is fused)
constexpr int NITEMS = HD * HH * HW * KG;  // 2400 16-byte items per halo chunk
constexpr int NIT = (NITEMS + 255) / 256;  // 10
}  // namespace c3

struct Conv3Args {
  const void* x; const void* w; const float* bias; void* y;
  double* stats;
  InXform xf;
  int N, D, H, W;
  int Cin, Cin_stride, Cin_off;     // valid input channels, buffer stride, offset (elements)
  int Cout, Cout_stride, Cout_off;
  int nchunks, ntiles, tiles_h, tiles_w, cout_pad;
};

template <typename T>
__global__ __launch_bounds__(256, 2) void conv3d_k3_kernel(Conv3Args a) {
  using namespace c3;
  using Frag = typename Elem<T>::Frag;
  constexpr int EPG = Elem<T>::EPG;
  constexpr int CK = KG * EPG;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* wlds = smem + HALO_BYTES;
  float* xsc = (float*)(smem + LDS_BYTES);     // scale / shift / add of the fused input transform
  float* xsh = xsc + a.nchunks * CK;
  float* xad = xsh + a.nchunks * CK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int tile = xcd_remap(blockIdx.x, a.ntiles);
  const int ct = blockIdx.y, n = blockIdx.z;
  const int tw_ = tile % a.tiles_w, th_ = (tile / a.tiles_w) % a.tiles_h, td_ = tile / (a.tiles_w * a.tiles_h);
  const int d0 = td_ * TD, h0 = th_ * TH, w0 = tw_ * TW;

  // ---- per-thread halo staging geometry (chunk invariant) ----
  const int kg_t = tid & (KG - 1);
  long goff[NIT]; int loff[NIT];
  const T* xin = (const T*)a.x + (long)n * a.D * a.H * a.W * a.Cin_stride + a.Cin_off;
#pragma unroll
  for (int j = 0; j < NIT; ++j) {
    int it = tid + 256 * j;
    int hv = it >> 2;
    int hd = hv / (HH * HW), rem = hv - hd * (HH * HW), hy = rem / HW, hx = rem - hy * HW;
    int gd = d0 + hd - 1, gh = h0 + hy - 1, gw = w0 + hx - 1;
    bool ok = it < NITEMS && gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
    goff[j] = ok ? (((long)gd * a.H + gh) * a.W + gw) * a.Cin_stride + kg_t * EPG : -1;
    loff[j] = it < NITEMS ? hd * PS + hy * RS + hx * VS + kg_t * 16 : -1;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][q][i] = 0.f;

  if (a.xf.stats != nullptr) xform_preamble(a.xf, n, a.Cin, xsc, xsh, xad);   // visible after the loop's first barrier
  const int a_base = wave * PS + (r >> 3) * RS + (r & 7) * VS + hh * 16;
  const int b_base = (hh * BN + r) * 16;
  const char* wsrc = (const char*)a.w + (long)ct * a.nchunks * 3 * WSLAB;

  for (int ch = 0; ch < a.nchunks; ++ch) {
    __syncthreads();  // every wave is done reading the previous chunk's halo and weights
    // ---- stage the halo tile of this chunk: global -> registers -> (transform) -> LDS ----
    {
      const int c0 = ch * CK + kg_t * EPG;
      const bool cok = c0 < a.Cin;
      Frag v[NIT];
#pragma unroll
      for (int j = 0; j < NIT; ++j) {
        if (goff[j] >= 0 && cok) v[j] = *(const Frag*)(xin + goff[j] + ch * CK);
        else
#pragma unroll
          for (int e = 0; e < EPG; ++e) v[j][e] = (T)0.f;
      }
      if (a.xf.stats != nullptr && cok) {
        float sc[EPG], sh[EPG], ad[EPG];
#pragma unroll
        for (int e = 0; e < EPG; ++e) { sc[e] = xsc[c0 + e]; sh[e] = xsh[c0 + e]; ad[e] = xad[c0 + e]; }
#pragma unroll
        for (int j = 0; j < NIT; ++j)
          if (goff[j] >= 0) v[j] = xform_frag<T>(v[j], sc, sh, ad, a.xf.slope);
      }
#pragma unroll
      for (int j = 0; j < NIT; ++j)
        if (loff[j] >= 0) *(Frag*)(halo + loff[j]) = v[j];
    }
#pragma unroll 1
    for (int kd = 0; kd < 3; ++kd) {
      if (kd > 0) __syncthreads();  // previous slab fully consumed
      {
        const char* src = wsrc + ((long)ch * 3 + kd) * WSLAB;
#pragma unroll
        for (int j = 0; j < WSLAB / 16 / 256; ++j)
          *(f32x4*)(wlds + (tid + 256 * j) * 16) = *(const f32x4*)(src + (tid + 256 * j) * 16);
      }
      __syncthreads();
      const char* ap = halo + a_base + kd * PS;
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
        const int kh = t9 / 3, kw = t9 % 3;
#pragma unroll
        for (int ks = 0; ks < KG / 2; ++ks) {
          Frag a0 = *(const Frag*)(ap + kh * RS + kw * VS + ks * 32);
          Frag a1 = *(const Frag*)(ap + (kh + 4) * RS + kw * VS + ks * 32);
          Frag b0 = *(const Frag*)(wlds + b_base + (t9 * KG + 2 * ks) * BN * 16);
          Frag b1 = *(const Frag*)(wlds + b_base + (t9 * KG + 2 * ks) * BN * 16 + 32 * 16);
          mma32(acc[0][0], a0, b0);
          mma32(acc[0][1], a0, b1);
          mma32(acc[1][0], a1, b0);
          mma32(acc[1][1], a1, b1);
        }
      }
    }
  }

  // ---- epilogue: bias, InstanceNorm partials, transpose through LDS, 16-byte stores ----
  __syncthreads();
  constexpr int OS = BN * (int)sizeof(T) + 16;  // padded row stride of the [voxel][cout] staging tile
  char* ot = smem + wave * 64 * OS;
  const int gd = d0 + wave;
  const bool dok = gd < a.D;
  float cnt = 0.f;
  {
    // valid voxels of this wave's slab (same for every lane)
    int vh = min(TH, a.H - h0), vw = min(TW, a.W - w0);
    cnt = dok ? (float)(vh * vw) : 0.f;
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int co = q * 32 + r;
    const float bq = a.bias[ct * BN + co];
    float s = 0.f;
    float vals[2][16];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int hl = 4 * m + (i >> 2), wl = (i & 3) + 4 * hh;
        const bool ok = dok && (h0 + hl < a.H) && (w0 + wl < a.W);
        T tv = (T)(acc[m][q][i] + bq);
        float fv = ok ? (float)tv : 0.f;
        vals[m][i] = fv;
        s += fv;
        *(T*)(ot + (m * 32 + acc_row(i, hh)) * OS + co * (int)sizeof(T)) = tv;
      }
    s += __shfl_xor(s, 32);
    const float mean = cnt > 0.f ? s / cnt : 0.f;
    float m2 = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int hl = 4 * m + (i >> 2), wl = (i & 3) + 4 * hh;
        const bool ok = dok && (h0 + hl < a.H) && (w0 + wl < a.W);
        const float dlt = vals[m][i] - mean;
        m2 += ok ? dlt * dlt : 0.f;
      }
    m2 += __shfl_xor(m2, 32);
    if (hh == 0) {
      float* e = (float*)(smem + 4 * 64 * OS) + (wave * BN + co) * 2;   // per-wave (sum, M2) exchange
      e[0] = s; e[1] = m2;
    }
  }
  if (lane == 0) ((float*)(smem + 4 * 64 * OS))[4 * BN * 2 + wave] = cnt;
  __syncthreads();
  if (wave == 0) {
    // combine the four slabs (Chan) and publish: sum x and sum x^2 of this tile, in fp64
    const float* e = (const float*)(smem + 4 * 64 * OS);
    double S = 0, Q = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float k = e[4 * BN * 2 + w];
      if (k > 0.f) {
        const double sw = (double)e[(w * BN + lane) * 2], mw = (double)e[(w * BN + lane) * 2 + 1];
        S += sw; Q += mw + sw * sw / (double)k;
      }
    }
    if (ct * BN + lane < a.Cout) stats_add(a.stats, n, a.cout_pad, blockIdx.x & (STAT_REPLICAS - 1), ct * BN + lane, S, Q);
  }
  if (dok) {
    constexpr int GPV = BN / EPG;            // 16-byte groups per voxel
    constexpr int VPI = 64 / GPV;            // voxels per wave-iteration
    T* yout = (T*)a.y + (long)n * a.D * a.H * a.W * a.Cout_stride + a.Cout_off + ct * BN;
#pragma unroll
    for (int it = 0; it < 64 / VPI; ++it) {
      const int v = it * VPI + lane / GPV, cg = lane % GPV;
      const int hl = v >> 3, wl = v & 7;
      const int gh = h0 + hl, gw = w0 + wl;
      if (gh < a.H && gw < a.W && ct * BN + cg * EPG < a.Cout) {
        Frag o = *(const Frag*)(ot + v * OS + cg * 16);
        *(Frag*)(yout + (((long)gd * a.H + gh) * a.W + gw) * a.Cout_stride + cg * EPG) = o;
      }
    }
  }
}

template <typename T>
static int launch_conv3(const dua_conv3_desc* d, const void* x, const void* w, const float* bias,
                        const dua_in_norm* in, void* y, double* stats, hipStream_t s) {
  using namespace c3;
  constexpr int CK = KG * Elem<T>::EPG;
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
  const int lds = LDS_BYTES + (in && in->stats ? 3 * 4 * a.nchunks * CK : 0);
  static int attr_lds = 0;
  if (lds > attr_lds) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3d_k3_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + 3 * 4 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_lds = LDS_BYTES + 3 * 4 * 1024;
  }
  if (a.nchunks * CK > 1024) return DUA_ERR_ARG;
  dim3 grid(a.ntiles, nct, d->N);
  hipLaunchKernelGGL(conv3d_k3_kernel<T>, grid, dim3(256), lds, s, a);
  return (int)hipGetLastError();
}

}  // namespace dua

extern "C" {

int dua_conv3d_k3_fwd(const dua_conv3_desc* d, const void* x, const void* w_packed, const float* bias_padded,
                      const dua_in_norm* in, void* y, double* out_stats, void* stream) {
  if (!d || !x || !w_packed || !bias_padded || !y || !out_stats) return DUA_ERR_ARG;
  if (d->Cin % 8 || d->Cout % 8 || d->Cin_stride % 8 || d->Cout_stride % 8 || d->Cin_off % 8 || d->Cout_off % 8)
    return DUA_ERR_ARG;
  if (in && in->stats && (!in->gamma || !in->beta || in->c_pad < d->Cin)) return DUA_ERR_ARG;
  if (d->dtype == DUA_F16)
    return dua::launch_conv3<dua::f16>(d, x, w_packed, bias_padded, in, y, out_stats, (hipStream_t)stream);
  if (d->dtype == DUA_F32)
    return dua::launch_conv3<float>(d, x, w_packed, bias_padded, in, y, out_stats, (hipStream_t)stream);
  return DUA_ERR_ARG;
}

}  // extern "C"
