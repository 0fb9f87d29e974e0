// ConvTranspose3d(kernel 2, stride 2, bias) as 8 tap-GEMMs with a pixel-shuffle store.
//
// Replaces MONAI UpSample(mode="deconv") at models/basic_unet/denoiser.py:161-170 and the
// "upsampled" half of torch.cat([x_e, x_0], 1) at denoiser.py:190: the result is written
// straight into channels [Cout_off, Cout_off+Cout) of the concat buffer.
//   out[n, 2d+i, 2h+j, 2w+k, co] = bias[co] + sum_ci x[n,d,h,w,ci] * W[ci,co,i,j,k]
// GEMM view per tap: M = input voxels, N = Cout, K = Cin.  Workgroup: 256 consecutive input
// voxels x 64 output channels x one tap; wave w owns voxels [64w, 64w+64) as 2x2 MFMA 32x32
// accumulators.  The producer's InstanceNorm+LeakyReLU is applied while staging (InXform).
#include "common.hpp"
#include "../../include/dua_hip.h"

namespace dua {

namespace dc {
constexpr int TM = 256, BN = 64, KG = 4;
constexpr int VS = KG * 16 + 16;          // 80 B per voxel: conflict-free for 32 consecutive rows
constexpr int A_BYTES = TM * VS;          // 20480
constexpr int W_BYTES = KG * BN * 16;     // 4096
}  // namespace dc

struct DeconvArgs {
  const void* x; const void* w; const float* bias; void* y;
  InXform xf;
  int N, D, H, W;                  // input spatial
  int Cin, Cin_stride, Cin_off, Cout, Cout_stride, Cout_off;
  int nchunks, nct, lds_base;
};

template <typename T>
__global__ __launch_bounds__(256) void deconv_k2s2_kernel(DeconvArgs a) {
  using namespace dc;
  using Frag = typename Elem<T>::Frag;
  constexpr int EPG = Elem<T>::EPG;
  constexpr int CK = KG * EPG;
  constexpr int OS = BN * (int)sizeof(T) + 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* alds = smem;
  char* wlds = smem + A_BYTES;
  float* xsc = (float*)(smem + a.lds_base);
  float* xsh = xsc + a.nchunks * CK;
  float* xad = xsh + a.nchunks * CK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const long vox = (long)a.D * a.H * a.W;
  const long v0 = (long)blockIdx.x * TM;
  const int tap = blockIdx.y / a.nct, ct = blockIdx.y % a.nct, n = blockIdx.z;
  const T* xin = (const T*)a.x + (long)n * vox * a.Cin_stride + a.Cin_off;
  const char* wsrc = (const char*)a.w + ((long)tap * a.nct + ct) * a.nchunks * W_BYTES;

  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][q][i] = 0.f;

  const int kg_t = tid & 3;
  if (a.xf.stats != nullptr) xform_preamble(a.xf, n, a.Cin, xsc, xsh, xad);
  for (int ch = 0; ch < a.nchunks; ++ch) {
    __syncthreads();
    const int c0 = ch * CK + kg_t * EPG;
    const bool cok = c0 < a.Cin;
    float sc[EPG], sh[EPG], ad[EPG];
    const bool xf = a.xf.stats != nullptr && cok;
    if (xf) {
#pragma unroll
      for (int e = 0; e < EPG; ++e) { sc[e] = xsc[c0 + e]; sh[e] = xsh[c0 + e]; ad[e] = xad[c0 + e]; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int vl = (tid >> 2) + 64 * j;
      const long v = v0 + vl;
      Frag f;
      if (v < vox && cok) {
        f = *(const Frag*)(xin + v * a.Cin_stride + c0);
        if (xf) f = xform_frag<T>(f, sc, sh, ad, a.xf.slope);
      } else {
#pragma unroll
        for (int e = 0; e < EPG; ++e) f[e] = (T)0.f;
      }
      *(Frag*)(alds + vl * VS + kg_t * 16) = f;
    }
    *(f32x4*)(wlds + tid * 16) = *(const f32x4*)(wsrc + (long)ch * W_BYTES + tid * 16);
    __syncthreads();
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
  for (int q = 0; q < 2; ++q) {
    const int co = q * 32 + r;
    const float bq = a.bias[ct * BN + co];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        *(T*)(ot + (m * 32 + acc_row(i, hh)) * OS + co * (int)sizeof(T)) = (T)(acc[m][q][i] + bq);
  }
  __syncthreads();
  constexpr int GPV = BN / EPG, VPI = 64 / GPV;
  const int ti = tap >> 2, tj = (tap >> 1) & 1, tk = tap & 1;
  const int H2 = 2 * a.H, W2 = 2 * a.W;
  T* yout = (T*)a.y + (long)n * vox * 8 * a.Cout_stride + a.Cout_off + ct * BN;
#pragma unroll
  for (int it = 0; it < 64 / VPI; ++it) {
    const int vl = it * VPI + lane / GPV, cg = lane % GPV;
    const long v = v0 + wave * 64 + vl;
    if (v < vox && ct * BN + cg * EPG < a.Cout) {
      const int w = (int)(v % a.W); const long t = v / a.W;
      const int h = (int)(t % a.H), d = (int)(t / a.H);
      const long ov = ((long)(2 * d + ti) * H2 + (2 * h + tj)) * W2 + (2 * w + tk);
      *(Frag*)(yout + ov * a.Cout_stride + cg * EPG) = *(const Frag*)(ot + vl * OS + cg * 16);
    }
  }
}

template <typename T>
static int launch_deconv(const dua_conv3_desc* d, const void* x, const void* w, const float* bias,
                         const dua_in_norm* in, void* y, hipStream_t s) {
  constexpr int CK = dc::KG * Elem<T>::EPG;
  DeconvArgs a;
  a.x = x; a.w = w; a.bias = bias; a.y = y;
  a.xf = make_xform(in, d->Cin);
  a.N = d->N; a.D = d->D; a.H = d->H; a.W = d->W;
  a.Cin = d->Cin; a.Cin_stride = d->Cin_stride; a.Cin_off = d->Cin_off;
  a.Cout = d->Cout; a.Cout_stride = d->Cout_stride; a.Cout_off = d->Cout_off;
  a.nchunks = (d->Cin + CK - 1) / CK;
  a.nct = (d->Cout + dc::BN - 1) / dc::BN;
  const long vox = (long)d->D * d->H * d->W;
  dim3 grid((unsigned)((vox + dc::TM - 1) / dc::TM), 8 * a.nct, d->N);
  constexpr int OS = dc::BN * (int)sizeof(T) + 16;
  constexpr int LDS = (dc::TM * OS > dc::A_BYTES + dc::W_BYTES) ? dc::TM * OS : dc::A_BYTES + dc::W_BYTES;
  a.lds_base = LDS;
  if (a.nchunks * CK > 1024) return DUA_ERR_ARG;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)deconv_k2s2_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS + 3 * 4 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(deconv_k2s2_kernel<T>, grid, dim3(256), LDS + (a.xf.stats ? 3 * 4 * a.nchunks * CK : 0), s, a);
  return (int)hipGetLastError();
}

}  // namespace dua

extern "C" int dua_deconv_k2s2_fwd(const dua_conv3_desc* d, const void* x, const void* w_packed,
                                   const float* bias_padded, const dua_in_norm* in, void* y, void* stream) {
  if (!d || !x || !w_packed || !bias_padded || !y) return DUA_ERR_ARG;
  if (in && in->stats && (!in->gamma || !in->beta || in->c_pad < d->Cin)) return DUA_ERR_ARG;
  if (d->Cin % 8 || d->Cout % 8 || d->Cin_stride % 8 || d->Cout_stride % 8 || d->Cin_off % 8 || d->Cout_off % 8)
    return DUA_ERR_ARG;
  if (d->dtype == DUA_F16) return dua::launch_deconv<dua::f16>(d, x, w_packed, bias_padded, in, y, (hipStream_t)stream);
  if (d->dtype == DUA_F32) return dua::launch_deconv<float>(d, x, w_packed, bias_padded, in, y, (hipStream_t)stream);
  return DUA_ERR_ARG;
}
