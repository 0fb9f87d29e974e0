// Measurement aid, not part of the hot path: what the matrix pipes of THIS device sustain on dense random fp16 data, and
// the shader clock it holds while doing so (MI355X_MICROARCH.md 'DVFS give-back' items 6 and 7).  bench.py launches it back
// to back for ~2 s and reports the rate as roofline.measured_mfma_ceiling next to the nominal 2.5 PFLOP/s: the chip lowers
// its clock under a dense MFMA load, so the nominal peak is not reachable by any kernel on random operands.
//
// One wave per SIMD (256-thread workgroups, one per CU), operands in registers, four independent accumulators,
// v_mfma_f32_32x32x16_f16 back to back; lane 0 of every workgroup stamps s_memtime / s_memrealtime around the loop into a
// buffer that nothing else reads.
#include "common.hpp"
#include "../../include/dua_hip.h"

namespace dua {

__global__ __launch_bounds__(256, 1) void mfma_probe_kernel(int iters, float* sink, unsigned long long* stamps) {
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  f16x8 a[2], b[2];
  unsigned s = t * 2654435761u + 12345u;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) {                      // uniform in [-1, 1): full-range random operands
      s = s * 1664525u + 1013904223u;
      a[j][e] = (f16)((float)(s >> 8) * (2.f / 16777216.f) - 1.f);
      s = s * 1664525u + 1013904223u;
      b[j][e] = (f16)((float)(s >> 8) * (2.f / 16777216.f) - 1.f);
    }
  f32x16 acc[4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
  unsigned long long c0 = 0, r0 = 0;
  if (threadIdx.x == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[1], acc[3], 0, 0, 0);
    }
  }
  if (threadIdx.x == 0 && stamps) {
    stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
    stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
  float v = 0.f;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) v += acc[m][i];
  if (v == 123.456f) sink[0] = v;                      // keeps the loop alive; practically never taken
}

// The floor of one launch of a normalise -> compute -> accumulate-statistics chain (what every layer of the <= 24^3 levels
// is): mode 0 = an empty kernel; 1 = one dependent global round trip (load, +1, store) per thread; 2 = 1 and the launch
// ends like a convolution does (workgroup reduction, one 64-bit system-scope atomic per channel lane of wave 0); 3 = 2 behind
// a read of the words the PREVIOUS launch's atomics wrote (the statistics preamble: a second dependent round trip, on lines
// that the atomics left outside L2).  tools/ubench_chain.py replays a graph of 24 such launches.
__global__ __launch_bounds__(256) void chain_probe_kernel(int mode, const float* in, float* out, unsigned long long* words) {
  if (mode == 0) return;
  __shared__ float red[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  float scale = 1.f;
  if (mode == 3) {
    const unsigned long long w = words[threadIdx.x & 63];
    scale = w == 0x7fffffffffffffffull ? 2.f : 1.f;            // depends on the loaded word; practically always 1
  }
  const float v = in[i] * scale + 1.f;
  out[i] = v;
  if (mode >= 2) {
    float s = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x < 64) {
      const float t = red[0] + red[1] + red[2] + red[3];
      __hip_atomic_fetch_add(words + threadIdx.x, (unsigned long long)(long long)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

}  // namespace dua

extern "C" int dua_chain_probe(int mode, int workgroups, const float* in, float* out, unsigned long long* words, void* stream) {
  if (mode < 0 || mode > 3 || workgroups <= 0 || !in || !out || !words) return DUA_ERR_ARG;
  hipLaunchKernelGGL(dua::chain_probe_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, mode, in, out, words);
  return (int)hipGetLastError();
}

// `workgroups` x 4 waves each issue iters * 16 MFMAs of 32 x 32 x 16 (32 768 FLOP each).  stamps (or NULL): 2 words per
// workgroup = (shader cycles, 100 MHz ticks) spent in the loop.
extern "C" int dua_mfma_probe(int workgroups, int iters, float* sink, unsigned long long* stamps, void* stream) {
  if (workgroups <= 0 || iters <= 0 || !sink) return DUA_ERR_ARG;
  hipLaunchKernelGGL(dua::mfma_probe_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, iters, sink, stamps);
  return (int)hipGetLastError();
}
