// Timestep embedding -> per-block channel biases, and the per-step "step begin" gather.
//
// Reference: models/diffusion/utils.py:6-54 (sinusoid [sin|cos], Linear(128,512) -> swish ->
// Linear(512,512)) followed, in every TwoConv, by temb_proj(swish(temb))
// (models/basic_unet/denoiser.py:51-52,65).  All of it depends only on the integer timestep,
// so the whole chain is evaluated once per weight update for every timestep the sampler can
// visit: table[t][:] = concat_b( temb_proj_b( swish( TimeStepEmbedder(t) ) ) ), P = sum of the
// nine blocks' Cout.  During sampling a one-workgroup kernel copies the row of the current
// step into the fixed buffer the convolution prologues read, together with that step's
// sampler coefficients -- so a captured hipGraph can be replayed for every step unchanged.
#include "common.hpp"
#include "../../include/dua_hip.h"

namespace dua {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float swishf(float x) { return x * (1.f / (1.f + expf(-x))); }

// one workgroup (256 threads = 4 waves) per timestep
__global__ __launch_bounds__(256) void temb_table_kernel(const int* __restrict__ ts, const float* __restrict__ freqs,
                                                         int half, int hid, const float* __restrict__ w0,
                                                         const float* __restrict__ b0, const float* __restrict__ w1,
                                                         const float* __restrict__ b1, const float* __restrict__ wc,
                                                         const float* __restrict__ bc, int P, float* __restrict__ table) {
  extern __shared__ float sm[];
  float* e = sm;                 // [2*half]
  float* h1 = e + 2 * half;      // [hid]
  float* h2 = h1 + hid;          // [hid]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float t = (float)ts[blockIdx.x];
  for (int j = tid; j < half; j += 256) {
    const float arg = t * freqs[j];
    e[j] = sinf(arg);
    e[half + j] = cosf(arg);
  }
  __syncthreads();
  const int ed = 2 * half;
  for (int o = wave; o < hid; o += 4) {
    float s = 0.f;
    for (int k = lane; k < ed; k += 64) s = fmaf(w0[o * ed + k], e[k], s);
    s = wave_sum(s);
    if (lane == 0) h1[o] = swishf(s + b0[o]);
  }
  __syncthreads();
  for (int o = wave; o < hid; o += 4) {
    float s = 0.f;
    for (int k = lane; k < hid; k += 64) s = fmaf(w1[o * hid + k], h1[k], s);
    s = wave_sum(s);
    if (lane == 0) h2[o] = swishf(s + b1[o]);   // swish applied by every TwoConv before temb_proj
  }
  __syncthreads();
  for (int o = wave; o < P; o += 4) {
    float s = 0.f;
    for (int k = lane; k < hid; k += 64) s = fmaf(wc[(long)o * hid + k], h2[k], s);
    s = wave_sum(s);
    if (lane == 0) table[(long)blockIdx.x * P + o] = s + bc[o];
  }
}

__global__ __launch_bounds__(256) void step_begin_kernel(int N, int P, const float* __restrict__ table, int table_rows,
                                                         const int* __restrict__ rows_per_sample,
                                                         const int* __restrict__ row_of_step, int nsteps,
                                                         const float* __restrict__ coef_table, int* counter,
                                                         float* __restrict__ cur_add, float* __restrict__ cur_coef,
                                                         int* step_word, int* err_word, f32x4* __restrict__ zero,
                                                         long zero_pieces) {
  // Blocks 1.. clear the statistics arena of the evaluation (16-byte pieces); block 0 does the step's bookkeeping.  The clear
  // used to be a hipMemsetAsync node in the captured step: replayed back to back, that runtime blit was seen starting before
  // the previous replay's tail kernel had finished (late workgroups of the tail then normalised with zeroed sums -- results
  // that depended on how far the host ran ahead).  As part of this kernel it is ordered like any other launch.
  if (blockIdx.x > 0) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (long i = (blockIdx.x - 1) * 256L + threadIdx.x; i < zero_pieces; i += (gridDim.x - 1) * 256L) zero[i] = z;
    return;
  }
  // Every index that comes from device memory is range-checked here: an out-of-range timestep or step counter must
  // not turn into a wild read (a GPU memory fault can reset the node).  Offenders are clamped and reported.
  int k = 0;
  bool bad = false;
  if (!rows_per_sample) {
    k = *counter;
    if (k < 0 || k >= nsteps) { bad = true; k = k < 0 ? 0 : nsteps - 1; }
  }
  for (int n = 0; n < N; ++n) {
    int row = rows_per_sample ? rows_per_sample[n] : row_of_step[k];
    if (row < 0 || row >= table_rows) { bad = true; row = row < 0 ? 0 : table_rows - 1; }
    for (int i = threadIdx.x; i < P; i += 256) cur_add[(long)n * P + i] = table[(long)row * P + i];
    if (coef_table && threadIdx.x < 8) cur_coef[8 * n + threadIdx.x] = coef_table[8 * k + threadIdx.x];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (bad && err_word) *err_word = 1;
    if (!rows_per_sample) {
      if (step_word) step_word[0] = k;
      *counter = k + 1;
    }
  }
}

}  // namespace dua

extern "C" {

int dua_temb_table(int count, const int* timesteps, const float* freqs, int half_dim, int hidden, const float* w0,
                   const float* b0, const float* w1, const float* b1, const float* w_cat, const float* b_cat, int P,
                   float* table, void* stream) {
  if (count <= 0 || !timesteps || !freqs || half_dim <= 0 || hidden <= 0 || !w0 || !b0 || !w1 || !b1 || !w_cat ||
      !b_cat || P <= 0 || !table)
    return DUA_ERR_ARG;
  const size_t lds = (size_t)(2 * half_dim + 2 * hidden) * sizeof(float);
  hipLaunchKernelGGL(dua::temb_table_kernel, dim3(count), dim3(256), lds, (hipStream_t)stream, timesteps, freqs,
                     half_dim, hidden, w0, b0, w1, b1, w_cat, b_cat, P, table);
  return (int)hipGetLastError();
}

int dua_step_begin_clear(int N, int P, const float* table, int table_rows, const int* rows_per_sample, const int* row_of_step,
                         int nsteps, const float* coef_table, int* counter, float* cur_add, float* cur_coef, int* step_word,
                         int* err_word, void* clear, long clear_bytes, void* stream) {
  if (N <= 0 || P <= 0 || !table || table_rows <= 0 || !cur_add) return DUA_ERR_ARG;
  if (!rows_per_sample && (!row_of_step || !counter || nsteps <= 0)) return DUA_ERR_ARG;
  if (coef_table && !cur_coef) return DUA_ERR_ARG;
  if (clear_bytes < 0 || clear_bytes % 16 || (clear_bytes > 0 && (!clear || ((size_t)clear & 15)))) return DUA_ERR_ARG;
  const long pieces = clear_bytes / 16;
  long zb = (pieces + 1023) / 1024;                 // four pieces per thread
  if (zb > 1024) zb = 1024;
  hipLaunchKernelGGL(dua::step_begin_kernel, dim3(1 + (unsigned)zb), dim3(256), 0, (hipStream_t)stream, N, P, table, table_rows,
                     rows_per_sample, row_of_step, nsteps, coef_table, counter, cur_add, cur_coef, step_word, err_word,
                     (dua::f32x4*)clear, pieces);
  return (int)hipGetLastError();
}

int dua_step_begin(int N, int P, const float* table, int table_rows, const int* rows_per_sample, const int* row_of_step,
                   int nsteps, const float* coef_table, int* counter, float* cur_add, float* cur_coef, int* step_word,
                   int* err_word, void* stream) {
  return dua_step_begin_clear(N, P, table, table_rows, rows_per_sample, row_of_step, nsteps, coef_table, counter, cur_add,
                              cur_coef, step_word, err_word, nullptr, 0, stream);
}

}  // extern "C"
