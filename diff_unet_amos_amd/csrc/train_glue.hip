// The training step around the network (train.py:214-268 of the reference): what is neither a convolution nor a
// normalisation pass.  Each kernel here replaces a chain of small torch / hipBLASLt launches that the captured step used to
// replay one after the other at ~4.8 us apiece (172 of them, 1.5 ms of a 15.6 ms step):
//   stats_channel_sums   bias gradient of a transposed convolution from the statistics words over its output gradient
//   seg_loss_finish      losses/loss.py:64-86, the scalar tail (terms, combine, derivative of the combine)
//   q_sample_affine      x_start = label * 2 - 1 (train.py:258) and q_sample (gaussian_diffusion.py:214-231) in one pass,
//                        coefficients gathered by the device-resident timesteps
//   temb_train_*         TimeStepEmbedder + swish + the nine temb_proj (utils.py:5-54, denoiser.py:51-52,65), forward and
//                        backward (eleven Linear layers = 33 library GEMMs, their bias reductions and gradient sums before)
//   grads_nonfinite / adamw_* torch.cuda.amp's unscale + inf check, torch.optim.AdamW and the loss-scale update (train.py:121-126,264-268)
// All of it is fp32 (fp64 where the torch code it replaces was), bandwidth- or latency-bound, deterministic (no floating-point atomics).
#include "common.hpp"
#include "../../include/dua_hip.h"

namespace dua {

__device__ __forceinline__ float tg_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float tg_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float tg_swish(float x) { return x * tg_sigmoid(x); }
__device__ __forceinline__ float tg_dswish(float x) {
  const float s = tg_sigmoid(x);
  return s * (1.f + x * (1.f - s));
}

// ---- bias gradient from statistics rows ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stats_channel_sums_kernel(int N, int C, int c_pad, const stat_t* __restrict__ stats,
                                                                 float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double acc = 0.0;
  for (int n = 0; n < N; ++n) {
    double S, Q;
    stats_read(stats, n, c_pad, c, S, Q);
    acc += S;
  }
  out[c] = (float)acc;
}

// ---- loss tail ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void seg_loss_finish_kernel(int NC, double M, int use_mse, int use_bce, int use_dice, int combine,
                                                             const double* __restrict__ sums, float* loss, float* dcomb) {
  double d = 0.0;
  for (int i = threadIdx.x; i < NC; i += 64) {
    const double* q = sums + 4L * i;
    d += 1.0 - (2.0 * q[0] + 1e-5) / (q[1] + q[2] + 1e-5);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
  if (threadIdx.x != 0) return;
  double total = 0.0;
  int count = 0;
  if (use_mse) { total += sums[4L * NC] / M; ++count; }
  if (use_bce) { total += sums[4L * NC + 1] / M; ++count; }
  if (use_dice) { total += d / (double)NC; ++count; }
  double L = total, dc = 1.0;
  if (count > 1 && combine == 1) { L = total / count; dc = 1.0 / count; }
  else if (count > 1 && combine == 2) { L = log(1.0 + total); dc = 1.0 / (1.0 + total); }
  loss[0] = (float)L;
  dcomb[0] = (float)dc;
}

// ---- q_sample on 2 * label - 1 ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void q_sample_affine_kernel(long per, const float* __restrict__ src, float a, float b,
                                                              const float* __restrict__ eps, const float* __restrict__ sched, int T,
                                                              const long long* __restrict__ t, float* __restrict__ out, int vec) {
  const int n = blockIdx.y;
  long long tn = t[n];
  tn = tn < 0 ? 0 : (tn >= T ? T - 1 : tn);
  const float c0 = sched[2 * tn], c1 = sched[2 * tn + 1];
  const long base = (long)n * per;
  if (vec) {
    const f32x4* s4 = (const f32x4*)(src + base);
    const f32x4* e4 = (const f32x4*)(eps + base);
    f32x4* o4 = (f32x4*)(out + base);
    const long n4 = per / 4;
    // four pieces of each stream per thread, requested before anything waits (the coefficient lookup t -> sched in front of a
    // single piece per thread made every workgroup three dependent round trips: 83 us for 340 MB)
    for (long i0 = blockIdx.x * 1024L + threadIdx.x; i0 < n4; i0 += (long)gridDim.x * 1024) {
      f32x4 s[4], e[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long i = i0 + u * 256 < n4 ? i0 + u * 256 : n4 - 1;
        s[u] = s4[i]; e[u] = e4[i];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (i0 + u * 256 >= n4) break;
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float x0 = a * s[u][k] + b;          // rounded to fp32 like the tensor torch would have materialised (-ffp-contract=off)
          o[k] = c0 * x0 + c1 * e[u][k];
        }
        o4[i0 + u * 256] = o;
      }
    }
  } else {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < per; i += (long)gridDim.x * 256) {
      const float x0 = a * src[base + i] + b;
      out[base + i] = c0 * x0 + c1 * eps[base + i];
    }
  }
}

// ---- timestep embedding, training forward: three dependent stages, each spread over rows / 16 workgroups per sample -------------
// STAGE 0: z1 = W0 e + b0, h1 = swish(z1) (e = [sin | cos](t freqs) recomputed by every workgroup, stored by the first);
// STAGE 1: z2 = W1 h1 + b1, s = swish(z2);  STAGE 2: add_b = Wp_b s + bp_b over the concatenated rows of the blocks.
// A wave owns four output rows at a time (four rows of weights in flight), lanes walk the row, one cross-lane sum per row.  One
// workgroup per sample for the whole chain (the shape of temb_table_kernel, which runs once per weight update) took ~2 600 dependent
// row reductions on four waves.
constexpr int TF_ROWS = 16;
template <int STAGE>
__global__ __launch_bounds__(256) void temb_train_fwd_kernel(const long long* __restrict__ ts, const float* __restrict__ freqs, int half,
                                                             int hid, const float* __restrict__ w, const float* __restrict__ bias,
                                                             dua_temb_blocks blk, int rows, int N, float* __restrict__ add,
                                                             float* __restrict__ saved) {
  extern __shared__ float in[];          // the stage's input vector
  const int ed = 2 * half;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = blockIdx.y;
  float* sv = saved + (long)n * (ed + 4 * hid);
  const int cols = STAGE == 0 ? ed : hid;
  if (STAGE == 0) {
    const float t = (float)ts[n];
    for (int j = tid; j < half; j += 256) {
      const float arg = t * freqs[j];
      const float sn = sinf(arg), cs = cosf(arg);
      in[j] = sn; in[half + j] = cs;
      if (blockIdx.x == 0) { sv[j] = sn; sv[half + j] = cs; }
    }
  } else {
    const float* src = sv + (STAGE == 1 ? ed + hid : ed + 3 * hid);
    for (int k = tid; k < hid; k += 256) in[k] = src[k];
  }
  __syncthreads();
  const int r0 = blockIdx.x * TF_ROWS + wave * 4;
  const float* wr[4];
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int bo[4], bb[4];                      // STAGE 2: (block, row inside it) of each of the four rows
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = min(r0 + q, rows - 1);
    if (STAGE == 2) {
      int off = 0, b = 0;
      while (r >= off + blk.cout[b]) { off += blk.cout[b]; ++b; }
      bb[q] = b; bo[q] = r - off;
      wr[q] = blk.w[b] + (long)(r - off) * hid;
    } else {
      bb[q] = 0; bo[q] = r;
      wr[q] = w + (long)r * cols;
    }
  }
  for (int k = lane; k < cols; k += 64) {
    const float x = in[k];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = fmaf(wr[q][k], x, acc[q]);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = tg_wave_sum(acc[q]);
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (r0 + q >= rows) break;
      if (STAGE == 2) {
        int off = 0;
        for (int b = 0; b < bb[q]; ++b) off += blk.cout[b];
        add[(long)N * off + (long)n * blk.cout[bb[q]] + bo[q]] = acc[q] + blk.b[bb[q]][bo[q]];
      } else {
        const float z = acc[q] + bias[bo[q]];
        const int base = STAGE == 0 ? ed : ed + 2 * hid;
        sv[base + bo[q]] = z;
        sv[base + hid + bo[q]] = tg_swish(z);      // STAGE 1: the swish every TwoConv applies before its temb_proj (denoiser.py:65)
      }
    }
  }
}

// ---- timestep embedding, backward ---------------------------------------------------------------------------------------------
// y[k] = sum_r W[r][k] d[r] for row-major W (a transposed matrix-vector product), twice in a chain (temb_proj rows -> d s, then
// W1 -> d h1), spread over workgroups of 64 rows: a thread owns four adjacent columns and every other row of the chunk (32
// independent 16-byte loads), the two row halves meet in LDS, the chunk's partial vector goes to scratch and the NEXT launch
// sums the chunks in a fixed order in its prologue.  (First form: one 16-wave workgroup per sample for the whole chain: 73 us.)
constexpr int TB_ROWS = 64;

// partial[k0..k0+3] over rows [r0, r1) of a matrix whose rows are (a) the concatenated temb_proj weights or (b) w1
template <bool BLOCKS>
__device__ __forceinline__ f32x4 tb_partial(const dua_temb_blocks& blk, const float* __restrict__ w1, int hid, int r0, int r1, int rstep,
                                            const float* dl, int k0) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int b = 0, off = 0;
  // eight rows at a time: first their addresses (the walk over the blocks is scalar work), then eight independent 16-byte loads --
  // with the walk inside the load loop the loads went out one per round trip (29 us for the 1 536 temb_proj rows)
  for (int rb = r0; rb < r1; rb += 8 * rstep) {
    const float* row[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = min(rb + u * rstep, r1 - 1);          // clamped address, masked use
      if (BLOCKS) {
        while (r >= off + blk.cout[b]) { off += blk.cout[b]; ++b; }
        row[u] = blk.w[b] + (long)(r - off) * hid + k0;
      } else {
        row[u] = w1 + (long)r * hid + k0;
      }
    }
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)row[u];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = rb + u * rstep;
      const float d = r < r1 ? dl[min(r, r1 - 1) - r0] : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = fmaf(v[u][q], d, acc[q]);
    }
  }
  return acc;
}

// scratch: dz2 [N][hid] | part1 [N][nch1][hid] | part2 [N][nch2][hid]
__global__ __launch_bounds__(256) void temb_bwd_ds_kernel(int N, int hid, dua_temb_blocks blk, int P, const float* __restrict__ dadd,
                                                          float* __restrict__ scratch, int nch1) {
  __shared__ float dl[TB_ROWS];
  __shared__ __attribute__((aligned(16))) float half1[512];
  const int tid = threadIdx.x, n = blockIdx.y, chunk = blockIdx.x;
  const int r0 = chunk * TB_ROWS, r1 = min(P, r0 + TB_ROWS);
  if (tid < r1 - r0) {                       // this sample's d add of the chunk's rows (block-major source)
    const int r = r0 + tid;
    int b = 0, off = 0;
    while (r >= off + blk.cout[b]) { off += blk.cout[b]; ++b; }
    dl[tid] = dadd[(long)N * off + (long)n * blk.cout[b] + (r - off)];
  }
  __syncthreads();
  const int col = tid & 127, rg = tid >> 7, k0 = col * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (k0 < hid) acc = tb_partial<true>(blk, nullptr, hid, r0 + rg, r1, 2, dl + rg, k0);
  if (rg == 1 && k0 < hid) *(f32x4*)(half1 + k0) = acc;
  __syncthreads();
  if (rg == 0 && k0 < hid) {
    const f32x4 o = *(const f32x4*)(half1 + k0);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] += o[q];
    *(f32x4*)(scratch + (long)N * hid + ((long)n * nch1 + chunk) * hid + k0) = acc;
  }
}

__global__ __launch_bounds__(256) void temb_bwd_dh_kernel(int N, int half, int hid, const float* __restrict__ w1, const float* __restrict__ saved,
                                                          float* __restrict__ scratch, int nch1, int nch2) {
  __shared__ float dv[512];                   // dz2 of this sample (all of it: every workgroup needs its own 64 rows AND none else, but
  __shared__ __attribute__((aligned(16))) float half1[512];     // the sum over part1 chunks is cheapest done once per workgroup for its rows)
  const int tid = threadIdx.x, n = blockIdx.y, chunk = blockIdx.x;
  const int ed = 2 * half;
  const float* z2 = saved + (long)n * (ed + 4 * hid) + ed + 2 * hid;
  const float* p1 = scratch + (long)N * hid + (long)n * nch1 * hid;
  const int r0 = chunk * TB_ROWS;             // rows of w1 = elements of dz2 this workgroup multiplies
  if (tid < TB_ROWS) {
    const int k = r0 + tid;
    float ds = 0.f;
    for (int c = 0; c < nch1; ++c) ds += p1[(long)c * hid + k];
    const float g = ds * tg_dswish(z2[k]);
    dv[tid] = g;
    scratch[(long)n * hid + k] = g;           // dz2, for the outer-product launch
  }
  __syncthreads();
  const int col = tid & 127, rg = tid >> 7, k0 = col * 4;
  dua_temb_blocks none{};
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (k0 < hid) acc = tb_partial<false>(none, w1, hid, r0 + rg, r0 + TB_ROWS, 2, dv + rg, k0);
  if (rg == 1 && k0 < hid) *(f32x4*)(half1 + k0) = acc;
  __syncthreads();
  if (rg == 0 && k0 < hid) {
    const f32x4 o = *(const f32x4*)(half1 + k0);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] += o[q];
    *(f32x4*)(scratch + (long)N * hid * (1 + nch1) + ((long)n * nch2 + chunk) * hid + k0) = acc;
  }
}

// ---- every parameter gradient, one workgroup per output row ------------------------------------------------------------------
__global__ __launch_bounds__(256) void temb_bwd_outer_kernel(int N, int half, int hid, dua_temb_blocks blk, int P,
                                                             const float* __restrict__ dadd, const float* __restrict__ saved,
                                                             const float* __restrict__ scratch, int nch1, int nch2, float* __restrict__ dw0,
                                                             float* __restrict__ db0, float* __restrict__ dw1, float* __restrict__ db1) {
  __shared__ float dn[64];           // this row's output gradient per sample
  const int row = blockIdx.x, tid = threadIdx.x;
  const int ed = 2 * half, sstride = ed + 4 * hid;
  const float* act;                  // per-sample activation vector the row multiplies: saved + n * sstride + act_off
  int cols, act_off;
  float* dst;
  float* bdst;
  if (row < P) {
    int off = 0, b = 0;
    while (row >= off + blk.cout[b]) { off += blk.cout[b]; ++b; }
    const int co = blk.cout[b], o = row - off;
    if (tid < N) dn[tid] = dadd[(long)N * off + (long)tid * co + o];
    cols = hid; act_off = ed + 3 * hid;
    dst = blk.dw[b] + (long)o * hid; bdst = blk.db[b] + o;
  } else if (row < P + hid) {
    const int o = row - P;
    if (tid < N) dn[tid] = scratch[(long)tid * hid + o];
    cols = hid; act_off = ed + hid;
    dst = dw1 + (long)o * hid; bdst = db1 + o;
  } else {
    const int o = row - P - hid;
    if (tid < N) {                   // dz1 = swish'(z1) * sum over the w1 row chunks, in chunk order
      const float* p2 = scratch + (long)N * hid * (1 + nch1) + (long)tid * nch2 * hid + o;
      float dh = 0.f;
      for (int c = 0; c < nch2; ++c) dh += p2[(long)c * hid];
      dn[tid] = dh * tg_dswish(saved[(long)tid * sstride + ed + o]);
    }
    cols = ed; act_off = 0;
    dst = dw0 + (long)o * ed; bdst = db0 + o;
  }
  act = saved + act_off;
  __syncthreads();
  for (int k = tid; k < cols; k += 256) {
    float acc = 0.f;
    for (int n = 0; n < N; ++n) acc = fmaf(dn[n], act[(long)n * sstride + k], acc);
    dst[k] = acc;
  }
  if (tid == 0) {
    float acc = 0.f;
    for (int n = 0; n < N; ++n) acc += dn[n];
    *bdst = acc;
  }
}

// ---- AdamW over a list of tensors ---------------------------------------------------------------------------------------------
constexpr int AD_CHUNK = 4096;          // elements per workgroup: 256 threads x 4 x 16-byte pieces
struct AdamArgs {
  dua_adamw_list l;
  int first_block[DUA_ADAMW_MAX_TENSORS + 1];
};

__device__ __forceinline__ int adam_find(const AdamArgs& a, int b) {
  int lo = 0, hi = a.l.count;          // first_block[lo] <= b < first_block[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (a.first_block[mid] <= b) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void grads_nonfinite_kernel(AdamArgs a, float* found_inf) {
  const int ti = adam_find(a, blockIdx.x);
  const long off = (long)(blockIdx.x - a.first_block[ti]) * AD_CHUNK;
  const long n = min((long)AD_CHUNK, a.l.numel[ti] - off);
  const float* g = a.l.g[ti] + off;
  bool bad = false;
  if ((((size_t)g) & 15) == 0) {
    const long n4 = n >> 2;
    for (long i = threadIdx.x; i < n4; i += 256) {
      const f32x4 v = ((const f32x4*)g)[i];
      bad |= !(isfinite(v[0]) && isfinite(v[1]) && isfinite(v[2]) && isfinite(v[3]));
    }
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) bad |= !isfinite(g[i]);
  } else {
    for (long i = threadIdx.x; i < n; i += 256) bad |= !isfinite(g[i]);
  }
  if (bad) *found_inf = 1.f;            // every writer stores the same value
}

__device__ __forceinline__ void adam_one(float& p, float& m, float& v, float g, float lr_wd, float w1, float b2, float w2,
                                         float step_size, float bc2s, float eps) {
  p -= lr_wd * p;
  m = m + w1 * (g - m);
  v = b2 * v + w2 * g * g;
  const float denom = sqrtf(v) / bc2s + eps;
  p -= step_size * m / denom;
}

__global__ __launch_bounds__(256) void adamw_kernel(AdamArgs a, float lr, const float* lr_dev, float beta1, float beta2, float eps,
                                                    float wd, const float* grad_scale, const float* found_inf, const int* step,
                                                    int store_grad) {
  if (found_inf && *found_inf != 0.f) return;
  __shared__ float cst[2];
  if (threadIdx.x == 0) {
    const double k = (double)(*step + 1);
    cst[0] = (float)(1.0 - pow((double)beta1, k));
    cst[1] = (float)sqrt(1.0 - pow((double)beta2, k));
  }
  __syncthreads();
  if (lr_dev) lr = *lr_dev;
  const float inv = grad_scale ? 1.f / *grad_scale : 1.f;
  const float lr_wd = lr * wd, w1 = 1.f - beta1, w2 = 1.f - beta2, step_size = lr / cst[0], bc2s = cst[1];
  const int ti = adam_find(a, blockIdx.x);
  const long off = (long)(blockIdx.x - a.first_block[ti]) * AD_CHUNK;
  const long n = min((long)AD_CHUNK, a.l.numel[ti] - off);
  float* p = a.l.p[ti] + off;
  float* g = a.l.g[ti] + off;
  float* m = a.l.m[ti] + off;
  float* v = a.l.v[ti] + off;
  long done = 0;
  if (((((size_t)p) | ((size_t)g) | ((size_t)m) | ((size_t)v)) & 15) == 0) {
    const long n4 = n >> 2;
    for (long i = threadIdx.x; i < n4; i += 256) {
      f32x4 pv = ((f32x4*)p)[i], mv = ((f32x4*)m)[i], vv = ((f32x4*)v)[i], gv = ((const f32x4*)g)[i];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float gq = gv[q] * inv;
        float pq = pv[q], mq = mv[q], vq = vv[q];
        adam_one(pq, mq, vq, gq, lr_wd, w1, beta2, w2, step_size, bc2s, eps);
        pv[q] = pq; mv[q] = mq; vv[q] = vq; gv[q] = gq;
      }
      ((f32x4*)p)[i] = pv; ((f32x4*)m)[i] = mv; ((f32x4*)v)[i] = vv;
      if (store_grad) ((f32x4*)g)[i] = gv;
    }
    done = n4 << 2;
  }
  for (long i = done + threadIdx.x; i < n; i += 256) {
    float pv = p[i], mv = m[i], vv = v[i];
    const float gv = g[i] * inv;
    adam_one(pv, mv, vv, gv, lr_wd, w1, beta2, w2, step_size, bc2s, eps);
    p[i] = pv; m[i] = mv; v[i] = vv;
    if (store_grad) g[i] = gv;
  }
}

__global__ void adamw_advance_kernel(int* step, float* found_inf, float* scale, int* growth, float growth_factor, float backoff,
                                     int interval, float* seen) {
  const bool bad = found_inf && *found_inf != 0.f;
  if (seen) *seen = bad ? 1.f : 0.f;
  if (bad) {
    if (scale) *scale *= backoff;
    if (growth) *growth = 0;
  } else {
    *step += 1;
    if (growth) {
      const int ok = *growth + 1;
      if (ok == interval) {
        if (scale) {
          const float ns = *scale * growth_factor;
          if (isfinite(ns)) *scale = ns;
        }
        *growth = 0;
      } else {
        *growth = ok;
      }
    }
  }
  if (found_inf) *found_inf = 0.f;
}

static int adam_args(const dua_adamw_list* list, AdamArgs& a) {
  if (!list || list->count <= 0 || list->count > DUA_ADAMW_MAX_TENSORS) return -1;
  a.l = *list;
  long blocks = 0;
  for (int i = 0; i < list->count; ++i) {
    if (list->numel[i] <= 0 || !list->g[i]) return -1;
    a.first_block[i] = (int)blocks;
    blocks += (list->numel[i] + AD_CHUNK - 1) / AD_CHUNK;
    if (blocks > 0x7fffffffL) return -1;
  }
  for (int i = list->count; i <= DUA_ADAMW_MAX_TENSORS; ++i) a.first_block[i] = (int)blocks;
  return (int)blocks;
}

}  // namespace dua

extern "C" {

int dua_stats_channel_sums(int N, int C, int c_pad, const dua_stat_word* stats, float* out, void* stream) {
  if (N <= 0 || C <= 0 || c_pad < C || !stats || !out) return DUA_ERR_ARG;
  hipLaunchKernelGGL(dua::stats_channel_sums_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, C, c_pad,
                     (const dua::stat_t*)stats, out);
  return (int)hipGetLastError();
}

int dua_seg_loss_finish(int N, int C, long voxels, int use_mse, int use_bce, int use_dice, int combine, const double* sums,
                        float* loss, float* dcomb, void* stream) {
  if (N <= 0 || C <= 0 || voxels <= 0 || !sums || !loss || !dcomb || combine < 0 || combine > 2 ||
      !(use_mse || use_bce || use_dice))
    return DUA_ERR_ARG;
  hipLaunchKernelGGL(dua::seg_loss_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, N * C,
                     (double)N * (double)C * (double)voxels, use_mse, use_bce, use_dice, combine, sums, loss, dcomb);
  return (int)hipGetLastError();
}

int dua_q_sample_affine(int N, long per_sample, const float* src, float a, float b, const float* eps, const float* sched, int T,
                        const long long* t, float* out, void* stream) {
  if (N <= 0 || N > 65535 || per_sample <= 0 || !src || !eps || !sched || T <= 0 || !t || !out) return DUA_ERR_ARG;
  const int vec = per_sample % 4 == 0 && ((((size_t)src) | ((size_t)eps) | ((size_t)out)) & 15) == 0;
  long blocks = vec ? (per_sample / 4 + 1023) / 1024 : (per_sample + 255) / 256;
  if (blocks > 65535) blocks = 65535;
  hipLaunchKernelGGL(dua::q_sample_affine_kernel, dim3((unsigned)blocks, N), dim3(256), 0, (hipStream_t)stream, per_sample, src, a, b,
                     eps, sched, T, t, out, vec);
  return (int)hipGetLastError();
}

static int temb_blocks_ok(const dua_temb_blocks* blocks, int hidden, bool bwd, int* P) {
  if (!blocks || blocks->nblocks <= 0 || blocks->nblocks > DUA_TEMB_MAX_BLOCKS) return 0;
  int p = 0;
  for (int b = 0; b < blocks->nblocks; ++b) {
    if (blocks->cout[b] <= 0 || !blocks->w[b] || (((size_t)blocks->w[b]) & 15)) return 0;
    if (bwd ? (!blocks->dw[b] || !blocks->db[b]) : !blocks->b[b]) return 0;
    p += blocks->cout[b];
  }
  *P = p;
  return p <= 4096;
}

int dua_temb_train_fwd(int N, const long long* t, const float* freqs, int half_dim, int hidden, const float* w0, const float* b0,
                       const float* w1, const float* b1, const dua_temb_blocks* blocks, float* add, float* saved, void* stream) {
  int P = 0;
  if (N <= 0 || N > 64 || !t || !freqs || half_dim <= 0 || 2 * half_dim > 1024 || hidden <= 0 || hidden % 256 || hidden > 512 ||
      !w0 || !b0 || !w1 || !b1 || !add || !saved || !temb_blocks_ok(blocks, hidden, false, &P))
    return DUA_ERR_ARG;
  const hipStream_t st = (hipStream_t)stream;
  const dim3 g01((hidden + dua::TF_ROWS - 1) / dua::TF_ROWS, N), g2((P + dua::TF_ROWS - 1) / dua::TF_ROWS, N);
  hipLaunchKernelGGL(dua::temb_train_fwd_kernel<0>, g01, dim3(256), 2 * half_dim * sizeof(float), st, t, freqs, half_dim, hidden, w0, b0,
                     *blocks, hidden, N, add, saved);
  hipLaunchKernelGGL(dua::temb_train_fwd_kernel<1>, g01, dim3(256), hidden * sizeof(float), st, t, freqs, half_dim, hidden, w1, b1,
                     *blocks, hidden, N, add, saved);
  hipLaunchKernelGGL(dua::temb_train_fwd_kernel<2>, g2, dim3(256), hidden * sizeof(float), st, t, freqs, half_dim, hidden, nullptr,
                     nullptr, *blocks, P, N, add, saved);
  return (int)hipGetLastError();
}

int dua_temb_train_bwd(int N, int half_dim, int hidden, const float* w1, const dua_temb_blocks* blocks, const float* dadd,
                       const float* saved, float* scratch, float* dw0, float* db0, float* dw1, float* db1, void* stream) {
  int P = 0;
  if (N <= 0 || N > 64 || half_dim <= 0 || 2 * half_dim > 1024 || hidden <= 0 || hidden % 256 || hidden > 512 || !w1 ||
      (((size_t)w1) & 15) || !dadd || !saved || !scratch || !dw0 || !db0 || !dw1 || !db1 || !temb_blocks_ok(blocks, hidden, true, &P))
    return DUA_ERR_ARG;
  const hipStream_t st = (hipStream_t)stream;
  const int nch1 = (P + dua::TB_ROWS - 1) / dua::TB_ROWS, nch2 = hidden / dua::TB_ROWS;
  hipLaunchKernelGGL(dua::temb_bwd_ds_kernel, dim3(nch1, N), dim3(256), 0, st, N, hidden, *blocks, P, dadd, scratch, nch1);
  hipLaunchKernelGGL(dua::temb_bwd_dh_kernel, dim3(nch2, N), dim3(256), 0, st, N, half_dim, hidden, w1, saved, scratch, nch1, nch2);
  hipLaunchKernelGGL(dua::temb_bwd_outer_kernel, dim3(P + 2 * hidden), dim3(256), 0, st, N, half_dim, hidden, *blocks, P, dadd, saved,
                     (const float*)scratch, nch1, nch2, dw0, db0, dw1, db1);
  return (int)hipGetLastError();
}

int dua_grads_nonfinite(const dua_adamw_list* list, float* found_inf, void* stream) {
  dua::AdamArgs a;
  const int blocks = dua::adam_args(list, a);
  if (blocks <= 0 || !found_inf) return DUA_ERR_ARG;
  hipLaunchKernelGGL(dua::grads_nonfinite_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, found_inf);
  return (int)hipGetLastError();
}

int dua_adamw_step(const dua_adamw_list* list, float lr, const float* lr_dev, float beta1, float beta2, float eps,
                   float weight_decay, const float* grad_scale, const float* found_inf, const int* step, int store_grad,
                   void* stream) {
  dua::AdamArgs a;
  const int blocks = dua::adam_args(list, a);
  if (blocks <= 0 || !step || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) || !(eps >= 0.f)) return DUA_ERR_ARG;
  for (int i = 0; i < list->count; ++i)
    if (!list->p[i] || !list->m[i] || !list->v[i]) return DUA_ERR_ARG;
  hipLaunchKernelGGL(dua::adamw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, lr, lr_dev, beta1, beta2, eps,
                     weight_decay, grad_scale, found_inf, step, store_grad);
  return (int)hipGetLastError();
}

int dua_adamw_advance(int* step, float* found_inf, float* scale, int* growth, float growth_factor, float backoff, int interval,
                      float* seen, void* stream) {
  if (!step || interval <= 0) return DUA_ERR_ARG;
  hipLaunchKernelGGL(dua::adamw_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, found_inf, scale, growth,
                     growth_factor, backoff, interval, seen);
  return (int)hipGetLastError();
}

}  // extern "C"
