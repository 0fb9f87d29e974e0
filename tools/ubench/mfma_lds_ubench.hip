// Micro-benchmark behind the conv3d_k3 kernel design: what the inner loop of the implicit
// GEMM -- operand fragments from LDS with ds_read_b128 + v_mfma_f32_32x32x16_f16 -- sustains on gfx950 on random
// data, by wave-tile shape (MA x 2 accumulators of 32x32), waves per SIMD, barrier cadence and weight staging.
// No global traffic except an optional L2-resident weight stream; outputs are written once so nothing is dead code.
//
// build: hipcc -O3 --offload-arch=gfx950 mfma_lds_ubench.hip -o bin/mfma_lds_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int VS = 64;
template <int S16> struct Geo { static constexpr int RS = 10 * VS + (S16 ? 32 : 16), PS = 10 * RS; };   // halo row / plane strides (conflict-free per MFMA shape)
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int BN = 64, KG = 4, SLAB = 3 * KG * BN * 16;    // 12 KB weight slab (kd, kh)

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// MA: 32-row blocks per wave (2 = 64x64 wave tile, 3 = 96x64, 4 = 128x64); KSPLIT: the wave takes one of the two
// 16-wide k-steps of each tap (pairs of waves share a tile); BAR: 0 none, 1 one s_barrier per 12 KB slab;
// STAGE: 1 = every thread also moves its share of the next weight slab global -> regs -> LDS per slab.
template <int MA, int KSPLIT, int BAR, int STAGE, int NT, int S16 = 0>
__global__ __launch_bounds__(NT, NT == 256 ? 2 : 1) void loop_kernel(const char* __restrict__ src, const char* __restrict__ wsrc, float* out,
                                                  int nslabs, int halo_bytes, unsigned long long* clk) {
  constexpr int RS = Geo<S16>::RS, PS = Geo<S16>::PS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* halo = smem;
  char* wbuf = smem + halo_bytes;
  for (int i = tid * 16; i < halo_bytes + 2 * SLAB; i += NT * 16) *(f32x4*)(smem + i) = *(const f32x4*)(src + (i % (1 << 20)));
  __syncthreads();
  const int nw = NT / 64;
  const int mq = KSPLIT ? wave % (nw / 2) : wave, kp = KSPLIT ? wave / (nw / 2) : 0;
  // A operand rows: shape 32x32x16: 32-row block b = (depth slice b >> 1, h half b & 1), lane row r -> (h = r >> 3, w = r & 7),
  // k-group pair from the lane half; shape 16x16x32: 16-row block c = (depth c >> 2, h pair c & 3), k-group = lane >> 4.
  constexpr int NA = S16 ? 2 * MA : MA;
  int a_base[NA];
#pragma unroll
  for (int m = 0; m < NA; ++m) {
    if (S16) {
      const int c = mq * NA + m, row = lane & 15;
      a_base[m] = ((c >> 2) & 3) * PS + ((c & 3) * 2 + (row >> 3)) * RS + (row & 7) * VS + (lane >> 4) * 16;
    } else {
      const int b = mq * MA + m, r = lane & 31;
      a_base[m] = ((b >> 1) & 3) * PS + ((b & 1) * 4 + (r >> 3)) * RS + (r & 7) * VS + (lane >> 5) * 16;
    }
  }
  const int b_base = S16 ? ((lane >> 4) * BN + (lane & 15)) * 16 : ((lane >> 5) * BN + (lane & 31)) * 16;
  constexpr int NACC = S16 ? NA * 4 : MA * 2;
  f32x16 acc32[S16 ? 1 : NACC];
  f32x4 acc16[S16 ? NACC : 1];
#pragma unroll
  for (int m = 0; m < NACC; ++m) {
    if (S16) { for (int i = 0; i < 4; ++i) acc16[m][i] = 0.f; }
    else { for (int i = 0; i < 16; ++i) acc32[m][i] = 0.f; }
  }
  constexpr int WITEMS = SLAB / 16;   // 768
  constexpr int NWR = (WITEMS + NT - 1) / NT;
  f32x4 wreg[NWR];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int g = 0; g < nslabs; ++g) {
    const int sl = g % 9, kd = sl / 3, kh = sl % 3;
    if (STAGE == 1) {
      if (g > 0) {
        char* dst = wbuf + ((g + 1) & 1) * SLAB;
#pragma unroll
        for (int j = 0; j < NWR; ++j)
          if (tid + j * NT < WITEMS) *(f32x4*)(dst + (tid + j * NT) * 16) = wreg[j];
      }
      const char* s = wsrc + (long)((g + 2) % 36) * SLAB;
#pragma unroll
      for (int j = 0; j < NWR; ++j)
        if (tid + j * NT < WITEMS) wreg[j] = *(const f32x4*)(s + (tid + j * NT) * 16);
    }
    if (STAGE == 2) {       // LDS-DMA: 12 pieces of 1 KB, wave-uniform destination
      const char* s = wsrc + (long)((g + 1) % 36) * SLAB;
      char* dst = wbuf + ((g + 1) & 1) * SLAB;
      for (int p = wave; p < 12; p += nw)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + p * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
    }
    const char* wb = wbuf + (g & 1) * SLAB + b_base;
    if (S16) {
      f16x8 fa[2][NA], fb[2][4];
      auto ld = [&](int t, int b) {     // t = kw
#pragma unroll
        for (int m = 0; m < NA; ++m) fa[b][m] = *(const f16x8*)(halo + a_base[m] + kd * PS + kh * RS + t * VS);
#pragma unroll
        for (int q = 0; q < 4; ++q) fb[b][q] = *(const f16x8*)(wb + t * KG * BN * 16 + q * 16 * 16);
      };
      ld(0, 0);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        if (t + 1 < 3) ld(t + 1, (t + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < NA; ++m)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc16[m * 4 + q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[t & 1][m], fb[t & 1][q], acc16[m * 4 + q], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      constexpr int NSTEP = KSPLIT ? 3 : 6;
      f16x8 fa[2][MA], fb[2][2];
      auto ld = [&](int t, int b) {
        const int kw = KSPLIT ? t : t >> 1, ks = KSPLIT ? kp : t & 1;
#pragma unroll
        for (int m = 0; m < MA; ++m) fa[b][m] = *(const f16x8*)(halo + a_base[m] + kd * PS + kh * RS + kw * VS + ks * 32);
        fb[b][0] = *(const f16x8*)(wb + (kw * KG + 2 * ks) * BN * 16);
        fb[b][1] = *(const f16x8*)(wb + (kw * KG + 2 * ks) * BN * 16 + 32 * 16);
      };
      ld(0, 0);
#pragma unroll
      for (int t = 0; t < NSTEP; ++t) {
        if (t + 1 < NSTEP) ld(t + 1, (t + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MA; ++m) {
          acc32[m * 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[t & 1][m], fb[t & 1][0], acc32[m * 2], 0, 0, 0);
          acc32[m * 2 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[t & 1][m], fb[t & 1][1], acc32[m * 2 + 1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (BAR) __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < NACC; ++m) {
    if (S16) { for (int i = 0; i < 4; ++i) s += acc16[m][i]; }
    else { for (int i = 0; i < 16; ++i) s += acc32[m][i]; }
  }
  out[(long)blockIdx.x * NT + tid] = s;
  if (tid == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

// 16x16x32 shape with operands in registers
template <int NBLK, int NT>
__global__ __launch_bounds__(NT) void mfma16_only_kernel(const char* __restrict__ src, float* out, int nslabs, unsigned long long* clk) {
  const int tid = threadIdx.x;
  f16x8 fa[NBLK / 4], fb[4];
#pragma unroll
  for (int m = 0; m < NBLK / 4; ++m) fa[m] = *(const f16x8*)(src + ((tid * 8 + m) * 16) % (1 << 20));
#pragma unroll
  for (int q = 0; q < 4; ++q) fb[q] = *(const f16x8*)(src + 65536 * q + 4096 + tid * 16);
  f32x4 acc[NBLK];
#pragma unroll
  for (int m = 0; m < NBLK; ++m)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[m][i] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int g = 0; g < nslabs; ++g) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int m = 0; m < NBLK; ++m)
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[m / 4], fb[m % 4], acc[m], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < NBLK; ++m)
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[m][i];
  out[(long)blockIdx.x * NT + tid] = s;
  if (tid == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

// legacy 16x16x16 shape (K = 16: 4 input elements per lane), operands in registers: cycles per instruction on gfx950
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
template <int NBLK, int NT>
__global__ __launch_bounds__(NT) void mfma16k16_only_kernel(const char* __restrict__ src, float* out, int nslabs, unsigned long long* clk) {
  const int tid = threadIdx.x;
  f16x4 fa[NBLK / 4], fb[4];
#pragma unroll
  for (int m = 0; m < NBLK / 4; ++m) fa[m] = *(const f16x4*)(src + ((tid * 8 + m) * 16) % (1 << 20));
#pragma unroll
  for (int q = 0; q < 4; ++q) fb[q] = *(const f16x4*)(src + 65536 * q + 4096 + tid * 16);
  f32x4 acc[NBLK];
#pragma unroll
  for (int m = 0; m < NBLK; ++m)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[m][i] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int g = 0; g < nslabs; ++g) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int m = 0; m < NBLK; ++m)
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x16f16(fa[m / 4], fb[m % 4], acc[m], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < NBLK; ++m)
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[m][i];
  out[(long)blockIdx.x * NT + tid] = s;
  if (tid == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

// operands in registers: the MFMA-only ceiling at the clock the chip holds on random data
template <int MA, int NT>
__global__ __launch_bounds__(NT) void mfma_only_kernel(const char* __restrict__ src, float* out, int nslabs, unsigned long long* clk) {
  const int tid = threadIdx.x;
  f16x8 fa[MA], fb[2];
#pragma unroll
  for (int m = 0; m < MA; ++m) fa[m] = *(const f16x8*)(src + ((tid * MA + m) * 16) % (1 << 20));
  fb[0] = *(const f16x8*)(src + 4096 + tid * 16);
  fb[1] = *(const f16x8*)(src + 65536 + tid * 16);
  f32x16 acc[MA][2];
#pragma unroll
  for (int m = 0; m < MA; ++m)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][q][i] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int g = 0; g < nslabs; ++g) {
#pragma unroll
    for (int t = 0; t < 6; ++t) {
#pragma unroll
      for (int m = 0; m < MA; ++m) {
        acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[m], fb[0], acc[m][0], 0, 0, 0);
        acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[m], fb[1], acc[m][1], 0, 0, 0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < MA; ++m)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) s += acc[m][q][i];
  out[(long)blockIdx.x * NT + tid] = s;
  if (tid == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

struct Ctx { char* src; char* wsrc; float* out; unsigned long long* clk; hipEvent_t e0, e1; };

template <typename F>
static void run(const char* name, Ctx& c, int grid, int nt, double mfma_per_block, F launch) {
  for (int i = 0; i < 2; ++i) launch();
  CHECK(hipDeviceSynchronize());
  float best = 1e30f, sum = 0;
  const int R = 7;
  for (int i = 0; i < R; ++i) {
    CHECK(hipEventRecord(c.e0));
    launch();
    CHECK(hipEventRecord(c.e1));
    CHECK(hipEventSynchronize(c.e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, c.e0, c.e1));
    best = ms < best ? ms : best; sum += ms;
  }
  CHECK(hipGetLastError());
  std::vector<unsigned long long> h(grid * 2);
  CHECK(hipMemcpy(h.data(), c.clk, grid * 16, hipMemcpyDeviceToHost));
  std::vector<double> ghz;
  double cyc = 0;
  for (int b = 0; b < grid; ++b) { ghz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1); cyc += (double)h[2 * b]; }
  std::sort(ghz.begin(), ghz.end());
  const double fl = mfma_per_block * grid * 32768.0;
  const double med = sum / R;
  // cycles per MFMA per SIMD = block cycles / (MFMAs of the block / 4 SIMDs) / blocks per CU is left to the reader: print raw
  printf("%-44s grid %5d x %4d  med %8.1f us  best %8.1f us  %7.1f TF/s (best %7.1f)  clock %.2f GHz  cyc/blk %.0f\n", name, grid, nt,
         med * 1e3, best * 1e3, fl / (med * 1e-3) / 1e12, fl / (best * 1e-3) / 1e12, ghz[ghz.size() / 2], cyc / grid);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int nslabs = argc > 1 ? atoi(argv[1]) : 9 * 4 * 8;      // 8 tiles' worth of a 128-channel layer
  Ctx c;
  CHECK(hipMalloc(&c.src, 1 << 21));
  CHECK(hipMalloc(&c.wsrc, 40 * SLAB));
  CHECK(hipMalloc(&c.out, 4096 * 1024 * 4));
  CHECK(hipMalloc(&c.clk, 4096 * 16));
  CHECK(hipEventCreate(&c.e0)); CHECK(hipEventCreate(&c.e1));
  {
    std::vector<f16> h((1 << 21) / 2);
    srand(1);
    for (auto& v : h) v = (f16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
    CHECK(hipMemcpy(c.src, h.data(), 1 << 21, hipMemcpyHostToDevice));
    for (int i = 0; i < 40 * SLAB; i += 1 << 20) CHECK(hipMemcpy(c.wsrc + i, h.data(), std::min(1 << 20, 40 * SLAB - i), hipMemcpyHostToDevice));
  }
  const int CUS = 256;
#define LAUNCH_LOOP(MA, KS, BAR, ST, NT, S16, GRID, HALO, LDSB)                                                                \
  {                                                                                                                      \
    CHECK(hipFuncSetAttribute((const void*)loop_kernel<MA, KS, BAR, ST, NT, S16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB)); \
    char nm[96];                                                                                                         \
    snprintf(nm, sizeof nm, "loop %s MA=%d ksplit=%d bar=%d stage=%d nt=%d lds=%dK", S16 ? "16x16x32" : "32x32x16", MA, KS, BAR, ST, NT, (LDSB) / 1024);        \
    run(nm, c, GRID, NT, (double)nslabs * (KS ? 3 : 6) * MA * 2 * (NT / 64), [&]() {                                     \
      hipLaunchKernelGGL((loop_kernel<MA, KS, BAR, ST, NT, S16>), dim3(GRID), dim3(NT), LDSB, 0, c.src, c.wsrc, c.out, nslabs, HALO, c.clk); \
    });                                                                                                                  \
  }
#define LAUNCH_MFMA(MA, NT, GRID)                                                                                      \
  {                                                                                                                    \
    char nm[96];                                                                                                       \
    snprintf(nm, sizeof nm, "mfma-only MA=%d nt=%d", MA, NT);                                                            \
    run(nm, c, GRID, NT, (double)nslabs * 6 * MA * 2 * (NT / 64), [&]() {                                               \
      hipLaunchKernelGGL((mfma_only_kernel<MA, NT>), dim3(GRID), dim3(NT), 0, 0, c.src, c.out, nslabs, c.clk);           \
    });                                                                                                                \
  }
  const int H6 = 6 * Geo<0>::PS, H8 = 8 * Geo<0>::PS, H10 = 10 * Geo<0>::PS, G6 = 6 * Geo<1>::PS, G10 = 10 * Geo<1>::PS;
  printf("nslabs %d\n", nslabs);
  LAUNCH_MFMA(2, 256, CUS * 2)        // 2 waves per SIMD
  LAUNCH_MFMA(4, 512, CUS)            // 2 waves per SIMD
  {
    run("mfma16-only 64x64 nt=256 x2", c, CUS * 2, 256, (double)nslabs * 3 * 16 * 4 * 0.5, [&]() {
      hipLaunchKernelGGL((mfma16_only_kernel<16, 256>), dim3(CUS * 2), dim3(256), 0, 0, c.src, c.out, nslabs, c.clk); });
    run("mfma16-only 64x64 nt=512", c, CUS, 512, (double)nslabs * 3 * 16 * 8 * 0.5, [&]() {
      hipLaunchKernelGGL((mfma16_only_kernel<16, 512>), dim3(CUS), dim3(512), 0, 0, c.src, c.out, nslabs, c.clk); });
  }
  if (argc > 3) {   // cycles of the legacy K = 16 instruction against the K = 32 one (same count of instructions per block)
    run("mfma16x16x32-only nt=256 x2 (FLOP as printed)", c, CUS * 2, 256, (double)nslabs * 3 * 16 * 4 * 0.5, [&]() {
      hipLaunchKernelGGL((mfma16_only_kernel<16, 256>), dim3(CUS * 2), dim3(256), 0, 0, c.src, c.out, nslabs, c.clk); });
    run("mfma16x16x16-only nt=256 x2 (HALF the FLOP printed)", c, CUS * 2, 256, (double)nslabs * 3 * 16 * 4 * 0.5, [&]() {
      hipLaunchKernelGGL((mfma16k16_only_kernel<16, 256>), dim3(CUS * 2), dim3(256), 0, 0, c.src, c.out, nslabs, c.clk); });
    return 0;
  }
  if (argc > 2) {   // round 5: the wide kernel's shape -- 2 workgroups x 4 waves per CU, 128x64 wave tiles (depth slices alias planes 0..3)
    LAUNCH_LOOP(4, 0, 1, 2, 256, 0, CUS * 2, H6, H6 + 2 * SLAB)
    LAUNCH_LOOP(4, 0, 1, 2, 256, 1, CUS * 2, G6, G6 + 2 * SLAB)
    LAUNCH_LOOP(4, 0, 1, 0, 256, 0, CUS * 2, H6, H6 + 2 * SLAB)
    LAUNCH_LOOP(4, 0, 1, 0, 256, 1, CUS * 2, G6, G6 + 2 * SLAB)
    LAUNCH_LOOP(4, 0, 1, 2, 256, 0, CUS * 2, H6, H6 + 2 * SLAB)
    LAUNCH_LOOP(4, 0, 1, 2, 256, 1, CUS * 2, G6, G6 + 2 * SLAB)
    return 0;
  }
  // today's kernel shape: 2 workgroups x 4 waves per CU, 64x64 wave tiles
  LAUNCH_LOOP(2, 0, 1, 0, 256, 0, CUS * 2, H6, H6 + 2 * SLAB)
  LAUNCH_LOOP(2, 0, 1, 1, 256, 0, CUS * 2, H6, H6 + 2 * SLAB)
  LAUNCH_LOOP(2, 0, 1, 2, 256, 0, CUS * 2, H6, H6 + 2 * SLAB)
  // same with the 16x16x32 shape
  LAUNCH_LOOP(2, 0, 1, 0, 256, 1, CUS * 2, G6, G6 + 2 * SLAB)
  LAUNCH_LOOP(2, 0, 1, 1, 256, 1, CUS * 2, G6, G6 + 2 * SLAB)
  LAUNCH_LOOP(2, 0, 1, 2, 256, 1, CUS * 2, G6, G6 + 2 * SLAB)
  // one workgroup of 8 waves, 64x64 wave tiles (8x8x8 tile)
  LAUNCH_LOOP(2, 0, 1, 0, 512, 0, CUS, H10, H10 + 2 * SLAB)
  LAUNCH_LOOP(2, 0, 1, 1, 512, 0, CUS, H10, H10 + 2 * SLAB)
  LAUNCH_LOOP(2, 0, 1, 2, 512, 0, CUS, H10, H10 + 2 * SLAB)
  LAUNCH_LOOP(2, 0, 1, 0, 512, 1, CUS, G10, G10 + 2 * SLAB)
  LAUNCH_LOOP(2, 0, 1, 1, 512, 1, CUS, G10, G10 + 2 * SLAB)
  LAUNCH_LOOP(2, 0, 1, 2, 512, 1, CUS, G10, G10 + 2 * SLAB)
  // 128x64 wave tiles, pairs of waves split K (8 waves, two per SIMD)
  LAUNCH_LOOP(4, 1, 1, 1, 512, 0, CUS, H10, 2 * H10 + 2 * SLAB)
  LAUNCH_LOOP(4, 1, 1, 2, 512, 0, CUS, H10, 2 * H10 + 2 * SLAB)
  // 96x64 wave tiles: 8 waves with two 6x8x8 tiles
  LAUNCH_LOOP(3, 0, 1, 1, 512, 0, CUS, 2 * H8, 2 * H8 + 2 * SLAB)
  LAUNCH_LOOP(3, 0, 1, 2, 512, 0, CUS, 2 * H8, 2 * H8 + 2 * SLAB)
  return 0;
}
