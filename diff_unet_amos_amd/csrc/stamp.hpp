// Diagnostic build only (-DDUA_STAMP, tools/build_diag.sh -> a separate library selected with DUA_HIP_LIB): lane 0 of every
// workgroup stamps the shader-clock counter at the phase boundaries of a convolution kernel and the 100 MHz real-time counter
// at its start and end.  The stamps go to a buffer of this translation unit's own that no kernel reads (stamps_out copies it
// out and clears it); the shipped library contains none of this.
#pragma once
#include <hip/hip_runtime.h>
#ifdef DUA_STAMP
namespace dua {
constexpr int STAMP_WGS = 8192, STAMP_SLOTS = 64;
static __device__ unsigned long long g_stamp[STAMP_WGS][STAMP_SLOTS];
static inline long stamps_out(void* host, long bytes) {
  const long all = (long)sizeof(unsigned long long) * STAMP_WGS * STAMP_SLOTS;
  if (!host || bytes < all) return all;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamp), all) != hipSuccess) return -1;
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_stamp)) == hipSuccess) (void)hipMemset(p, 0, all);
  return all;
}
}  // namespace dua
#define DUA_STAMP_AT(slot, rt)                                                                          \
  do {                                                                                                   \
    if (threadIdx.x == 0) {                                                                              \
      const int wg_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;                    \
      if (wg_ < dua::STAMP_WGS && (slot) < dua::STAMP_SLOTS)                                            \
        dua::g_stamp[wg_][slot] = (rt) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); \
    }                                                                                                    \
  } while (0)
#else
#define DUA_STAMP_AT(slot, rt) do { } while (0)
#endif
