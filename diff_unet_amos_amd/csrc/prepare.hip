// Host-side once-per-device preparation of the library: dynamic-LDS limits of every kernel that needs more than the
// default 64 KB, and the device's compute-unit count.  See common.hpp (LdsAttrs / ensure_prepared).
//
// Why it is central: the launchers used to raise "their" limit lazily through unsynchronised function-local flags.  The
// training step runs its backward launches on autograd's worker thread, and a whole step is captured into a HIP graph; a
// lazily executed hipFuncSetAttribute could therefore run for the first time inside a capture or on two threads.  Now the
// first launcher call on a device (or dua_prepare(), which plans and trainers call at construction) does all of it, once.
#include <atomic>
#include <mutex>
#include <vector>
#include "common.hpp"
#include "../../include/dua_hip.h"

namespace dua {

namespace {
std::mutex& reg_mutex() { static std::mutex m; return m; }
std::vector<LdsAttr>& registry() { static std::vector<LdsAttr> r; return r; }
std::atomic<int> g_ready[64];
std::atomic<int> g_cus[64];
}  // namespace

void register_lds_attrs(const LdsAttr* list, int n) {
  std::lock_guard<std::mutex> lock(reg_mutex());
  for (int i = 0; i < n; ++i) registry().push_back(list[i]);
}

int ensure_prepared() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return DUA_ERR_ARG;
  if (g_ready[dev].load(std::memory_order_acquire)) return 0;
  std::lock_guard<std::mutex> lock(reg_mutex());
  if (g_ready[dev].load(std::memory_order_acquire)) return 0;
  for (const LdsAttr& a : registry()) {
    hipError_t e = hipFuncSetAttribute(a.fn, hipFuncAttributeMaxDynamicSharedMemorySize, a.bytes);
    if (e != hipSuccess) return (int)e;
  }
  int cus = 0;
  hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (e != hipSuccess) return (int)e;
  g_cus[dev].store(cus, std::memory_order_relaxed);
  g_ready[dev].store(1, std::memory_order_release);
  return 0;
}

int device_cus() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  if (!g_ready[dev].load(std::memory_order_acquire)) return 0;
  return g_cus[dev].load(std::memory_order_relaxed);
}

}  // namespace dua

extern "C" {

int dua_abi_version(void) { return DUA_ABI_VERSION; }

int dua_prepare(void) { return dua::ensure_prepared(); }

int dua_prepared_kernels(void) {
  std::lock_guard<std::mutex> lock(dua::reg_mutex());
  return (int)dua::registry().size();
}

}  // extern "C"
