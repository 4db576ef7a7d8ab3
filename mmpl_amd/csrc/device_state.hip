// Per-device launcher state.  Function attributes (the > 64 KiB dynamic-LDS limit) and the CU count are properties of a
// DEVICE, and several host threads may drive several GPUs from one process (the reference's own thread-per-GPU model,
// Wan_fps_inference_parallel_4gpu_20s.py:229-256), so both are cached per device id under a mutex.
#include <mutex>
#include <set>
#include <utility>

#include "kernels.h"

namespace {
std::mutex g_mu;
std::set<std::pair<int, const void*>> g_attr_done;
int g_per_xcd[64];
}  // namespace

hipError_t mmpl_dyn_smem_once(const void* func, int bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_attr_done.count({dev, func})) return hipSuccess;
  e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) g_attr_done.insert({dev, func});
  return e;
}

int mmpl_cus_per_xcd() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 32;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_per_xcd[dev]) {
    hipDeviceProp_t prop;
    g_per_xcd[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8 ? prop.multiProcessorCount / 8 : 32;
  }
  return g_per_xcd[dev];
}
