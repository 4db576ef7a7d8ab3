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


// ---------------------------------------------------------------- run-time switches (mmpl_config.h)
#include <stdlib.h>

#include "mmpl_config.h"
const MmplRuntimeConfig& mmpl_config() {
  static const MmplRuntimeConfig cfg = [] {
    auto flag = [](const char* n) { const char* v = getenv(n); return v != nullptr && atoi(v) != 0; };
    auto num = [](const char* n, int dflt) { const char* v = getenv(n); return v ? atoi(v) : dflt; };
    MmplRuntimeConfig c{};
    c.attn_v1 = flag("MMPL_ATTN_V1"); c.attn_nosplit = flag("MMPL_ATTN_NOSPLIT"); c.attn_no_merge = flag("MMPL_ATTN_NO_MERGE");
    c.cross_w64 = flag("MMPL_CROSS_W64");
    c.gemm_v1 = flag("MMPL_GEMM_V1"); c.gemm_v2 = flag("MMPL_GEMM_V2"); c.gemm_direct_epilogue = flag("MMPL_GEMM_DIRECT_EPILOGUE");
    c.gemm_static_tiles = flag("MMPL_GEMM_STATIC_TILES"); c.gemm_no_sync_sweeps = flag("MMPL_GEMM_NO_SYNC_SWEEPS");
    c.gemm_group = num("MMPL_GEMM_GROUP", 0);
    c.gemm_pf = num("MMPL_GEMM_PF", 2);
    c.vae_no_halo = flag("MMPL_VAE_NO_HALO");
    return c;
  }();
  return cfg;
}
