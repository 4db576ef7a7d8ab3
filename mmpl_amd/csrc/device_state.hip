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
int g_xcd_dispatch[64];      // 0 = not probed yet, 1 = workgroup b of a launch runs on XCD b & 7, -1 = it does not

// every workgroup reports the XCC it runs on (HW_REG_XCC_ID[3:0])
__global__ void xcc_probe_kernel(int* out) {
  if (threadIdx.x == 0) {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
    out[blockIdx.x] = x;
  }
}
}  // namespace

// Workgroups of a launch are dealt round-robin to the 8 XCDs.  The tile / head orders rely on that for L2 LOCALITY only: since
// round 4 nothing relies on it for visibility (the split-K GEMM launch exchanges its partials with system-scope accesses,
// gemm.hip).  The probe stays as a diagnostic (C ABI mmpl_device_xcd_round_robin): once per device, outside any stream capture.
bool mmpl_xcd_dispatch_ok(bool may_probe) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_xcd_dispatch[dev] == 0 && may_probe) {
    constexpr int n = 2048;
    int* d = nullptr;
    int h[n];
    bool ok = hipMalloc(&d, n * sizeof(int)) == hipSuccess;
    if (ok) {
      ok = hipMemset(d, 0xff, n * sizeof(int)) == hipSuccess;
      hipLaunchKernelGGL(xcc_probe_kernel, dim3(n), dim3(64), 0, 0, d);
      ok = ok && hipGetLastError() == hipSuccess;
      ok = hipMemcpy(h, d, n * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess && ok;
      (void)hipFree(d);
    }
    unsigned seen = 0;
    for (int x = 0; ok && x < 8; ++x) {
      ok = h[x] >= 0 && h[x] < 8 && !(seen & (1u << h[x]));     // 8 residues, 8 different XCCs
      if (ok) seen |= 1u << h[x];
    }
    for (int b = 8; ok && b < n; ++b) ok = h[b] == h[b & 7];
    g_xcd_dispatch[dev] = ok ? 1 : -1;
  }
  return g_xcd_dispatch[dev] == 1;
}

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
    c.gemm_v1 = flag("MMPL_GEMM_V1"); c.gemm_v2 = flag("MMPL_GEMM_V2");
    c.gemm_no_splitk = flag("MMPL_GEMM_NO_SPLITK"); c.gemm_no_subtile = flag("MMPL_GEMM_NO_SUBTILE");
    c.gemm_group = num("MMPL_GEMM_GROUP", 0);
    c.gemm_pf = num("MMPL_GEMM_PF", 2);
    c.gemm_v8 = num("MMPL_GEMM_V8", -1);
    c.vae_no_fuse_norm = flag("MMPL_VAE_NO_FUSE_NORM");
    c.ln_pipeline_min_rows = num("MMPL_LN_PIPELINE_MIN_ROWS", 16384);
    c.check_share = flag("MMPL_CHECK_SHARE");
    return c;
  }();
  return cfg;
}
