// Same-run calibration of the box (bench.py `roofline.sustained_probe_tflops`): what this chip sustains on NOTHING BUT back-to-back
// MFMAs on random bf16 operands, by instruction shape.  MI355X is power-limited under dense MFMA streams: the
// guide's 2.5 PFLOP/s is never reached on non-zero operands, boxes differ by 4-5 % at identical code, and the sustained rate of
// the two shapes the kernels use (32x32x16: attention, 16x16x32: GEMM / VAE) is the ceiling a kernel's wall clock is priced
// against next to the contractual fraction of the nominal peak.  One wave per SIMD, accumulators in the accumulator file -- the
// register arrangement of attn_w64_kernel and gemm_bf16_v8_kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/mmpl_hip.h"

extern int mmpl_set_error(const char* where, const char* what);  // api.hip

namespace {
typedef __attribute__((ext_vector_type(8))) short pbf16x8;
typedef __attribute__((ext_vector_type(4))) float pf32x4;
typedef __attribute__((ext_vector_type(16))) float pf32x16;

template <int SHAPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void probe_mfma_kernel(const pbf16x8* in, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  pbf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(lane + 64 * i) & 1023]; b[i] = in[(lane * 7 + 64 * i + 13) & 1023]; }
  float r = 0.f;
  if constexpr (SHAPE == 32) {
    pf32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[u]), "v"(b[i & 3]));
    }
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) r += acc[i][j];
  } else {
    pf32x4 acc[32];
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[(i >> 3) ^ u]), "v"(b[i & 3]));
    }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 4; ++j) r += acc[i][j];
  }
  if (r == 12345.678f) out[0] = r;                       // keeps the accumulators live; never true
}
}  // namespace

extern "C" int mmpl_probe_mfma_tflops(int shape, double seconds, double* tflops) {
  if ((shape != 32 && shape != 16) || !tflops || !(seconds > 0.0) || seconds > 30.0) return mmpl_set_error("mmpl_probe_mfma_tflops", "shape must be 32 or 16, 0 < seconds <= 30");
  std::vector<uint16_t> h(1024 * 8);
  uint32_t s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 + ((s >> 9) & 0x3ff) + ((s >> 31) << 15)); }   // random sign, full mantissa, |x| in [0.0078, 0.03]
  pbf16x8* in = nullptr;
  float* out = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipMalloc((void**)&in, h.size() * 2);
  if (e == hipSuccess) e = hipMalloc((void**)&out, 64);
  if (e == hipSuccess) e = hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  const int blocks = 256 * 4, iters = 2000;              // one launch ~ 30 ms
  const double flops_per_launch = (double)blocks * 4.0 * iters * 32.0 * 32768.0;   // both shapes: 32 x 32768 = 64 x 16384 FLOP per wave per iteration
  double total_ms = 0.0, tail_ms = 0.0;
  int tail_launches = 0;
  // launches in groups of 4 until `seconds` have passed; the rate reported is that of the second half (clocks / power settled)
  while (e == hipSuccess && total_ms < seconds * 1e3) {
    e = hipEventRecord(e0, nullptr);
    for (int k = 0; k < 4 && e == hipSuccess; ++k) {
      if (shape == 32) hipLaunchKernelGGL(probe_mfma_kernel<32>, dim3(blocks), dim3(256), 0, nullptr, in, out, iters);
      else hipLaunchKernelGGL(probe_mfma_kernel<16>, dim3(blocks), dim3(256), 0, nullptr, in, out, iters);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    total_ms += ms;
    if (total_ms >= seconds * 0.5e3) { tail_ms += ms; tail_launches += 4; }
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (in) (void)hipFree(in);
  if (out) (void)hipFree(out);
  if (e != hipSuccess) return mmpl_set_error("mmpl_probe_mfma_tflops", hipGetErrorString(e));
  *tflops = tail_launches ? flops_per_launch * tail_launches / (tail_ms * 1e-3) / 1e12 : 0.0;
  return 0;
}
