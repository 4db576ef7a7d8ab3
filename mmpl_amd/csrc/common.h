// Shared device helpers for the mmpl_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define MMPL_DEV __device__ __forceinline__

MMPL_DEV float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even fp32 -> bf16 (same rounding as PyTorch's c10::BFloat16); gfx950 has the hardware
// conversion v_cvt_pk_bf16_f32, which the compiler emits for these casts.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16v2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2;
MMPL_DEV uint32_t pack2bf(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){lo, hi}, bf16v2_t));
}
MMPL_DEV bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
MMPL_DEV float rbf(float f) { return bf2f(f2bf(f)); }  // round through bf16

MMPL_DEV float gelu_tanh(float x) {
  // torch GELU(approximate='tanh'): 0.5*x*(1+tanh(u)), u = sqrt(2/pi)*(x+0.044715*x^3); 0.5*(1+tanh(u)) == 1/(1+exp(-2u)) ==
  // 1/(1+2^a) with a = x*(c1 + c2*x^2), c1 = -2*sqrt(2/pi)*log2(e), c2 = 0.044715*c1: 3 multiplies, one FMA, one add, v_exp_f32
  // (which IS 2^a) and v_rcp_f32 (1 ulp) instead of the IEEE division sequence (v_div_scale x2, v_rcp, 4 FMAs, v_div_fmas,
  // v_div_fixup: 10 VALU instructions per value, 128 values per lane in a GEMM epilogue).  The value is rounded to bf16 right
  // after, 2^15 times coarser than the difference.
  constexpr float c1 = -2.0f * 0.7978845608028654f * 1.4426950408889634f, c2 = 0.044715f * c1;
  const float a = x * __builtin_fmaf(x * x, c2, c1);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a));
}
MMPL_DEV float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

MMPL_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

struct uint4_ { uint32_t x, y, z, w; };
