// Helpers shared by the one-wave-per-SIMD kernels (attn_w64.hip): compile-time loops whose bodies are
// `asm volatile` statements with literal register numbers, and the clobber list that hands the whole accumulator file to them.
#pragma once
#include <type_traits>
#include <utility>

#include "common.h"

namespace w64 {
template <int I> using ic = std::integral_constant<int, I>;
template <class F, int... I> MMPL_DEV void sfor_(F&& f, std::integer_sequence<int, I...>) { (f(ic<I>{}), ...); }
// static for: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>), fully unrolled, indices usable as "i" operands
template <int N, class F> MMPL_DEV void sfor(F&& f) { sfor_(f, std::make_integer_sequence<int, N>{}); }
}  // namespace w64

#define MMPL_A10(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
// One `asm volatile("s_nop 0" ::: MMPL_ALL_AGPRS)` makes the kernel descriptor allocate all 256 accumulator registers; the
// kernels then name them literally (a[0:15] ...) and tests/test_isa_audit.py checks that the compiler never touches them.
#define MMPL_ALL_AGPRS                                                                                                            \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", MMPL_A10(1), MMPL_A10(2), MMPL_A10(3), MMPL_A10(4), MMPL_A10(5),    \
      MMPL_A10(6), MMPL_A10(7), MMPL_A10(8), MMPL_A10(9), MMPL_A10(10), MMPL_A10(11), MMPL_A10(12), MMPL_A10(13), MMPL_A10(14),   \
      MMPL_A10(15), MMPL_A10(16), MMPL_A10(17), MMPL_A10(18), MMPL_A10(19), MMPL_A10(20), MMPL_A10(21), MMPL_A10(22),             \
      MMPL_A10(23), MMPL_A10(24), "a250", "a251", "a252", "a253", "a254", "a255"
