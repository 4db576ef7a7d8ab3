// HW probe (dev tool): semantics of `buffer_load_dwordx4 ... offen lds` on gfx950 that attn_w64.hip relies on --
//   (1) out-of-range lanes (voffset + inst_offset >= num_records [- soffset?]) write ZEROS to LDS, no fault
//   (2) whether soffset takes part in the range check
//   (3) the LDS destination is M0 + inst_offset + 16 * lane, for M0 above 64 KiB too
//   hipcc --offload-arch=gfx950 -O2 tools/probe_bufdma.hip -o tools/build/probe_bufdma && tools/build/probe_bufdma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__global__ void probe(const uint32_t* src, uint32_t* out, uint32_t num_records, uint32_t soff, uint32_t m0_base, int use_imm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 160 * 1024 / 4; i += 64) ((uint32_t*)smem)[i] = 0xdeadbeefu;
  __syncthreads();
  u32x4 srd;
  const uint64_t p = (uint64_t)src;
  srd[0] = __builtin_amdgcn_readfirstlane((uint32_t)p);
  srd[1] = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32) & 0xffffu);
  srd[2] = __builtin_amdgcn_readfirstlane(num_records);
  srd[3] = 0x00020000u;
  const uint32_t voff = lane * 16;
  const uint32_t s = __builtin_amdgcn_readfirstlane(soff), m = __builtin_amdgcn_readfirstlane(m0_base);
  if (use_imm)
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2 offen offset:1024 lds\n\ts_waitcnt vmcnt(0)" ::"v"(voff), "s"(srd), "s"(s), "s"(m) : "memory");
  else
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_waitcnt vmcnt(0)" ::"v"(voff), "s"(srd), "s"(s), "s"(m) : "memory");
  __syncthreads();
  const uint32_t base = m0_base + (use_imm ? 1024 : 0);
  for (int i = lane; i < 256; i += 64) out[i] = ((uint32_t*)(smem + base))[i];
  if (lane == 0) out[256] = ((uint32_t*)(smem + m0_base))[0];          // untouched when use_imm (destination moved by the offset)
}

int main() {
  const int N = 4096;
  std::vector<uint32_t> h(N);
  for (int i = 0; i < N; ++i) h[i] = 0x1000000u + i;
  uint32_t *src, *out;
  hipMalloc(&src, N * 4); hipMalloc(&out, 1028 * 4);
  hipMemcpy(src, h.data(), N * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  struct { uint32_t nr, soff, m0; int imm; const char* what; } cases[] = {
      {1024, 0, 0, 0, "num_records 1024, soffset 0: all 64 lanes in range"},
      {512, 0, 0, 0, "num_records 512: lanes >= 32 out of range"},
      {1024, 512, 0, 0, "num_records 1024, soffset 512: is soffset range-checked?"},
      {4096, 0, 100 * 1024, 0, "M0 = 100 KiB"},
      {4096, 0, 100 * 1024, 1, "M0 = 100 KiB, inst offset 1024 (moves BOTH addresses?)"},
      {1536, 0, 0, 1, "num_records 1536, inst offset 1024: lanes >= 32 out of range if the check includes inst_offset"},
  };
  for (auto& c : cases) {
    hipMemset(out, 0, 1028 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 160 * 1024, 0, src, out, c.nr, c.soff, c.m0, c.imm);
    hipError_t e = hipDeviceSynchronize();
    uint32_t r[257];
    hipMemcpy(r, out, 257 * 4, hipMemcpyDeviceToHost);
    printf("%s (%s)\n  lane0 %08x %08x | lane1 %08x | lane31 %08x | lane32 %08x | lane63 %08x %08x  | word at M0: %08x\n", c.what, hipGetErrorString(e), r[0], r[1],
           r[4], r[124], r[128], r[252], r[255], r[256]);
  }
  return 0;
}
