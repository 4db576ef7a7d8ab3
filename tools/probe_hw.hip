// Hardware probe (dev tool, not product code): checks the gfx950 MFMA fragment layouts and the
// ds_read_b64_tr_b16 / global_load_lds semantics that mmpl_amd/csrc kernels assume.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_hw.hip -o gpurun_out/probe_hw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
static inline unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// A: [32][16] row-major bf16, B: [16][32] row-major (k,n), C: [32][32]
__global__ void k_mfma32(const unsigned short* A, const unsigned short* B, float* C) {
  int l = threadIdx.x; bf16x8 a, b;
  for (int j = 0; j < 8; j++) { a[j] = A[(l & 31) * 16 + 8 * (l >> 5) + j]; b[j] = B[(8 * (l >> 5) + j) * 32 + (l & 31)]; }
  f32x16 c = {0}; c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; r++) { int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31; C[row * 32 + col] = c[r]; }
}
// A: [16][32], B: [32][16] (k,n), C [16][16]
__global__ void k_mfma16(const unsigned short* A, const unsigned short* B, float* C) {
  int l = threadIdx.x; bf16x8 a, b;
  for (int j = 0; j < 8; j++) { a[j] = A[(l & 15) * 32 + 8 * (l >> 4) + j]; b[j] = B[(8 * (l >> 4) + j) * 16 + (l & 15)]; }
  f32x4 c = {0}; c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) { int row = (l >> 4) * 4 + r, col = l & 15; C[row * 16 + col] = c[r]; }
}
// tr read probe: LDS[i] = i (ushort).  Each lane supplies a byte address; dump the 4 ushorts it gets back.
__global__ void k_tr(unsigned short* out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  int l = threadIdx.x; unsigned addr;
  if (mode == 0) addr = l * 8;                                   // lane-linear 8 B
  else if (mode == 1) addr = (l & 15) * 8 + (l >> 4) * 512;      // 16 lanes contiguous, groups 512 B apart
  else if (mode == 2) addr = (l & 3) * 8 + ((l >> 2) & 3) * 256 + (l >> 4) * 2048;  // 4 rows of 32 B, row stride 256 B
  else addr = (l & 15) * 256 + (l >> 4) * 8;                      // each lane its own row (stride 256 B)
  unsigned base = (unsigned)(size_t)(&lds[0]);
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr + base) : "memory");
  out[l * 4 + 0] = (unsigned short)(v & 0xffff); out[l * 4 + 1] = (unsigned short)((v >> 16) & 0xffff);
  out[l * 4 + 2] = (unsigned short)((v >> 32) & 0xffff); out[l * 4 + 3] = (unsigned short)((v >> 48) & 0xffff);
}
// global_load_lds 16 B probe: 64 lanes, per-lane global source, LDS dest = uniform base + lane*16
__global__ void k_glds(const unsigned* src, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[512];
  for (int i = threadIdx.x; i < 512; i += 64) lds[i] = 0xdeadbeef;
  __syncthreads();
  int l = threadIdx.x;
  const unsigned* g = src + ((l * 7) % 64) * 4;  // permuted source
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)&lds[64], 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
// permlane32_swap probe
__global__ void k_perm(unsigned* out) {
  int l = threadIdx.x; unsigned a = 1000 + l, b = 2000 + l;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[l * 2] = r[0]; out[l * 2 + 1] = r[1];
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s arch %s CUs %d clock %d kHz mem %.1f GB\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate, p.totalGlobalMem / 1e9);
  {  // mfma 32x32x16
    std::vector<unsigned short> A(32 * 16), B(16 * 32); std::vector<float> Af(32 * 16), Bf(16 * 32), C(32 * 32), R(32 * 32, 0.f);
    for (int i = 0; i < 32 * 16; i++) { Af[i] = (float)((i * 7 + 3) % 13 - 6); A[i] = f2bf(Af[i]); }
    for (int i = 0; i < 16 * 32; i++) { Bf[i] = (float)((i * 5 + 1) % 11 - 5); B[i] = f2bf(Bf[i]); }
    for (int m = 0; m < 32; m++) for (int n = 0; n < 32; n++) for (int k = 0; k < 16; k++) R[m * 32 + n] += Af[m * 16 + k] * Bf[k * 32 + n];
    unsigned short *dA, *dB; float* dC; CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dC, C.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    k_mfma32<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize()); CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 1024; i++) if (C[i] != R[i]) bad++;
    printf("MFMA32x32x16 layout check: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  }
  {  // mfma 16x16x32
    std::vector<unsigned short> A(16 * 32), B(32 * 16); std::vector<float> Af(16 * 32), Bf(32 * 16), C(256), R(256, 0.f);
    for (int i = 0; i < 512; i++) { Af[i] = (float)((i * 7 + 3) % 13 - 6); A[i] = f2bf(Af[i]); Bf[i] = (float)((i * 5 + 1) % 11 - 5); B[i] = f2bf(Bf[i]); }
    for (int m = 0; m < 16; m++) for (int n = 0; n < 16; n++) for (int k = 0; k < 32; k++) R[m * 16 + n] += Af[m * 32 + k] * Bf[k * 16 + n];
    unsigned short *dA, *dB; float* dC; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 1024));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
    k_mfma16<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize()); CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 256; i++) if (C[i] != R[i]) bad++;
    printf("MFMA16x16x32 layout check: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  }
  for (int mode = 0; mode < 4; mode++) {
    unsigned short* d; CK(hipMalloc(&d, 64 * 4 * 2)); std::vector<unsigned short> h(256);
    k_tr<<<1, 64>>>(d, mode); CK(hipDeviceSynchronize()); CK(hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost));
    printf("tr_b16 mode %d (lane: 4 elems):\n", mode);
    for (int l = 0; l < 64; l++) { printf(" [%2d]%4d,%4d,%4d,%4d", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]); if (l % 4 == 3) printf("\n"); }
  }
  {
    std::vector<unsigned> s(256), o(512); for (int i = 0; i < 256; i++) s[i] = i;
    unsigned *ds, *dout; CK(hipMalloc(&ds, 1024)); CK(hipMalloc(&dout, 2048)); CK(hipMemcpy(ds, s.data(), 1024, hipMemcpyHostToDevice));
    k_glds<<<1, 64>>>(ds, dout); CK(hipDeviceSynchronize()); CK(hipMemcpy(o.data(), dout, 2048, hipMemcpyDeviceToHost));
    int bad = 0; for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++) if (o[64 + l * 4 + j] != (unsigned)(((l * 7) % 64) * 4 + j)) bad++;
    printf("global_load_lds x16 lane-linear dest: %s (%d mismatches) guard before=%x after=%x\n", bad ? "FAIL" : "OK", bad, o[63], o[64 + 256]);
  }
  {
    unsigned* d; CK(hipMalloc(&d, 512)); std::vector<unsigned> h(128);
    k_perm<<<1, 64>>>(d); CK(hipDeviceSynchronize()); CK(hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost));
    printf("permlane32_swap(a=1000+l, b=2000+l): lane0 r=(%u,%u) lane31 r=(%u,%u) lane32 r=(%u,%u) lane63 r=(%u,%u)\n", h[0], h[1], h[62], h[63], h[64], h[65], h[126], h[127]);
  }
  return 0;
}
