// bf16 GEMM with fused epilogues for the Wan DiT linears:  C[M,N] = epi( A[M,K] . W[N,K]^T + bias[N] ).
//
// A is token-major activations, W is an nn.Linear weight ([out, in], K contiguous) -- both operands are
// K-contiguous, which is exactly the MFMA A/B fragment shape, so no transposes anywhere.
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 fragments of
// v_mfma_f32_16x16x32_bf16.  Operands are staged global -> registers -> XOR-swizzled LDS (conflict-free
// ds_read_b128), double-buffered with the next tile's global loads in flight under the
// current tile's MFMAs; one barrier per K tile.  The MFMA is issued "swapped" (W fragment as the A
// operand) so each lane ends up with 4 consecutive output columns -> 8-byte epilogue loads/stores.
// Rounding points of the epilogues follow the reference's bf16 module boundaries (DESIGN.md "numerics").
#include <stdlib.h>

#include "common.h"
#include "dev_knobs.h"
#include "kernels.h"
#include "w64_util.h"
#include "mmpl_config.h"

// per-phase shader-cycle counters of the large-problem kernels: development builds with -DGEMM6_TIMING=1 only (dev_knobs.h)
#ifdef MMPL_DEV_ABLATIONS
#define GEMM_DEV_TICKS [[maybe_unused]] unsigned long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0
#define GEMM_DEV_TICK(x) do { if constexpr (GEMM6_TIMING) x = __builtin_readcyclecounter(); } while (0)
#else
#define GEMM_DEV_TICKS do { } while (0)
#define GEMM_DEV_TICK(x) do { } while (0)
#endif

namespace {
using w64::sfor;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

// Shared epilogue: wave tile 64x64 at (mw, nw); lane holds, per fragment (i,j), column m = ..+(lane&15) and rows
// n = ..+4*(lane>>4)+{0..3} (the MFMA is issued with the W fragment as the A operand).
template <int EPI>
MMPL_DEV void gemm_epilogue(const GemmArgs& g, const f32x4 (&acc)[4][4], int mw, int nw, int frow, int fchunk) {
  const int m0 = 0, n0 = 0, wm = 0, wn = 0;
  (void)m0; (void)n0; (void)wm; (void)wn;
  // ---- epilogue: lane holds, per fragment (i,j), column m = ..+(lane&15) and rows n = ..+4*(lane>>4)+{0..3}.
  // Pass 1 issues every load of the tile (bias, residual, gate) before pass 2 stores anything: the residual usually IS the
  // output buffer (x += ...), so with loads and stores interleaved per fragment hipcc must keep each load behind the previous
  // store and the tile's epilogue becomes 16 dependent HBM round trips; like this they are all in flight at once.
  uint2 bias4[4], res4[4][4], gate4[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = nw + 16 * j + 4 * fchunk;
    bias4[j] = (EPI != EPI_F32_SCALE && g.bias && n < g.N) ? *reinterpret_cast<const uint2*>(g.bias + n) : uint2{0u, 0u};
  }
  if (EPI == EPI_GATE_RES || EPI == EPI_RES) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mw + 16 * i + frow;
      const int frame = (EPI == EPI_GATE_RES && m < g.M) ? (m / g.rows_per_frame) : 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = nw + 16 * j + 4 * fchunk;
        const bool ok = m < g.M && n < g.N;
        res4[i][j] = ok ? *reinterpret_cast<const uint2*>(g.res + (size_t)m * g.ldres + n) : uint2{0u, 0u};
        if (EPI == EPI_GATE_RES)
          gate4[i][j] = ok ? *reinterpret_cast<const uint2*>(g.gate + (size_t)frame * g.gate_frame_stride + n) : uint2{0u, 0u};
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = mw + 16 * i + frow;
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = nw + 16 * j + 4 * fchunk;
      if (n >= g.N) continue;  // N is a multiple of 4 for every caller
      if (EPI == EPI_F32_SCALE) {
        f32x4 o4;
#pragma unroll
        for (int r = 0; r < 4; ++r) o4[r] = acc[i][j][r] * g.alpha;
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + (size_t)m * g.ldc + n) = o4;
        continue;
      }
      float v[4];
      {
        const uint2 bb = bias4[j];
        const float b[4] = {bf2f(bb.x & 0xffff), bf2f(bb.x >> 16), bf2f(bb.y & 0xffff), bf2f(bb.y >> 16)};
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = rbf(acc[i][j][r] + b[r]);  // Linear output rounds to bf16
      }
      if (EPI == EPI_BIAS_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_tanh(v[r]);
      } else if (EPI == EPI_BIAS_SILU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = silu(v[r]);
      } else if (EPI == EPI_GATE_RES || EPI == EPI_RES) {
        const uint2 xx = res4[i][j];
        float x[4] = {bf2f(xx.x & 0xffff), bf2f(xx.x >> 16), bf2f(xx.y & 0xffff), bf2f(xx.y >> 16)};
        if (EPI == EPI_GATE_RES) {
          const uint2 ee = gate4[i][j];
          float e[4] = {bf2f(ee.x & 0xffff), bf2f(ee.x >> 16), bf2f(ee.y & 0xffff), bf2f(ee.y >> 16)};
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = rbf(v[r] * e[r]);  // y * e rounds, then x + (.) rounds
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = x[r] + v[r];
      }
      uint2 o;
      o.x = pack2bf(v[0], v[1]);
      o.y = pack2bf(v[2], v[3]);
      if (EPI == EPI_BIAS_VPAGES && n >= g.v_col0) {
        const int fr = m / g.rows_per_frame;
        *reinterpret_cast<uint2*>(g.v_dst[fr] + (size_t)(m - fr * g.rows_per_frame) * g.v_ld + (n - g.v_col0)) = o;
      } else {
        *reinterpret_cast<uint2*>(g.C + (size_t)m * g.ldc + n) = o;
      }
    }
  }
}

// The 128 x 128 x 64 register-staged tile at (m0, n0): body of the small-problem kernel and of the sub-tile tail launch below.
template <int EPI>
MMPL_DEV void gemm128_tile(const GemmArgs& g, char* smem, int m0, int n0) {
  char* As = smem;                   // [2][128][64] bf16, swizzled
  char* Ws = smem + 2 * TILE_BYTES;  // [2][128][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- staging: thread loads 4x16B of A and 4x16B of W per K tile
  const int srow = tid >> 3, schunk = tid & 7;
  const bf16_t* a_ptr[4];
  const bf16_t* w_ptr[4];
  int lds_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = srow + 32 * j;
    a_ptr[j] = g.A + (size_t)min(m0 + row, g.M - 1) * g.lda + schunk * 8;
    w_ptr[j] = g.W + (size_t)min(n0 + row, g.N - 1) * g.ldw + schunk * 8;
    lds_off[j] = row * 128 + ((schunk ^ (row & 7)) << 4);
  }
  u32x4 ra[4], rw[4];
  const int nt = g.K / BK;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    ra[j] = *reinterpret_cast<const u32x4*>(a_ptr[j]);
    rw[j] = *reinterpret_cast<const u32x4*>(w_ptr[j]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    *reinterpret_cast<u32x4*>(As + lds_off[j]) = ra[j];
    *reinterpret_cast<u32x4*>(Ws + lds_off[j]) = rw[j];
  }
  __syncthreads();

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (bytes within a tile): row = base + 16*i + (lane&15), 16-B chunk = 4*ks + (lane>>4)
  const int frow = lane & 15, fchunk = lane >> 4;
  int a_off[4], w_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ar = 64 * wm + 16 * i + frow, wr = 64 * wn + 16 * i + frow;
    a_off[i] = ar * 128 + ((fchunk ^ (ar & 7)) << 4);
    w_off[i] = wr * 128 + ((fchunk ^ (wr & 7)) << 4);
  }

  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) {
      const int koff = (t + 1) * BK;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ra[j] = *reinterpret_cast<const u32x4*>(a_ptr[j] + koff);
        rw[j] = *reinterpret_cast<const u32x4*>(w_ptr[j] + koff);
      }
    }
    const char* Ac = As + cur * TILE_BYTES;
    const char* Wc = Ws + cur * TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // chunk index for k-step ks is (4*ks + fchunk); XOR with (row&7) commutes with adding 4*ks (bit 2)
        af[i] = *reinterpret_cast<const bf16x8*>(Ac + (a_off[i] ^ (ks << 6)));
        wf[i] = *reinterpret_cast<const bf16x8*>(Wc + (w_off[i] ^ (ks << 6)));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < nt) {
      char* An = As + (cur ^ 1) * TILE_BYTES;
      char* Wn = Ws + (cur ^ 1) * TILE_BYTES;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        *reinterpret_cast<u32x4*>(An + lds_off[j]) = ra[j];
        *reinterpret_cast<u32x4*>(Wn + lds_off[j]) = rw[j];
      }
    }
    __syncthreads();
  }

  gemm_epilogue<EPI>(g, acc, m0 + 64 * wm, n0 + 64 * wn, frow, fchunk);
}

template <int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_bf16_kernel(GemmArgs g) {
  if (g.batch > 1) {
    g.A += (size_t)blockIdx.y * g.sA;
    g.W += (size_t)blockIdx.y * g.sW;
    g.C = (bf16_t*)((char*)g.C + (size_t)blockIdx.y * g.sC * (g.epi == EPI_F32_SCALE ? 4 : 2));
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // ---- block -> tile mapping: XCD-aware (blocks b, b+8, ... share an L2) + grouped along M
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;  // bijective for any nwg
  }
  constexpr int GROUP = 8;
  const int per_group = GROUP * tiles_n;
  const int gid = bid / per_group;
  const int first_m = gid * GROUP;
  const int gsz = min(tiles_m - first_m, GROUP);
  const int tm = first_m + (bid % per_group) % gsz;
  const int tn = (bid % per_group) / gsz;
  gemm128_tile<EPI>(g, smem, tm * BM, tn * BN);
}

// ---------------------------------------------------------------------------------------------------------------
// v2: 256x128x64 tile, 512 threads = 8 waves (4x2, 64x64 each).  Operands go global -> LDS directly
// (global_load_lds_dwordx4, no VGPR staging) into a 3-deep LDS ring, so the loads of tile t+2 are in flight while
// tile t is multiplied; the ring is ordered with a COUNTED s_waitcnt vmcnt(6) (6 LDS-DMA instructions per wave per
// tile) + a raw s_barrier -- never __syncthreads(), which would drain the DMA queue.  LDS image is lane-linear, so
// the XOR swizzle is applied to the per-lane SOURCE chunk and again on the fragment read (same involution).
constexpr int BM2 = 256, BN2 = 128;
constexpr int A2_BYTES = BM2 * BK * 2, W2_BYTES = BN2 * BK * 2, STAGE2 = A2_BYTES + W2_BYTES, NSTAGE2 = 3;

MMPL_DEV void glds16(const void* g, char* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

template <int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_bf16_v2_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_m = (g.M + BM2 - 1) / BM2, tiles_n = (g.N + BN2 - 1) / BN2;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  constexpr int GROUP = 4;
  const int per_group = GROUP * tiles_n;
  const int gid = bid / per_group;
  const int first_m = gid * GROUP;
  const int gsz = min(tiles_m - first_m, GROUP);
  const int tm = first_m + (bid % per_group) % gsz;
  const int tn = (bid % per_group) / gsz;
  const int m0 = tm * BM2, n0 = tn * BN2;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // per-lane DMA sources: LDS chunk p (16 B) of a tile holds global chunk (p&7)^(row&7) of row p>>3
  const bf16_t* a_src[4];
  const bf16_t* w_src[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = (j * 8 + wave) * 64 + lane, row = p >> 3, c = (p & 7) ^ (row & 7);
    a_src[j] = g.A + (size_t)min(m0 + row, g.M - 1) * g.lda + c * 8;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int p = (j * 8 + wave) * 64 + lane, row = p >> 3, c = (p & 7) ^ (row & 7);
    w_src[j] = g.W + (size_t)min(n0 + row, g.N - 1) * g.ldw + c * 8;
  }
  auto issue = [&](int t) {
    char* st = smem + (t % NSTAGE2) * STAGE2;
    const int koff = t * BK;
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(a_src[j] + koff, st + (j * 8 + wave) * 1024);
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16(w_src[j] + koff, st + A2_BYTES + (j * 8 + wave) * 1024);
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fchunk = lane >> 4;
  int a_off[4], w_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ar = 64 * wm + 16 * i + frow, wr = 64 * wn + 16 * i + frow;
    a_off[i] = ar * 128 + ((fchunk ^ (ar & 7)) << 4);
    w_off[i] = A2_BYTES + wr * 128 + ((fchunk ^ (wr & 7)) << 4);
  }

  const int nt = g.K / BK;
  issue(0);
  if (nt > 1) {
    issue(1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();

  for (int t = 0; t < nt; ++t) {
    if (t + 2 < nt) issue(t + 2);
    const char* st = smem + (t % NSTAGE2) * STAGE2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = *reinterpret_cast<const bf16x8*>(st + (a_off[i] ^ (ks << 6)));
        wf[i] = *reinterpret_cast<const bf16x8*>(st + (w_off[i] ^ (ks << 6)));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    // tile t+1 must have landed (everything but the 6 newest DMA ops of this wave), then everyone syncs
    if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  gemm_epilogue<EPI>(g, acc, m0 + 64 * wm, n0 + 64 * wn, frow, fchunk);
}

// (v3 -- 256x256x32 tile, 3-stage DMA ring, 990 TFLOP/s -- was removed once v4 / v6 superseded it; the tile constants stay)
constexpr int BM3 = 256, BN3 = 256;

// ---------------------------------------------------------------------------------------------------------------
// (v4 -- the lock-step predecessor of v6: same tile and 2-stage ring, both wave groups reading then multiplying together,
// 1070-1200 TFLOP/s -- was removed in round 2: unreachable in normal operation; its geometry constants stay)
constexpr int BK4 = 64, A4_BYTES = BM3 * BK4 * 2, STAGE4 = 2 * A4_BYTES;
MMPL_DEV int swz64(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }

// ---------------------------------------------------------------------------------------------------------------
// v6 "ping-pong" (default for large problems): v4's tile (256x256x64, 8 waves 2x4, 2-stage LDS-DMA ring) with the two wave groups (waves 0-3 / 4-7:
// waves w and w+4 share a SIMD) running half a k-tile apart.  Per k-tile a wave runs
//   R_t = { request its 24 LDS fragments of tile t (+ group A: issue ALL 64 LDS-DMA ops of tile t+1) }
//   M_t = { 64 MFMAs }
// and while one group is in M the other is in R, so each SIMD's matrix pipe is fed by one wave while its partner's
// LDS reads / DMA issue stalls happen in the shadow, instead of both waves reading, then both multiplying (the removed v4).
// A DMA op is global_load_lds_dwordx4 voff, s[base] with voff = (clamped row * ld + swizzled chunk) * 2: needs
// M * lda * 2 and N * ldw * 2 < 4 GiB (checked by the launcher, else the 64-bit-addressed v2).  Same MFMA order as v4 was;
// measured in one process on the 14B/720p shapes: +4 ... +14 % over v4 (1220-1290 vs 1070-1200 TFLOP/s, 1415 vs 1310
// at 8192^3).  With the DMA ops removed the same kernel runs at 1450-1470: the LDS-DMA issue stalls (~100 cycles per
// op) are the remaining cost.  A BK = 32 / 4-deep-ring variant in which both groups issue their own DMA ops inside
// their R segments was slower (twice the barriers: 1040-1150).
// cache-policy modifier of the operand streams' LDS-DMA loads (dev_knobs.h GEMM_POLICY_A / _W; "" in every shipped build)
#define GEMM_POL_STR_0 ""
#define GEMM_POL_STR_1 " nt"
#define GEMM_POL_STR_2 " sc1"
#define GEMM_POL_STR_3 " sc0 sc1"
#define GEMM_POL_CAT(n) GEMM_POL_STR_##n
#define GEMM_POL_STR(n) GEMM_POL_CAT(n)
#define GEMM_POL_A GEMM_POL_STR(GEMM_POLICY_A)
#define GEMM_POL_W GEMM_POL_STR(GEMM_POLICY_W)
MMPL_DEV void glds16s(const void* base, uint32_t voff, char* lds) {
  const uint32_t dst = (uint32_t)(size_t)((__attribute__((address_space(3))) char*)lds);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(dst) : "memory");
}
MMPL_DEV void glds16s_a(const void* base, uint32_t voff, char* lds) {
  const uint32_t dst = (uint32_t)(size_t)((__attribute__((address_space(3))) char*)lds);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" GEMM_POL_A ::"v"(voff), "s"(base), "s"(dst) : "memory");
}
MMPL_DEV void glds16s_w(const void* base, uint32_t voff, char* lds) {
  const uint32_t dst = (uint32_t)(size_t)((__attribute__((address_space(3))) char*)lds);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" GEMM_POL_W ::"v"(voff), "s"(base), "s"(dst) : "memory");
}

MMPL_DEV void store16_c(void* p, const uint4& v) {
  const u32x4 w = {v.x, v.y, v.z, v.w};
#if GEMM6_STORE == 1
  asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
#elif GEMM6_STORE == 2
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
#elif GEMM6_STORE == 3
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
#else
  *reinterpret_cast<u32x4*>(p) = w;
#endif
}
// L2 prefetch of one 64-byte half line per lane: an LDS-DMA dword into a dump area nobody reads (no destination VGPR, nothing to
// wait for in the loop; the issuing wave drains vmcnt before it leaves the kernel)
MMPL_DEV void glds4s(const void* base, uint32_t voff, char* lds) {
  const uint32_t dst = (uint32_t)(size_t)((__attribute__((address_space(3))) char*)lds);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(base), "s"(dst) : "memory");
}

// v6 epilogue, LDS-staged.  The MFMA leaves each lane with one token row and 4 consecutive output columns per fragment: stored
// directly that is 8 bytes per lane into 64 different rows per instruction.  Instead: bias, the Linear's bf16 rounding and the
// activation happen on the MFMA side, the wave's 128 x 64 sub-tile goes to its 16 KiB of the (now idle) LDS ring as bf16
// (16-byte chunk c of row r at r*128 + 16*(c ^ (r & 7))), and is read back row-major: a lane owns 8 consecutive columns of
// a row -> 16-byte residual / gate loads and stores, 128 contiguous bytes per row.  Every residual load of the sub-tile is
// issued before the first store (the residual usually IS the output buffer) and all of them are in flight at once.
struct EpiIn {                 // what a wave's 128 x 64 sub-tile epilogue reads from memory
  uint2 bias4[4];              // MFMA side: this lane's 4 output columns of each of the 4 column fragments
  uint4 res8[16];              // row-major side: the residual of row step u (8 columns)
  uint4 gate_lo, gate_hi;      // the two candidate per-frame gate rows
  int m_split;                 // first row of the sub-tile that belongs to the SECOND candidate frame (rows below it: gate_lo)
};
// The bias FIRST: loads return in order and the staging needs nothing but the bias -- issued behind the 16 residual loads (round 4)
// its `s_waitcnt` was a vmcnt(0) that held the whole staging back for an HBM round trip of the residual (seen in the ISA).
template <int EPI>
MMPL_DEV void epi_load_bias(const GemmArgs& g, EpiIn& in, int nw, int lane) {
  const int fchunk = lane >> 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int nb = nw + 16 * j + 4 * fchunk;
    in.bias4[j] = (g.bias && nb < g.N) ? *reinterpret_cast<const uint2*>(g.bias + nb) : uint2{0u, 0u};
  }
  asm volatile("" ::"v"(in.bias4[0].x), "v"(in.bias4[1].x), "v"(in.bias4[2].x), "v"(in.bias4[3].x));   // (keeps hipcc from sinking them behind the residual loads)
}
// Every residual load of the sub-tile is issued before its first store (the residual usually IS the output buffer) and all of them
// are in flight at once.  The gate is a per-FRAME vector: the sub-tile's 128 rows touch at most two frames (rows_per_frame >= 128 on this
// path, launcher), so its two candidate rows are fetched once, beside the residual, instead of one dependent L2 round trip per row step.
template <int EPI, int U0 = 0, int U1 = 16>       // row steps [U0, U1) of the residual; the gate rows come with U0 == 0
MMPL_DEV void epi_load_res(const GemmArgs& g, EpiIn& in, int mw, int nw, int lane) {
  constexpr bool HAS_RES = EPI == EPI_GATE_RES || EPI == EPI_RES;
  const int erow = lane >> 3, ec = lane & 7, n = nw + 8 * ec;      // row-major side: step u -> row 8 u + erow, columns 8 ec .. + 7
  const bool n_ok = n < g.N;                                        // N % 8 == 0 on this path (launcher)
  if (HAS_RES) {
#pragma unroll
    for (int u = U0; u < U1; ++u) {
      const int m = mw + 8 * u + erow;
      if ((GEMM6_ABL & 16) && g.M >= 0) {                              // dev mock (results are garbage): no residual loads at all -- the upper
        in.res8[u] = uint4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};      // bound of what hiding them under the k loop can buy
      } else if (n_ok && m < g.M) {
        const u32x4* rp = reinterpret_cast<const u32x4*>(g.res + (size_t)m * g.ldres + n);
        const u32x4 rv = GEMM6_RESLD ? __builtin_nontemporal_load(rp) : *rp;
        in.res8[u] = uint4{rv[0], rv[1], rv[2], rv[3]};
      } else {
        in.res8[u] = uint4{0u, 0u, 0u, 0u};
      }
    }
  }
  if (U0 != 0) return;
  in.gate_lo = uint4{0u, 0u, 0u, 0u};
  in.gate_hi = in.gate_lo;
  in.m_split = 0;
  if (EPI == EPI_GATE_RES && n_ok) {
    const int f_lo = min(mw, g.M - 1) / g.rows_per_frame;
    const int f_hi = min(mw + 127, g.M - 1) / g.rows_per_frame;
    in.m_split = (f_lo + 1) * g.rows_per_frame;                      // (one division per sub-tile instead of one per row step)
    in.gate_lo = *reinterpret_cast<const uint4*>(g.gate + (size_t)f_lo * g.gate_frame_stride + n);
    in.gate_hi = *reinterpret_cast<const uint4*>(g.gate + (size_t)f_hi * g.gate_frame_stride + n);
  }
}
// MFMA side: bias, the Linear's bf16 rounding and the activation, the sub-tile into its 16 KiB of the idle ring (16-byte chunk c of row r
// at r * 128 + 16 (c ^ (r & 7))), then the wave-level fence that lets the row-major side read it back
template <int EPI>
MMPL_DEV void epi_stage(const f32x4 (&acc)[2][4][4], const EpiIn& in, char* stg, int lane) {
  const int frow = lane & 15, fchunk = lane >> 4;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 64 * h + 16 * i + frow;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float b[4] = {__uint_as_float(in.bias4[j].x << 16), __uint_as_float(in.bias4[j].x & 0xffff0000u),
                            __uint_as_float(in.bias4[j].y << 16), __uint_as_float(in.bias4[j].y & 0xffff0000u)};
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = rbf(acc[h][i][j][r] + b[r]);                       // Linear output rounds to bf16
          if (EPI == EPI_BIAS_GELU) v[r] = gelu_tanh(v[r]);
          if (EPI == EPI_BIAS_SILU) v[r] = silu(v[r]);
        }
        const int c = 2 * j + (fchunk >> 1);                        // 16-byte chunk of the row; (fchunk & 1) picks its half
        *reinterpret_cast<uint2*>(stg + row * 128 + ((c ^ (row & 7)) << 4) + 8 * (fchunk & 1)) = uint2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
      }
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Row-major side: a lane owns 8 consecutive columns of a row -> 16-byte residual / gate operands and stores, 128 contiguous bytes per row
template <int EPI>
MMPL_DEV void epi_finish(const GemmArgs& g, const EpiIn& in, const char* stg, int mw, int nw, int lane) {
  constexpr bool HAS_RES = EPI == EPI_GATE_RES || EPI == EPI_RES;
  const int erow = lane >> 3, ec = lane & 7, n = nw + 8 * ec;
  const bool n_ok = n < g.N;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int row = 8 * u + erow, m = mw + row;
    const uint4 yv = *reinterpret_cast<const uint4*>(stg + row * 128 + ((ec ^ (row & 7)) << 4));
    if (m >= g.M || !n_ok) continue;
    uint4 ov = yv;
    if (HAS_RES) {
      uint4 ev = uint4{0u, 0u, 0u, 0u};
      if (EPI == EPI_GATE_RES) ev = (m < in.m_split) ? in.gate_lo : in.gate_hi;
      const uint32_t yw[4] = {yv.x, yv.y, yv.z, yv.w}, xw[4] = {in.res8[u].x, in.res8[u].y, in.res8[u].z, in.res8[u].w}, ew[4] = {ev.x, ev.y, ev.z, ev.w};
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float y = __uint_as_float((e & 1) ? (yw[e >> 1] & 0xffff0000u) : (yw[e >> 1] << 16));
        const float x = __uint_as_float((e & 1) ? (xw[e >> 1] & 0xffff0000u) : (xw[e >> 1] << 16));
        if (EPI == EPI_GATE_RES) y = rbf(y * __uint_as_float((e & 1) ? (ew[e >> 1] & 0xffff0000u) : (ew[e >> 1] << 16)));   // y * e rounds
        v[e] = x + y;                                               // x + (.) rounds at the pack
      }
      ov.x = pack2bf(v[0], v[1]); ov.y = pack2bf(v[2], v[3]); ov.z = pack2bf(v[4], v[5]); ov.w = pack2bf(v[6], v[7]);
    }
    if ((GEMM6_ABL & 8) && g.M >= 0) { asm volatile("" ::"v"(ov.x), "v"(ov.y), "v"(ov.z), "v"(ov.w)); continue; }
    if (EPI == EPI_BIAS_VPAGES && n >= g.v_col0) {
      const int fr = m / g.rows_per_frame;
      store16_c(g.v_dst[fr] + (size_t)(m - fr * g.rows_per_frame) * g.v_ld + (n - g.v_col0), ov);
    } else {
      store16_c(g.C + (size_t)m * g.ldc + n, ov);
    }
  }
}
template <int EPI>
MMPL_DEV void gemm_epilogue_staged(const GemmArgs& g, const f32x4 (&acc)[2][4][4], char* stg, int mw, int nw, int lane) {
  EpiIn in;
  epi_load_bias<EPI>(g, in, nw, lane);
  epi_load_res<EPI>(g, in, mw, nw, lane);
  epi_stage<EPI>(acc, in, stg, lane);
  epi_finish<EPI>(g, in, stg, mw, nw, lane);
}

// 16-byte system-scope accesses of the split-K partials: written through past the L2, read past L1 and L2 (valid between any two
// CUs of the device whatever XCDs they sit on).  Asm because hipcc has no 16-byte atomic-scope access; the loads carry their own
// wait (hipcc does not count asm loads, cdna_hip_programming.md 5.7 item 1), the store's s_nop keeps its data registers intact.
MMPL_DEV void store16_sys(f32x4* p, const f32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
MMPL_DEV void load16x8_sys(const f32x4* p, f32x4 (&v)[8]) {      // rows p, p + 64, ... p + 448 (1 KiB apart)
  asm volatile("global_load_dwordx4 %0, %8, off sc0 sc1\n\t"
               "global_load_dwordx4 %1, %8, off offset:1024 sc0 sc1\n\t"
               "global_load_dwordx4 %2, %8, off offset:2048 sc0 sc1\n\t"
               "global_load_dwordx4 %3, %8, off offset:3072 sc0 sc1\n\t"
               "global_load_dwordx4 %4, %9, off sc0 sc1\n\t"
               "global_load_dwordx4 %5, %9, off offset:1024 sc0 sc1\n\t"
               "global_load_dwordx4 %6, %9, off offset:2048 sc0 sc1\n\t"
               "global_load_dwordx4 %7, %9, off offset:3072 sc0 sc1\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
               : "v"(p), "v"(p + 256)
               : "memory");
}

// SPLITK = true: the second launch of a GEMM whose tile count leaves a partial last round of one-tile-per-CU (launch_v6).  The main
// launch stops at the last full round of every XCD's list; here each leftover tile is computed by `splitk_s` blocks over 1/s of the
// k-tiles each.  Every part leaves its fp32 accumulators in splitk_ws (lane-private layout, fully coalesced) and draws a ticket of
// the tile's counter; whoever draws the LAST one sums all parts IN PART ORDER (so the result does not depend on who was last) and
// runs the ordinary epilogue.
template <int EPI, bool SPLITK = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_bf16_v6_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  GEMM_DEV_TICKS;
  GEMM_DEV_TICK(tk0);
  const int tiles_m = (g.M + BM3 - 1) / BM3, tiles_n = (g.N + BN3 - 1) / BN3;
  const int nwg = tiles_m * tiles_n;
  // Tile order: every XCD owns a list of tiles in grouped-M order (below), so that the blocks sharing one L2 work on neighbouring
  // tiles.  Without a tile counter block b is entry b >> 3 of XCD b & 7's list (the hardware deals blocks round-robin to the
  // XCDs).  With one (g.tile_counter) the kernel is launched once per CU and a block takes the next tile of ITS XCD's list when
  // it is done with the previous one: equal tiles do not take equal time (blocks that are first to touch an
  // operand panel wait on HBM, 33-42 cycles per MFMA across blocks), and a lock-step round lasts as long as its slowest block.
  // Ticket protocol: exactly chunk + (blocks of the XCD) tickets are drawn per launch; whoever draws the last one knows every
  // other block of the XCD is leaving and zeroes the counter for the next launch.
  __shared__ int s_ticket;
  const bool persistent = !SPLITK && g.tile_counter != nullptr;
  const int my_xcd = blockIdx.x & 7;
  const int GROUP = g.group;
  const int per_group = GROUP * tiles_n;
  // Sweep-synchronous order (g.sync_sweeps): the M-groups (GROUP row panels x all column panels) are dealt to the XCDs like cards
  // -- XCD x works through groups x, x + 8, x + 16, ... -- so all eight XCDs walk the column panels of W at the same time, a W
  // panel comes out of HBM once per sweep and the other seven XCDs find it in the Infinity Cache.  With contiguous chunks of
  // the tile list (the previous order) the XCDs sat at eight different column positions and W was streamed from HBM eight
  // times over: 8.5 GB of fabric reads per qkv GEMM, 2.7 TB/s, every L2 miss an HBM miss.  What the deal leaves over (fewer than
  // 8 full groups, a ragged last group) is split into eight contiguous chunks as before.
  const int full_groups = tiles_m / GROUP;
  const int rounds = g.sync_sweeps ? full_groups >> 3 : 0;
  const int dealt = rounds * per_group;                              // tiles per XCD that come from dealt groups
  const int left = nwg - 8 * dealt;                                  // tiles of the remainder list
  const int lq = left >> 3, lr = left & 7;
  const int chunk_all = dealt + lq + (my_xcd < lr ? 1 : 0);
  const int chunk_main = g.splitk_s > 1 ? chunk_all - chunk_all % g.splitk_per : chunk_all;   // the tail belongs to the split-K launch
  const int chunk = SPLITK ? chunk_all : chunk_main;
  [[maybe_unused]] int part = 0, slot = 0;
  const int blocks_x = (int)(gridDim.x >> 3) + (my_xcd < (int)(gridDim.x & 7) ? 1 : 0);
  for (;;) {
  int idx = blockIdx.x >> 3;                                         // index into THIS XCD's tile list (block b of a one-block-per-tile
  if constexpr (SPLITK) {
    const int t_local = idx / g.splitk_s;
    part = idx - t_local * g.splitk_s;
    slot = my_xcd * g.splitk_tb + t_local;
    idx = chunk_main + t_local;
    if (idx >= chunk) return;
  }
  if (persistent) {                                                  // launch runs on XCD b & 7: the hardware deals blocks round-robin)
    if (threadIdx.x == 0) s_ticket = atomicAdd(g.tile_counter + my_xcd, 1);
    __syncthreads();
    const int ticket = s_ticket;
    if (ticket >= chunk) {
      if (ticket == chunk + blocks_x - 1 && threadIdx.x == 0) g.tile_counter[my_xcd] = 0;
      return;
    }
    idx = ticket;
  }
  int bid;                                                           // position in the grouped-M tile list
  if (idx < dealt) {
    bid = ((idx / per_group) * 8 + my_xcd) * per_group + idx % per_group;
  } else {
    bid = 8 * dealt + (my_xcd < lr ? my_xcd * (lq + 1) : lr * (lq + 1) + (my_xcd - lr) * lq) + (idx - dealt);
  }
  const int gid = bid / per_group;
  const int first_m = gid * GROUP;
  const int gsz = min(tiles_m - first_m, GROUP);
  const int tm = first_m + (bid % per_group) % gsz;
  const int tn = (bid % per_group) / gsz;
  const int m0 = tm * BM3, n0 = tn * BN3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3, grp = wave >> 2;

  // ---- DMA (group A only): wave w issues pieces 4q + w, q = 0..15 (q < 8: A rows 8*(4q+w).., q >= 8: W rows)
  const int prow = lane >> 3, lc = ((lane & 7) ^ (prow & 7)) << 3;   // row within the piece, swizzled source chunk (elements)
  const int nt_all = g.K / BK4;
  const int kt0 = SPLITK ? part * nt_all / g.splitk_s : 0;          // this block's k-tiles: [kt0, kt0 + nt)
  const int nt = SPLITK ? (part + 1) * nt_all / g.splitk_s - kt0 : nt_all;
  const bf16_t* const A_k0 = g.A + (size_t)kt0 * BK4;
  const bf16_t* const W_k0 = g.W + (size_t)kt0 * BK4;
  const bf16_t* a_k = A_k0;                                        // advance by BK4 elements per tile
  const bf16_t* w_k = W_k0;
  auto issue_piece = [&](int q, char* st) {
    const bool isw = q >= 8;
    const int p = 4 * (q & 7) + wave;                              // piece of the A (or W) tile: rows 8p .. 8p+7
    int row = isw ? min(n0 + 8 * p + prow, g.N - 1) : min(m0 + 8 * p + prow, g.M - 1);
    if constexpr (GEMM6_ABL & 1) row = 8 * p + prow;
    const uint32_t voff = (uint32_t)(row * (isw ? g.ldw : g.lda) + lc) * 2u;
    if (isw) glds16s_w(w_k, voff, st + A4_BYTES + p * 1024);
    else glds16s_a(a_k, voff, st + p * 1024);
  };

  f32x4 acc[2][4][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fchunk = lane >> 4;
  int a_off[8], w_off[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) a_off[i] = swz64(128 * wm + 16 * i + frow, fchunk);
#pragma unroll
  for (int i = 0; i < 4; ++i) w_off[i] = A4_BYTES + swz64(64 * wn + 16 * i + frow, fchunk);
  // L2 prefetch shares (see the loop): wave-uniform scalars
  const bool pf_on = g.pf_dist > 0 && grp == 1 && gsz >= 2;
  const int pf_P = 32 / gsz;
  const int pf_nw = __builtin_amdgcn_readfirstlane(2 * (256 / gsz)), pf_na = __builtin_amdgcn_readfirstlane(2 * (256 / pf_P));
  const int pf_wrow0 = __builtin_amdgcn_readfirstlane(n0 + ((bid % per_group) % gsz) * (256 / gsz));
  const int pf_arow0 = __builtin_amdgcn_readfirstlane(m0 + (tn % pf_P) * (256 / pf_P));

  if (grp == 0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) issue_piece(q, smem);
    a_k += BK4;
    w_k += BK4;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();          // group B runs half a tile late
  GEMM_DEV_TICK(tk1);

  for (int t = 0; t < nt; ++t) {
    // =========================== R_t
    // L2 prefetch (g.pf_dist k-tiles ahead, group B: its vmcnt has no waiter in the loop; issued at the top of R_t, where the
    // fragment registers are dead).  The tiles in flight on an XCD share their operand slices -- the gsz tiles of one column panel
    // its W slice, the ~32 / gsz tiles of one row panel its A slice -- and run in near lock step, so every k-tile somebody takes
    // the fabric / HBM miss and the sharers queue behind it (with every fetch an L2 hit the same loop runs 10-20 % faster,
    // GEMM6_ABL=1).  Each tile therefore touches ITS share of both slices early: rows tm_l * 256 / gsz ... of W and
    // (tn % P) * 256 / P ... of A (P = 32 / gsz), two half lines (64 B) a row, one per lane.
    if (pf_on && t + g.pf_dist < nt) {
      const int l = (wave - 4) * 64 + lane;
      char* dump = smem + 2 * STAGE4 + (wave - 4) * 256;
      const size_t koff = (size_t)(t + g.pf_dist) * BK4;
      if (l < pf_nw) glds4s(W_k0 + koff, (uint32_t)(min(pf_wrow0 + (l >> 1), g.N - 1) * g.ldw + (l & 1) * 32) * 2u, dump);
      if (l < pf_na) glds4s(A_k0 + koff, (uint32_t)(min(pf_arow0 + (l >> 1), g.M - 1) * g.lda + (l & 1) * 32) * 2u, dump);
    }
    const char* st = smem + (t & 1) * STAGE4;
    char* nx = smem + ((t + 1) & 1) * STAGE4;
    const bool do_issue = grp == 0 && t + 1 < nt && !(GEMM6_ABL & 4);
    bf16x8 af[2][8], wf[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        wf[ks][i] = *reinterpret_cast<const bf16x8*>(st + (w_off[i] ^ (ks << 6)));
        if (do_issue) issue_piece(ks * 8 + i, nx);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        af[ks][i] = *reinterpret_cast<const bf16x8*>(st + (a_off[i] ^ (ks << 6)));
        if ((i & 1) && do_issue) issue_piece(ks * 8 + 4 + (i >> 1), nx);
      }
    }
    if (do_issue) { a_k += BK4; w_k += BK4; }
    // the fragments are complete HERE (hipcc would otherwise sink the reads next to their MFMAs, into the M segment)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      asm volatile("" : "+v"(af[ks][0]), "+v"(af[ks][1]), "+v"(af[ks][2]), "+v"(af[ks][3]), "+v"(af[ks][4]), "+v"(af[ks][5]),
                        "+v"(af[ks][6]), "+v"(af[ks][7]), "+v"(wf[ks][0]), "+v"(wf[ks][1]), "+v"(wf[ks][2]), "+v"(wf[ks][3]));
    }
    __builtin_amdgcn_s_barrier();
    // =========================== M_t
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i >> 2][i & 3][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][j], af[ks][i], acc[i >> 2][i & 3][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile t+1 landed before anyone reads it
    // group A's barrier here pairs with group B's after R_t: it also tells A that B's last fragment reads of the ring are done.
    // B's own barrier after its LAST multiply would only pair with an extra one in A, and nothing needs it (no tile follows,
    // the epilogues stage through disjoint halves of the ring): without the pair group A runs its epilogue -- residual loads,
    // LDS staging, stores -- while group B is still multiplying, and the tile's HBM burst comes in two halves.
    if (grp == 0 || t + 1 < nt) __builtin_amdgcn_s_barrier();
  }
  GEMM_DEV_TICK(tk2);
  if constexpr (SPLITK) {
    // partial of this k range: register r of lane l of wave w at ((part slot * 8 + w) * 32 + r) * 64 + l (16 bytes each)
    const size_t part_floats = (size_t)8 * 32 * 64 * 4;
    f32x4* wsp = reinterpret_cast<f32x4*>(g.splitk_ws + ((size_t)slot * g.splitk_s + part) * part_floats) + (size_t)wave * 32 * 64 + lane;
#pragma unroll
    for (int r = 0; r < 32; ++r) store16_sys(wsp + r * 64, acc[r >> 4][(r >> 2) & 3][r & 3]);
    // PLACEMENT-INDEPENDENT exchange (round 4; MI355X_MICROARCH.md "inter-workgroup visibility", form {sc0 sc1 stores and loads on
    // both sides}): the partials are written through to memory past the writer's L2 (which drops the lines) and read back past
    // the reader's L1 / L2, the ticket is an agent-scope atomic behind the drained stores.  Nothing here depends on which XCD a
    // part runs on any more -- round 3 relied on "workgroup b runs on XCD b & 7" (probed once, on an idle device) for VISIBILITY;
    // a CU mask, another queue or a second process sharing the GPU would have made the last part sum stale L2 lines.  The tile
    // order still puts the parts of a tile on one XCD, for speed only.  A device-scope release fence instead of write-through
    // stores writes back the whole L2: +60 us per GEMM (profiles/r03S_gemm_splitk_device_fence_ab_rejected.log).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the partial stores (and group B's prefetch DMAs into this block's LDS)
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(g.splitk_cnt + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != g.splitk_s - 1) return;                         // block-uniform
    if (threadIdx.x == 0) g.splitk_cnt[slot] = 0;                    // every other part of the tile has left: zero for the next launch
    const f32x4* rsp = reinterpret_cast<const f32x4*>(g.splitk_ws + (size_t)slot * g.splitk_s * part_floats) + (size_t)wave * 32 * 64 + lane;
    // summed IN PART ORDER whoever is last (part 0 first, this block's own partial re-read like the others: same bits either way)
    for (int p = 0; p < g.splitk_s; ++p) {
#pragma unroll
      for (int r0 = 0; r0 < 32; r0 += 8) {
        f32x4 v[8];
        load16x8_sys(rsp + r0 * 64, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int r = r0 + i;
          f32x4& a = acc[r >> 4][(r >> 2) & 3][r & 3];
          if (p == 0) a = v[i];
          else { a[0] += v[i][0]; a[1] += v[i][1]; a[2] += v[i][2]; a[3] += v[i][3]; }
        }
      }
      rsp += part_floats / 4;
    }
  }
  if constexpr (GEMM6_ABL & 2) {
    if (g.M < 0) gemm_epilogue<EPI>(g, acc[0], m0 + 128 * wm, n0 + 64 * wn, frow, fchunk);      // keeps the accumulators alive
  } else if (EPI != EPI_F32_SCALE && g.staged_epilogue) {
    gemm_epilogue_staged<EPI>(g, acc, smem + wave * (128 * 128), m0 + 128 * wm, n0 + 64 * wn, lane);
  } else {
    gemm_epilogue<EPI>(g, acc[0], m0 + 128 * wm, n0 + 64 * wn, frow, fchunk);
    gemm_epilogue<EPI>(g, acc[1], m0 + 128 * wm + 64, n0 + 64 * wn, frow, fchunk);
  }
#ifdef MMPL_DEV_ABLATIONS
  if constexpr (GEMM6_TIMING) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tk3 = __builtin_readcyclecounter();
    __syncthreads();
    if (lane == 0 && (wave & 1) == 0) {                      // 4 of the 8 waves: the layout tools/bench_kernels.py gemmphases reads
      float* tp = reinterpret_cast<float*>(g.C) + (blockIdx.x * 4 + (wave >> 1)) * 4;
      tp[0] = (float)(tk1 - tk0);
      tp[1] = (float)(tk2 - tk1);
      tp[2] = (float)(tk3 - tk2);
      tp[3] = (float)(2 * nt);                                // in 32-wide k stages
    }
  }
#endif
  if (g.pf_dist > 0 && grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // prefetch DMAs target this block's LDS
  if (!persistent) return;
  __syncthreads();                 // the ring (and s_ticket) are free again: every wave is done with its epilogue's staging
  }
}

// The same 128 x 128 x 64 tile for the sub-tile tail launch, fed by a 4-stage LDS-DMA ring (128 KiB: a tail block has its CU to itself).
// The register-staged body above keeps ONE k-tile in flight; run by the handful of tail blocks on an otherwise idle chip it is pure
// latency (24 k-steps of ~1.2 us: a round of 128 x 128 quadrants took 0.8 of a full round of 256 x 256 tiles, profiles/r04f_*).  Here
// three k-tiles are in flight behind the one being multiplied.  Same MFMA shape, operand roles, k order and epilogue: same bits.
template <int EPI>
MMPL_DEV void gemm128_ring_tile(const GemmArgs& g, char* smem, int m0, int n0) {
  constexpr int NST = 4, STG = 2 * TILE_BYTES;                  // stage = A rows [0, 16 KiB) + W rows [16 KiB, 32 KiB)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // piece p of an operand tile = rows 8p .. 8p + 7 (1 KiB, lane-linear in LDS: row 8p + (lane >> 3), chunk position lane & 7, which holds
  // logical chunk (lane & 7) ^ (row & 7)); wave w issues pieces 4q + w, q = 0..3, of A and of W
  const int prow = lane >> 3, lc = ((lane & 7) ^ (prow & 7)) << 3;
  uint32_t a_vo[4], w_vo[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int p = 4 * q + wave;
    a_vo[q] = (uint32_t)(min(m0 + 8 * p + prow, g.M - 1) * g.lda + lc) * 2u;
    w_vo[q] = (uint32_t)(min(n0 + 8 * p + prow, g.N - 1) * g.ldw + lc) * 2u;
  }
  const int nt = g.K / BK;
  auto issue = [&](int t) {                                      // all 8 pieces of k-tile t into stage t & 3
    char* st = smem + (t & (NST - 1)) * STG;
    const bf16_t* ak = g.A + (size_t)t * BK;
    const bf16_t* wk = g.W + (size_t)t * BK;
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16s(ak, a_vo[q], st + (4 * q + wave) * 1024);
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16s(wk, w_vo[q], st + TILE_BYTES + (4 * q + wave) * 1024);
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fchunk = lane >> 4;
  int a_off[4], w_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ar = 64 * wm + 16 * i + frow, wr = 64 * wn + 16 * i + frow;
    a_off[i] = ar * 128 + ((fchunk ^ (ar & 7)) << 4);
    w_off[i] = TILE_BYTES + wr * 128 + ((fchunk ^ (wr & 7)) << 4);
  }
  const int pre = nt < NST - 1 ? nt : NST - 1;
  for (int t = 0; t < pre; ++t) issue(t);
  // in flight per wave, oldest first: 8 pieces per k-tile.  k-tile t has landed when at most 8 x (tiles issued after it) are outstanding
  if (pre >= 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if (pre == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll 1
  for (int t = 0; t < nt; ++t) {
    if (t + NST - 1 < nt) issue(t + NST - 1);                    // into the stage k-tile t - 1 left (everybody is past it: barrier below)
    const char* st = smem + (t & (NST - 1)) * STG;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = *reinterpret_cast<const bf16x8*>(st + (a_off[i] ^ (ks << 6)));
        wf[i] = *reinterpret_cast<const bf16x8*>(st + (w_off[i] ^ (ks << 6)));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    // k-tile t + 1 must have landed: tiles t + 2 and t + 3 (those that exist) may still be on their way
    const int later = (t + 3 < nt ? 1 : 0) + (t + 2 < nt ? 1 : 0);
    if (later == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  gemm_epilogue<EPI>(g, acc, m0 + 64 * wm, n0 + 64 * wn, frow, fchunk);
}

// Sub-tile launch for the partial last round of SHORT-K GEMMs (K < 4096, where the split-K launch loses to its fp32 partial exchange:
// Wan 1.3B's qkv / o / cross-q / cross-o at 480p have 774 = 3 x 256 + 6 and 258 = 256 + 2 tiles, i.e. a whole extra round of one
// tile per CU for 2-6 tiles).  The main launch stops at the full rounds exactly as for split-K (GemmArgs.splitk_s > 1); here every
// leftover 256 x 256 tile of every XCD's list is computed as four 128 x 128 quadrants (gemm128_ring_tile), one block per CU, all of
// them in one short round.  Same MFMA shape, operand roles and k order per accumulator as v6, same epilogue arithmetic:
// bit-identical to the one-launch result (tests/test_kernels_gpu.py::test_gemm_subtile_tail).
// RING: the 4-stage DMA-ring body (128 KiB of LDS, one block per CU: leftovers that fit one round that way, <= 8 tiles per XCD);
// else the register-staged body (64 KiB, two blocks per CU: up to 16 tiles per XCD, e.g. the 78-tile GEMMs that have no full round).
template <int EPI, bool RING>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_tail128_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_m = (g.M + BM3 - 1) / BM3, tiles_n = (g.N + BN3 - 1) / BN3;
  const int nwg = tiles_m * tiles_n;
  const int my_xcd = blockIdx.x & 7, r = blockIdx.x >> 3, t_local = r >> 2, quad = r & 3;
  // the XCD's tile list, as in gemm_bf16_v6_kernel
  const int GROUP = g.group, per_group = GROUP * tiles_n;
  const int rounds = g.sync_sweeps ? (tiles_m / GROUP) >> 3 : 0;
  const int dealt = rounds * per_group, left = nwg - 8 * dealt, lq = left >> 3, lr = left & 7;
  const int chunk_all = dealt + lq + (my_xcd < lr ? 1 : 0);
  const int idx = chunk_all - chunk_all % g.splitk_per + t_local;
  if (idx >= chunk_all) return;
  int bid;
  if (idx < dealt) bid = ((idx / per_group) * 8 + my_xcd) * per_group + idx % per_group;
  else bid = 8 * dealt + (my_xcd < lr ? my_xcd * (lq + 1) : lr * (lq + 1) + (my_xcd - lr) * lq) + (idx - dealt);
  const int first_m = (bid / per_group) * GROUP;
  const int gsz = min(tiles_m - first_m, GROUP);
  const int tm = first_m + (bid % per_group) % gsz, tn = (bid % per_group) / gsz;
  const int m0 = tm * BM3 + 128 * (quad >> 1), n0 = tn * BN3 + 128 * (quad & 1);
  if (m0 >= g.M || n0 >= g.N) return;
  if constexpr (RING) gemm128_ring_tile<EPI>(g, smem, m0, n0);
  else gemm128_tile<EPI>(g, smem, m0, n0);
}

// ---------------------------------------------------------------------------------------------------------------
// v8 "w128": the same 256x256x64 block tile, 2-stage LDS-DMA ring, LDS image, tile order / tickets and epilogues as v6, but FOUR waves,
// one per SIMD, each a 128 x 128 sub-tile = 8 x 8 fragments of v_mfma_f32_16x16x32_bf16 whose 256 fp32 accumulators are the
// accumulator file (a[0:255], named literally; acc(i, j) = a[4 (8 i + j) ..+3]).  Per k-tile a wave issues 128 MFMAs and reads 32
// fragments (16 activation + 16 weight) -- 0.25 ds_read_b128 per MFMA where v6's 128 x 64 wave tile needs 0.375: on this power-limited
// chip LDS bytes are energy, and the wide block GEMMs (qkv, ffn0: half of a forward's GEMM FLOPs) are where the vendor library's
// kernel of this geometry was 9-14 % ahead of v6 (profiles/r03b_gemm6_diag.log).  Round 3's first cut of this geometry kept a 4-deep
// ring of 32-wide stages and lost to v6; this one holds the WHOLE current k-tile in registers (2 x 16 fragments = 128 VGPRs):
//   phase 1 of tile t:  64 MFMAs on the first 32-wide half (fragments already in registers) | the 16 fragments of the second half are
//                       read from stage t & 1; once every wave has them (lgkmcnt(0) + barrier) that stage is dead, and the LDS-DMA of
//                       tile t+2 goes straight into it
//   phase 2 of tile t:  64 MFMAs on the second half | the rest of tile t+2's 16 DMA pieces; vmcnt(16) + barrier = tile t+1 has landed
//                       for everybody; its first-half fragments are read from the other stage
// so two tiles are in flight with two stages, every DMA piece has > 128 MFMA times (~1 us) to land, and nothing but MFMAs, 32 reads,
// 16 DMA pieces and 2 barriers is issued per k-tile.  Same MFMA operand order and per-accumulator k order as v6: bit-identical results.
struct G8 {
  bf16x8 af[2][8], wf[2][8];      // [32-wide half of the k-tile][fragment]: activations (B operand), weights (A operand)
  uint32_t ra[2][2], rw[2][2];    // fragment-0 LDS read address [stage][half]; fragment i at + 2048 i
  uint32_t a_vo[8], w_vo[8];      // per-piece LDS-DMA source byte offsets (constant over k)
  const bf16_t* a_k; const bf16_t* w_k;   // operand base of the next k-tile to fetch
  const bf16_t* a_last; const bf16_t* w_last;   // ... of the last k-tile
  uint32_t dst0;                  // LDS address of this wave's piece 0 of stage 0's activation half (piece q at + 4096 q)
  uint32_t pf_wo, pf_ao, pf_dump; // L2 prefetch (GEMM8_PF): this lane's byte offsets into the tile's shares of W and A, dump address

  template <int KS, int I, int J, bool ZERO = false> MMPL_DEV void mfma() {      // ZERO: C = 0 (a tile's very first MFMA per accumulator)
    constexpr int c = 4 * (8 * I + J);
    if constexpr (ZERO) asm volatile("v_mfma_f32_16x16x32_bf16 a[%c0:%c1], %2, %3, 0" ::"i"(c), "i"(c + 3), "v"(wf[KS][J]), "v"(af[KS][I]));
    else asm volatile("v_mfma_f32_16x16x32_bf16 a[%c0:%c1], %2, %3, a[%c0:%c1]" ::"i"(c), "i"(c + 3), "v"(wf[KS][J]), "v"(af[KS][I]));
  }
  template <int S, int KS, int R> MMPL_DEV void lds() {      // fragment R of half KS of the tile in stage S: weights first, then activations
    if constexpr (GEMM8_ABL & 4) return;
    if constexpr (R < 8) asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(wf[KS][R]) : "v"(rw[S][KS]), "i"(2048 * R));
    else asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(af[KS][R - 8]) : "v"(ra[S][KS]), "i"(2048 * (R - 8)));
  }
  template <int S, int Q> MMPL_DEV void dma() {              // piece Q of the next tile into stage S: Q < 8 activation rows, else weight rows
    if constexpr (GEMM8_ABL & 1) return;
    constexpr int q = Q & 7;
    const uint32_t m = dst0 + S * STAGE4 + (Q < 8 ? 0 : A4_BYTES) + 4096 * q;
    if constexpr (Q < 8) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" GEMM_POL_A ::"v"(a_vo[q]), "s"(a_k), "s"(m) : "memory");
    else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" GEMM_POL_W ::"v"(w_vo[q]), "s"(w_k), "s"(m) : "memory");
  }
  // v6's L2 prefetch of the tile's SHARE of its operand slices (see gemm_bf16_v6_kernel), one k-tile ahead of the ring's DMA: two
  // LDS-DMA dwords per wave into a dump area nobody reads.  ALWAYS two ops per wave and k-tile (lanes outside the share repeat
  // lane 0's address), so that the loop's counted vmcnt waits stay compile-time constants.
  MMPL_DEV void prefetch() {
    if constexpr (GEMM8_ABL & 1) return;
    const bf16_t* pw = w_k + BK4 <= w_last ? w_k + BK4 : w_last;      // (past the last k-tile: that tile again, never past the row)
    const bf16_t* pa = a_k + BK4 <= a_last ? a_k + BK4 : a_last;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(pf_wo), "s"(pw), "s"(pf_dump) : "memory");
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(pf_ao), "s"(pa), "s"(pf_dump + 256u) : "memory");
  }
};

// gap placement of one k-tile: GEMM8_BAR1 / _DMAS / _BAR2 / _RD / _FINEWAIT / _PF (dev_knobs.h)
// S: ring stage of this tile; ISSUE: tile t+2 exists (fetch it into stage S); NEXT: tile t+1 exists (read its first half)
template <int S, bool ISSUE, bool NEXT, bool FIRST = false> MMPL_DEV void gemm8_tile(G8& k) {
  constexpr int D1 = GEMM8_BAR1 + 2;                                   // first DMA gap of phase 1
  constexpr int N1 = (64 - D1 + GEMM8_DMAS - 1) / GEMM8_DMAS;          // pieces issued in phase 1
  static_assert(N1 >= 0 && N1 <= 16 && (16 - N1) * GEMM8_DMAS + 1 < GEMM8_BAR2 && GEMM8_BAR2 + GEMM8_RD * 16 < 64 && GEMM8_RD * 15 < GEMM8_BAR1, "placement");
  // this tile's first-half fragments, read at the end of the previous tile in the order w0..w7, a0..a7 (LDS returns in order): MFMA row i
  // needs everything up to a_i, i.e. at most 7 - i of those reads may still be outstanding -- plus the second-half reads this phase has
  // issued by then (the counter is 4 bits: 15 = "all of the old ones", since the newest 16 are then the new reads)
  if constexpr (GEMM8_FINEWAIT) asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
  else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  sfor<64>([&k](auto gi) {
    constexpr int g = decltype(gi)::value;
    if constexpr (GEMM8_FINEWAIT && g % 8 == 0 && g > 0) {
      constexpr int i = g / 8, n_new = (g + GEMM8_RD - 1) / GEMM8_RD < 16 ? (g + GEMM8_RD - 1) / GEMM8_RD : 16;
      constexpr int n = 7 - i + n_new < 15 ? 7 - i + n_new : 15;
      asm volatile("s_waitcnt lgkmcnt(%c0)" ::"i"(n) : "memory");
    }
    k.template mfma<0, g / 8, g % 8, FIRST>();
    if constexpr (g % GEMM8_RD == 0 && g / GEMM8_RD < 16) k.template lds<S, 1, g / GEMM8_RD>();
    if constexpr (g == GEMM8_BAR1 && !(GEMM8_ABL & 8)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (ISSUE && g >= D1 && (g - D1) % GEMM8_DMAS == 0) k.template dma<S, (g - D1) / GEMM8_DMAS>();
  });
  sfor<64>([&k](auto gi) {
    constexpr int g = decltype(gi)::value;
    k.template mfma<1, g / 8, g % 8>();
    if constexpr (ISSUE && g % GEMM8_DMAS == 0 && N1 + g / GEMM8_DMAS < 16) k.template dma<S, N1 + g / GEMM8_DMAS>();
    if constexpr (ISSUE && GEMM8_PF && g == (16 - N1) * GEMM8_DMAS) k.prefetch();
    if constexpr (NEXT && g == GEMM8_BAR2 && !(GEMM8_ABL & 8)) {
      // in flight, oldest first: tile t+1's 16 pieces [+ 2 prefetch ops], tile t+2's 16 [+ 2]: everything but the first 16 may be
      if constexpr (ISSUE) asm volatile("s_waitcnt vmcnt(%c0)\n\ts_barrier" ::"i"(GEMM8_PF ? 20 : 16) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if constexpr (NEXT && g > GEMM8_BAR2 && (g - GEMM8_BAR2) % GEMM8_RD == 0 && (g - GEMM8_BAR2) / GEMM8_RD <= 16)
      k.template lds<S ^ 1, 0, (g - GEMM8_BAR2) / GEMM8_RD - 1>();
  });
  if constexpr (ISSUE) { k.a_k += BK4; k.w_k += BK4; }
}

template <int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_bf16_v8_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_ticket;
  const int tiles_m = (g.M + BM3 - 1) / BM3, tiles_n = (g.N + BN3 - 1) / BN3;
  const int nwg = tiles_m * tiles_n;
  // tile order, tickets and the split between the main launch and the split-K tail launch: exactly v6's (see gemm_bf16_v6_kernel)
  const bool persistent = g.tile_counter != nullptr;
  const int my_xcd = blockIdx.x & 7;
  const int GROUP = g.group;
  const int per_group = GROUP * tiles_n;
  const int full_groups = tiles_m / GROUP;
  const int rounds = g.sync_sweeps ? full_groups >> 3 : 0;
  const int dealt = rounds * per_group;
  const int left = nwg - 8 * dealt;
  const int lq = left >> 3, lr = left & 7;
  const int chunk_all = dealt + lq + (my_xcd < lr ? 1 : 0);
  const int chunk = g.splitk_s > 1 ? chunk_all - chunk_all % g.splitk_per : chunk_all;
  const int blocks_x = (int)(gridDim.x >> 3) + (my_xcd < (int)(gridDim.x & 7) ? 1 : 0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 15, fchunk = lane >> 4;
  GEMM_DEV_TICKS;
  GEMM_DEV_TICK(tk0);
  asm volatile("s_nop 0" ::: MMPL_ALL_AGPRS);

  G8 k;
  {
    const uint32_t ring = (uint32_t)(size_t)((__attribute__((address_space(3))) char*)smem);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        k.ra[s][ks] = ring + s * STAGE4 + (swz64(128 * wm + frow, fchunk) ^ (ks << 6));
        k.rw[s][ks] = ring + s * STAGE4 + A4_BYTES + (swz64(128 * wn + frow, fchunk) ^ (ks << 6));
      }
    k.dst0 = ring + 1024 * wave;
    k.pf_dump = ring + 2 * STAGE4 + 512 * wave;
  }
  const int prow = lane >> 3, lc = ((lane & 7) ^ (prow & 7)) << 3;     // row within a piece, swizzled source chunk (elements)
  const int nt = g.K / BK4;

  for (;;) {
    int idx = blockIdx.x >> 3;
    if (persistent) {
      if (threadIdx.x == 0) s_ticket = atomicAdd(g.tile_counter + my_xcd, 1);
      __syncthreads();
      const int ticket = s_ticket;
      if (ticket >= chunk) {
        if (ticket == chunk + blocks_x - 1 && threadIdx.x == 0) g.tile_counter[my_xcd] = 0;
        return;
      }
      idx = ticket;
    } else if (idx >= chunk) {
      return;
    }
    int bid;
    if (idx < dealt) bid = ((idx / per_group) * 8 + my_xcd) * per_group + idx % per_group;
    else bid = 8 * dealt + (my_xcd < lr ? my_xcd * (lq + 1) : lr * (lq + 1) + (my_xcd - lr) * lq) + (idx - dealt);
    const int gid = bid / per_group;
    const int first_m = gid * GROUP;
    const int gsz = min(tiles_m - first_m, GROUP);
    const int tm = first_m + (bid % per_group) % gsz;
    const int tn = (bid % per_group) / gsz;
    const int m0 = tm * BM3, n0 = tn * BN3;

    // piece p = 4 q + wave of an operand's 32: rows 8 p .. 8 p + 7 (clamped at the matrix edge; the epilogue never stores those rows)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int p = 4 * q + wave;
      k.a_vo[q] = (uint32_t)(min(m0 + 8 * p + prow, g.M - 1) * g.lda + lc) * 2u;
      k.w_vo[q] = (uint32_t)(min(n0 + 8 * p + prow, g.N - 1) * g.ldw + lc) * 2u;
    }
    k.a_k = g.A; k.w_k = g.W;
    k.a_last = g.A + (size_t)(nt - 1) * BK4; k.w_last = g.W + (size_t)(nt - 1) * BK4;
    if constexpr (GEMM8_PF) {
      // the share of this tile: rows tm_l * 256 / gsz ... of its W panel (the gsz tiles of a column panel split it), rows
      // (tn % P) * 256 / P ... of its A panel (P = 32 / gsz tiles of a row panel in flight on the XCD); one 64-byte half line per
      // lane, 256 lanes of the block = wave * 64 + lane; lanes past a share repeat its first line.  g.pf_dist == 0: own first rows.
      const int P = 32 / gsz, nw = 2 * (256 / gsz), na = 2 * (256 / P);
      const int l = wave * 64 + lane;
      const int wrow0 = n0 + ((bid % per_group) % gsz) * (256 / gsz), arow0 = m0 + (tn % P) * (256 / P);
      const int lw = (g.pf_dist > 0 && l < nw) ? l : 0, la = (g.pf_dist > 0 && l < na) ? l : 0;
      k.pf_wo = (uint32_t)(min(wrow0 + (lw >> 1), g.N - 1) * g.ldw + (lw & 1) * 32) * 2u;
      k.pf_ao = (uint32_t)(min(arow0 + (la >> 1), g.M - 1) * g.lda + (la & 1) * 32) * 2u;
    }
    // (accumulators: zeroed by the first k-tile's MFMAs themselves, C = 0, whenever the peeled first pair below exists)
    if (nt < 4) sfor<256>([](auto ii) { asm volatile("v_accvgpr_write_b32 a[%c0], 0" ::"i"(decltype(ii)::value)); });

    // ---- prologue: tiles 0 and 1 into the two stages, tile 0's first-half fragments into registers
    sfor<16>([&k](auto q) { k.template dma<0, decltype(q)::value>(); });
    k.a_k += BK4; k.w_k += BK4;
    if (nt > 1) {
      sfor<16>([&k](auto q) { k.template dma<1, decltype(q)::value>(); });
      if constexpr (GEMM8_PF) k.prefetch();                          // (same queue pattern as a loop iteration: 16 pieces, 2 prefetch ops)
      k.a_k += BK4; k.w_k += BK4;
      asm volatile("s_waitcnt vmcnt(%c0)\n\ts_barrier" ::"i"(GEMM8_PF ? 18 : 16) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    sfor<16>([&k](auto r) { k.template lds<0, 0, decltype(r)::value>(); });

    GEMM_DEV_TICK(tk1);
    int t = 0;
    if (nt >= 4) {
      gemm8_tile<0, true, true, true>(k);
      gemm8_tile<1, true, true>(k);
      t = 2;
    }
#pragma unroll 1
    for (; t + 3 < nt; t += 2) {
      gemm8_tile<0, true, true>(k);
      gemm8_tile<1, true, true>(k);
    }
    if (nt - t == 3) {
      gemm8_tile<0, true, true>(k);
      gemm8_tile<1, false, true>(k);
      gemm8_tile<0, false, false>(k);
    } else if (nt - t == 2) {
      gemm8_tile<0, false, true>(k);
      gemm8_tile<1, false, false>(k);
    } else {
      gemm8_tile<0, false, false>(k);
    }
    // every wave is past the last tile's phase-1 barrier: nobody reads the ring any more, the epilogue stages through it.
    // (the MFMA results need their passes before a VALU instruction may read the accumulator file)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    GEMM_DEV_TICK(tk2);

    // ---- epilogue: v6's, on the two 128 x 64 halves of the wave's sub-tile.  Its lane-dependent addresses are derived from an
    // OPAQUE copy of the lane id defined here: otherwise hipcc hoists them above the k loop and, short of VGPRs there, parks them
    // in accumulator registers -- which this kernel owns by name (tests/test_isa_audit.py fails on any compiler access to a[..]).
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int frow_e = lane_e & 15, fchunk_e = lane_e >> 4;
    const int mw = m0 + 128 * wm, nw0 = n0 + 128 * wn;
    auto read_half = [](auto nhi, f32x4 (&acc)[2][4][4]) {
      constexpr int nh = decltype(nhi)::value;
      sfor<128>([&acc](auto ii) {
        constexpr int x = decltype(ii)::value, i8 = x >> 4, j = (x >> 2) & 3, r = x & 3;
        float v;
        asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(v) : "i"(4 * (8 * i8 + 4 * nh + j) + r));
        acc[i8 >> 2][i8 & 3][j][r] = v;
      });
    };
    if (EPI != EPI_F32_SCALE && g.staged_epilogue) {
      // Round 5: the second half's bias / residual / gate loads are issued BEFORE the first half's read-back and stores (disjoint
      // columns, so nothing it reads is about to be written), i.e. they fly under ~a third of the epilogue instead of stalling the
      // second half's read-back for an HBM round trip.
      char* stg0 = smem + wave * 32768;
      EpiIn in0, in1;
      epi_load_bias<EPI>(g, in0, nw0, lane_e);
      epi_load_res<EPI>(g, in0, mw, nw0, lane_e);
      {
        f32x4 acc[2][4][4];
        read_half(w64::ic<0>{}, acc);
        epi_stage<EPI>(acc, in0, stg0, lane_e);
      }
      epi_load_bias<EPI>(g, in1, nw0 + 64, lane_e);
      // (the first NPRE row steps: the register file has no room for all 16 beside the second half's accumulators -- with more, hipcc
      // spills or, worse, parks values in the accumulator file that still holds them: tests/test_isa_audit.py.  The gated epilogue only
      // fits since its per-row `m / rows_per_frame` became one compare against EpiIn::m_split.)
      constexpr int NPRE = 8;
      epi_load_res<EPI, 0, NPRE>(g, in1, mw, nw0 + 64, lane_e);
      epi_finish<EPI>(g, in0, stg0, mw, nw0, lane_e);
      {
        f32x4 acc[2][4][4];
        read_half(w64::ic<1>{}, acc);
        epi_stage<EPI>(acc, in1, stg0 + 16384, lane_e);
      }
      epi_load_res<EPI, NPRE, 16>(g, in1, mw, nw0 + 64, lane_e);
      epi_finish<EPI>(g, in1, stg0 + 16384, mw, nw0 + 64, lane_e);
    } else {
      sfor<2>([&](auto nhi) {
        constexpr int nh = decltype(nhi)::value;
        f32x4 acc[2][4][4];
        read_half(nhi, acc);
        gemm_epilogue<EPI>(g, acc[0], mw, nw0 + 64 * nh, frow_e, fchunk_e);
        gemm_epilogue<EPI>(g, acc[1], mw + 64, nw0 + 64 * nh, frow_e, fchunk_e);
      });
    }
#ifdef MMPL_DEV_ABLATIONS
    if constexpr (GEMM6_TIMING) {      // per-wave { prologue, k loop, epilogue } shader cycles over the output (tools/bench_kernels.py gemmphases)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      tk3 = __builtin_readcyclecounter();
      __syncthreads();
      if (lane == 0) {
        float* tp = reinterpret_cast<float*>(g.C) + (blockIdx.x * 4 + wave) * 4;
        tp[0] = (float)(tk1 - tk0);
        tp[1] = (float)(tk2 - tk1);
        tp[2] = (float)(tk3 - tk2);
        tp[3] = (float)(2 * nt);
      }
    }
#endif
    if (!persistent) return;
    __syncthreads();                 // the ring (and s_ticket) are free again
  }
}

template <int EPI>
hipError_t launch_v6(const GemmArgs& g, hipStream_t s) {
  constexpr int smem = 2 * STAGE4 + 1024;        // ring + the L2 prefetch's dump area
  if (hipError_t e = mmpl_dyn_smem_once(reinterpret_cast<const void*>(gemm_bf16_v6_kernel<EPI>), smem); e != hipSuccess) return e;
  if (hipError_t e = mmpl_dyn_smem_once(reinterpret_cast<const void*>(gemm_bf16_v6_kernel<EPI, true>), smem); e != hipSuccess) return e;
  if (hipError_t e = mmpl_dyn_smem_once(reinterpret_cast<const void*>(gemm_bf16_v8_kernel<EPI>), 2 * STAGE4 + 2048); e != hipSuccess) return e;
  if (hipError_t e = mmpl_dyn_smem_once(reinterpret_cast<const void*>(gemm_tail128_kernel<EPI, true>), 8 * TILE_BYTES); e != hipSuccess) return e;
  if (hipError_t e = mmpl_dyn_smem_once(reinterpret_cast<const void*>(gemm_tail128_kernel<EPI, false>), 4 * TILE_BYTES); e != hipSuccess) return e;
  // main-launch kernel: v8 (one wave per SIMD, 128 x 128 per wave) or v6 (two, 128 x 64).  MMPL_GEMM_V8 = 1 / 0 forces it.
  const int env_v8 = mmpl_config().gemm_v8;
  const bool use_v8 = env_v8 >= 0 ? env_v8 != 0 : g.N >= 8192;       // the wide GEMMs (qkv, ffn0): profiles/r04c_gemm_v8_*.log
  const int tiles = ((g.M + BM3 - 1) / BM3) * ((g.N + BN3 - 1) / BN3);
  GemmArgs g2 = g;
  const MmplRuntimeConfig& rc = mmpl_config();
  const int env_group = rc.gemm_group;
  // M-tile group of the block order (how many row panels the tiles in flight on an XCD span).  Round 3 swept it on the kernels of the
  // time (profiles/r03d_*, r03C_*) and gave the narrow (N < 8192) GEMMs groups of 3 (2 for the long-K one) where the row panels divide
  // by 3; re-swept in round 6 on today's kernels (staged epilogue in phases, split-K tail; profiles/r06w_gemm_group_sweep_v6_shapes.log,
  // TFLOP/s at group 2 / 3 / 4 / 6 / 8, one box):
  //   M = 25200 (99 row panels): o 1289 / 1301 / 1333 / 1310 / 1362, cross-o 1317 / 1316 / 1355 / 1338 / 1373, ffn2 1363 / 1353 / 1401 / 1360 / 1392
  //   M = 21600 (85 row panels): o 1293 / 1307 / 1331 / 1310 / 1326, ffn2 1344 / 1345 / 1396 / 1350 / 1388
  // -> 4 (also what the wide GEMMs on v8 keep: profiles/r05g_gemm_group_sweep.log), except the narrow short-K GEMMs (o, cross-q,
  // cross-o) of the 7-frame stage (>= 96 row panels): 8.
  const int tiles_m_ = (g.M + BM3 - 1) / BM3;
  g2.group = env_group > 0 ? env_group : ((tiles_m_ >= 96 && g.N < 8192 && g.K < 8192) ? 8 : 4);
  // the 16-byte epilogue needs 8-element alignment of everything it touches; otherwise the direct 8-byte one
  // (strides AND base pointers: mmpl_gemm is public ABI and callers hand in views such as a column-offset C)
  auto al = [](const void* p, uintptr_t a) { return p == nullptr || reinterpret_cast<uintptr_t>(p) % a == 0; };
  bool ptrs_ok = al(g.C, 16) && al(g.bias, 8);
  if (g.epi == EPI_GATE_RES || g.epi == EPI_RES) ptrs_ok = ptrs_ok && al(g.res, 16);
  if (g.epi == EPI_GATE_RES) ptrs_ok = ptrs_ok && al(g.gate, 16);
  if (g.epi == EPI_BIAS_VPAGES)
    for (int i = 0; i < 8; ++i) ptrs_ok = ptrs_ok && al(g.v_dst[i], 16);
  g2.staged_epilogue = ptrs_ok && g.N % 8 == 0 && g.ldc % 8 == 0 &&
                       (g.epi != EPI_GATE_RES || (g.gate_frame_stride % 8 == 0 && g.rows_per_frame >= 128)) &&
                       ((g.epi != EPI_GATE_RES && g.epi != EPI_RES) || g.ldres % 8 == 0) &&
                       (g.epi != EPI_BIAS_VPAGES || (g.v_col0 % 8 == 0 && g.v_ld % 8 == 0));
  // L2 prefetch distance (k-tiles): 2 measured best on the 14B / 720p block shapes (+1 % qkv / o / ffn0, +6 % ffn2 whose A operand is
  // 700 MB; 1 = no gain, 4 and more lose again); MMPL_GEMM_PF=0 switches it off
  g2.pf_dist = rc.gemm_pf;
  g2.sync_sweeps = 1;
  const int per = mmpl_cus_per_xcd(), n_cu = 8 * per;
  // Split-K launch for the partial last round.  With one tile per CU a GEMM of R * 256 + t tiles takes R + 1 rounds however small t
  // is (Wan 1.3B at 480p: 43 x 6 = 258 tiles for o / ffn2 at s1, 78 at s0; 14B / 720p s0: 580).  When every XCD's leftover (its
  // list length mod 32, same arithmetic as the kernel) fits one round in s >= 2 parts, the main launch stops at the full rounds and
  // the leftover tiles run as s blocks each over 1/s of K.  s <= 4: the last part to arrive reads all s partials (256 KiB each)
  // alone.  Only for K >= 4096: the few blocks of the tail launch stream their operands alone (no neighbours sharing the L2 lines)
  // at about half the usual rate and the partials' round trip is ~15 us, so short-K GEMMs lose (K = 1536: -9...-23 %; K = 5120:
  // +1...+10 %; K = 8960 / 13824: +14...+58 %, profiles/r03S_gemm_splitk_micro.log).
  g2.splitk_s = 1; g2.splitk_tb = 0; g2.splitk_per = per;
  if (g2.tile_counter && g.splitk_ws && g.splitk_cnt && !rc.gemm_no_splitk && EPI != EPI_F32_SCALE && g.K / BK4 >= 64) {
    const int tiles_n_ = (g.N + BN3 - 1) / BN3, per_group = g2.group * tiles_n_;
    const int dealt = g2.sync_sweeps ? ((tiles_m_ / g2.group) >> 3) * per_group : 0, left = tiles - 8 * dealt;
    int tb = 0, main_tiles = 0;
    for (int x = 0; x < 8; ++x) {
      const int chunk = dealt + (left >> 3) + (x < (left & 7) ? 1 : 0);
      tb = chunk % per > tb ? chunk % per : tb;
      main_tiles += chunk - chunk % per;
    }
    int sp = tb > 0 ? per / tb : 1;
    sp = sp > 4 ? 4 : sp;
    if (sp >= 2 && (size_t)8 * tb * sp * (BM3 * BN3 * sizeof(float)) <= mmpl_gemm_splitk_ws_bytes()) {
      g2.splitk_s = sp; g2.splitk_tb = tb;
      if (main_tiles > 0) {
        if (use_v8) hipLaunchKernelGGL(gemm_bf16_v8_kernel<EPI>, dim3(main_tiles < n_cu ? main_tiles : n_cu), dim3(256), 2 * STAGE4 + 2048, s, g2);
        else hipLaunchKernelGGL(gemm_bf16_v6_kernel<EPI>, dim3(main_tiles < n_cu ? main_tiles : n_cu), dim3(512), smem, s, g2);
      }
      hipLaunchKernelGGL((gemm_bf16_v6_kernel<EPI, true>), dim3(8 * tb * sp), dim3(512), smem, s, g2);
      return hipGetLastError();
    }
  }
  // Sub-tile launch for the partial last round of short-K GEMMs (gemm_tail128_kernel): when every XCD's leftover is at most half a
  // round of tiles, the main launch stops at the full rounds and the leftovers run as 128 x 128 quadrants (a 4-stage DMA-ring body, one
  // short round) instead of a whole round of one 256 x 256 tile per CU.  Needs the tile tickets (like the split-K launch), no scratch;
  // MMPL_GEMM_NO_SUBTILE (or MMPL_GEMM_NO_SPLITK) switches it off.
  if (g2.tile_counter && !rc.gemm_no_splitk && !rc.gemm_no_subtile && EPI != EPI_F32_SCALE && g.K / BK4 < 64) {
    const int tiles_n_ = (g.N + BN3 - 1) / BN3, per_group = g2.group * tiles_n_;
    const int dealt = g2.sync_sweeps ? ((tiles_m_ / g2.group) >> 3) * per_group : 0, left = tiles - 8 * dealt;
    int tb = 0, main_tiles = 0;
    for (int x = 0; x < 8; ++x) {
      const int chunk = dealt + (left >> 3) + (x < (left & 7) ? 1 : 0);
      tb = chunk % per > tb ? chunk % per : tb;
      main_tiles += chunk - chunk % per;
    }
    if (tb > 0 && 2 * tb <= per) {
      g2.splitk_s = 4; g2.splitk_tb = tb;                       // (the main kernel only looks at splitk_s > 1: stop at the full rounds)
      // (tickets, like the split-K launch: the XCDs' full-round counts can differ by a whole round, which only the ticket loop absorbs)
      if (main_tiles > 0) {
        if (use_v8) hipLaunchKernelGGL(gemm_bf16_v8_kernel<EPI>, dim3(main_tiles < n_cu ? main_tiles : n_cu), dim3(256), 2 * STAGE4 + 2048, s, g2);
        else hipLaunchKernelGGL(gemm_bf16_v6_kernel<EPI>, dim3(main_tiles < n_cu ? main_tiles : n_cu), dim3(512), smem, s, g2);
      }
      if (4 * tb <= per) hipLaunchKernelGGL((gemm_tail128_kernel<EPI, true>), dim3(8 * tb * 4), dim3(256), 8 * TILE_BYTES, s, g2);
      else hipLaunchKernelGGL((gemm_tail128_kernel<EPI, false>), dim3(8 * tb * 4), dim3(256), 4 * TILE_BYTES, s, g2);
      return hipGetLastError();
    }
  }
  const int blocks = g2.tile_counter && tiles > n_cu ? n_cu : tiles;
  if (blocks == tiles) g2.tile_counter = nullptr;             // one round or less: nothing to balance
  if (use_v8) hipLaunchKernelGGL(gemm_bf16_v8_kernel<EPI>, dim3(blocks), dim3(256), 2 * STAGE4 + 2048, s, g2);
  else hipLaunchKernelGGL(gemm_bf16_v6_kernel<EPI>, dim3(blocks), dim3(512), smem, s, g2);
  return hipGetLastError();
}

template <int EPI>
hipError_t launch_v2(const GemmArgs& g, hipStream_t s) {
  constexpr int smem = NSTAGE2 * STAGE2;
  if (hipError_t e = mmpl_dyn_smem_once(reinterpret_cast<const void*>(gemm_bf16_v2_kernel<EPI>), smem); e != hipSuccess) return e;
  const int tiles = ((g.M + BM2 - 1) / BM2) * ((g.N + BN2 - 1) / BN2);
  hipLaunchKernelGGL(gemm_bf16_v2_kernel<EPI>, dim3(tiles), dim3(512), smem, s, g);
  return hipGetLastError();
}

template <int EPI>
hipError_t launch(const GemmArgs& g, hipStream_t s) {
  // kernel-selection overrides for A/B runs (mmpl_config.h).  (gemm_w64.hip -- one wave per SIMD, 32x32x16 MFMAs, 85 % matrix-pipe
  // issue rate inside its k loop yet 3-6 % slower end to end than v6 on this power-limited chip -- was removed in round 3;
  // DESIGN.md section 3.2 keeps what it taught.)
  const bool env_v1 = mmpl_config().gemm_v1, env_v2 = mmpl_config().gemm_v2;
  const bool big = g.batch <= 1 && g.M >= 1024 && g.N >= 256 && g.K >= 128 && !env_v1 && !env_v2;
  // v6 addresses its operands with 32-bit byte offsets from the base pointers; anything larger goes to v2 (64-bit pointers)
  if (big && (long long)g.M * g.lda < (1ll << 31) && (long long)g.N * g.ldw < (1ll << 31)) return launch_v6<EPI>(g, s);
  if (g.batch <= 1 && g.M >= 1024 && g.N >= 128 && g.K >= 128 && !env_v1) return launch_v2<EPI>(g, s);
  constexpr int smem = 4 * TILE_BYTES;
  if (hipError_t e = mmpl_dyn_smem_once(reinterpret_cast<const void*>(gemm_bf16_kernel<EPI>), smem); e != hipSuccess) return e;
  const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  hipLaunchKernelGGL(gemm_bf16_kernel<EPI>, dim3(tiles, g.batch > 1 ? g.batch : 1), dim3(256), smem, s, g);
  return hipGetLastError();
}

}  // namespace

size_t mmpl_gemm_splitk_ws_bytes() { return (size_t)256 * BM3 * BN3 * sizeof(float); }   // <= one part per CU

hipError_t mmpl_launch_gemm(const GemmArgs& g, hipStream_t s) {
  if (g.M <= 0 || g.N <= 0) return hipSuccess;
  if (g.K % BK != 0 || g.N % 4 != 0 || g.lda % 8 != 0 || g.ldw % 8 != 0 || g.ldc % 4 != 0) return hipErrorInvalidValue;
  switch (g.epi) {
    case EPI_BIAS: return launch<EPI_BIAS>(g, s);
    case EPI_BIAS_GELU: return launch<EPI_BIAS_GELU>(g, s);
    case EPI_BIAS_SILU: return launch<EPI_BIAS_SILU>(g, s);
    case EPI_GATE_RES: return launch<EPI_GATE_RES>(g, s);
    case EPI_RES: return launch<EPI_RES>(g, s);
    case EPI_F32_SCALE: return launch<EPI_F32_SCALE>(g, s);
    case EPI_BIAS_VPAGES: return launch<EPI_BIAS_VPAGES>(g, s);
  }
  return hipErrorInvalidValue;
}
