// bf16 GEMM for the large DiT linears, ONE wave per SIMD:  C[M,N] = epi( A[M,K] . W[N,K]^T + bias[N] )   (same contract and
// epilogues as gemm.hip; replaces the nn.Linear calls of CausalWanAttentionBlock, causal_fps_model.py:322-365).
//
// Block = 256 x 256 output tile, 4 waves (2 x 2), each wave a 128 x 128 sub-tile = 4 x 4 tiles of v_mfma_f32_32x32x16_bf16 whose
// 256 fp32 accumulators ARE the accumulator file (a[0:255], named literally; the compiler never sees them).  K advances in
// stages of 32: a stage is 32 KiB of LDS (A rows [0,16K), W rows [16K,32K), 64 B per row), 4 stages form a ring.  Per stage a
// wave issues 32 MFMAs and, in their shadow, everything else (MI355X_MICROARCH.md, one wave per SIMD: ~5 issue slots per MFMA):
//     16 ds_read_b128   the NEXT stage's fragments into the other register buffer (each fragment feeds 4 MFMAs)
//      8 LDS-DMA pieces stage t+4 of the ring, one piece every 3 gaps -- never in consecutive gaps: the four waves run in step,
//                       and pieces in a row pile up at the CU's single address path (measured in attn_w64: 145 cycles per 8)
//      1 barrier        s_waitcnt vmcnt(16) + s_barrier in gap 8: stage t+1 has landed for every wave (two whole stages of
//                       latency cover) and every wave is done reading stage t's slot, which this stage's DMA overwrites
// The predecessor (gemm_bf16_v6_kernel: two waves per SIMD, 2-stage ring drained with vmcnt(0) every k-tile, all 64 DMA ops of
// a tile issued back to back by one wave group) ran the matrix pipe 55-61 % busy (profiles/r02e_pmc_mfma_busy.md).
//
// Operand addressing: buffer descriptors over the tile's rows (base = first row, num_records = bytes up to the end of the
// matrix' last row), so rows past M / N read as zeros by the hardware range check -- no clamping -- and the k offset is the
// scalar offset.  LDS image: row r, logical 16-byte chunk c at r*64 + 16*(c ^ ((r >> 2) & 3)) (conflict-free b128 reads);
// the LDS-DMA writes lane-linear, so the swizzle is applied to the per-lane SOURCE chunk.
//
// MFMA D = Wfrag . Afrag^T: lane (l31, hi) holds token row m = l31 and output columns n = 8 (r >> 2) + (r & 3) + 4 hi of each
// 32 x 32 tile.  The epilogue rounds (bias, activation) on that side and transposes each 32-row slab through 8 KiB of LDS per
// wave so that a lane owns 8 consecutive columns of one row: 16-byte stores, coalesced over 256 B per row.  The residual tile
// (in-place x += ... for two of a block's four big linears) is fetched by the k loop's last four stages, whose DMA slots
// would otherwise idle: with every CU of the chip finishing its tile at the same moment, loading it in the epilogue was a
// 128 KiB-per-CU HBM burst that cost 24k cycles per tile (measured), 12 % of the whole GEMM.
// Rounding points as in gemm.hip (Linear output -> bf16, y*e -> bf16, x + (.) -> bf16).
#include <stdlib.h>

#include "common.h"
#include "kernels.h"
#include "w64_util.h"

namespace {
using w64::sfor;

constexpr int GM = 256, GN = 256, GK = 32, G_RING = 4;
constexpr int G_HALF = GM * GK * 2;            // 16 KiB: one operand's rows of a stage
constexpr int G_STAGE = 2 * G_HALF;            // 32 KiB
constexpr int G_RING_BYTES = G_RING * G_STAGE; // 128 KiB
constexpr int G_STG = 32 * 128 * 2;            // epilogue staging per wave: a 32-row slab of its 128 bf16 columns, 8 KiB
constexpr int G_SMEM = G_RING_BYTES + 4 * G_STG;   // 160 KiB: all of the CU's LDS
// Timing ablations (dev, results are garbage): -DGEMM_ABL=<bits>  1 no LDS-DMA in the loop, 4 no fragment reads, 8 no barrier,
// 16 leave per-wave loop cycle counts in C (tools/bench_kernels.py gemmcycles).  0 in every shipped build.
#ifndef GEMM_ABL
#define GEMM_ABL 0
#endif

struct GCtx {
  bf16x8 af[2][2][4], wf[2][2][4];     // [register buffer][k step of 16][32-row tile]: activations (MFMA B operand), weights (A)
  uint32_t a_voff[4], w_voff[4];       // per-piece LDS-DMA source offsets (constant)
  uint32_t a_rd[2], w_rd[2];           // per-k-step fragment read offsets within a stage (constant)
  uint32_t ra[2], rw[2];               // the same, plus the ring slot being read
  u32x4 asrd, wsrd;
  uint32_t ksoff;                      // k byte offset of the DMA cursor
  uint32_t dslot, rslot;               // ring slot (byte offset) the DMA cursor writes / the next fragment reads come from
  uint32_t wave_off;                   // 4096 * wave: this wave's pieces within an operand's half of a stage
  u32x4 rsrd; uint32_t r_voff, res_soff, res_step;   // residual tile: descriptor, per-lane offset, slab offset, bytes per 4 rows

  template <int P, int G> MMPL_DEV void mfma() {       // G = 16 ks + 4 i + j: acc(i, j) += W frag j . A frag i
    constexpr int ks = G >> 4, i = (G >> 2) & 3, j = G & 3, acc = 16 * (4 * i + j);
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c0:%c1], %2, %3, a[%c0:%c1]" ::"i"(acc), "i"(acc + 15), "v"(wf[P][ks][j]), "v"(af[P][ks][i]));
  }
  template <int P, int R> MMPL_DEV void lds() {        // read R of the next stage: the order the next stage's MFMAs want them in
    constexpr int ks = R >> 3, w = (R >> 2) & 1, i = R & 3;
    if constexpr (GEMM_ABL & 4) return;
    if constexpr (w == 0) asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(wf[P][ks][i]) : "v"(rw[ks]), "i"(2048 * i));
    else asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(af[P][ks][i]) : "v"(ra[ks]), "i"(2048 * i));
  }
  MMPL_DEV void read_addr() {                          // fragment addresses of the slot under the read cursor; cursor to the next slot
    asm volatile("v_add_u32 %0, %4, %5\n\tv_add_u32 %1, %4, %6\n\tv_add_u32 %2, %4, %7\n\tv_add_u32 %3, %4, %8"
                 : "=&v"(ra[0]), "=&v"(ra[1]), "=&v"(rw[0]), "=&v"(rw[1]) : "s"(rslot), "v"(a_rd[0]), "v"(a_rd[1]), "v"(w_rd[0]), "v"(w_rd[1]));
    rslot = (rslot + G_STAGE) & (G_RING_BYTES - 1);
  }
  // LDS-DMA piece Q of the stage under the DMA cursor: Q < 4 activations rows, Q >= 4 weight rows.  M0 (the LDS address of the
  // wave's piece 0 of that operand) is written with piece 0 / 4; the instruction offset moves BOTH addresses by 1 KiB per
  // piece (the source offsets are pre-compensated).  "s_mov m0 + s_nop 3" is also the 5 wait states between a scalar write of
  // the offset register (the cursor advance, which hipcc may place right before this asm) and the VMEM instruction reading it.
  template <int Q> MMPL_DEV void dma() {
    constexpr int q = Q & 3;
    if constexpr (GEMM_ABL & 1) return;
    if constexpr (Q < 4) {
      if constexpr (q == 0) {
        const uint32_t m = dslot + wave_off;
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 3\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(a_voff[0]), "s"(asrd), "s"(ksoff), "s"(m) : "memory");
      } else {
        asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%c3 lds" ::"v"(a_voff[q]), "s"(asrd), "s"(ksoff), "i"(1024 * q) : "memory");
      }
    } else {
      if constexpr (q == 0) {
        const uint32_t m = dslot + wave_off + G_HALF;
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 3\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(w_voff[0]), "s"(wsrd), "s"(ksoff), "s"(m) : "memory");
      } else {
        asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%c3 lds" ::"v"(w_voff[q]), "s"(wsrd), "s"(ksoff), "i"(1024 * q) : "memory");
      }
    }
  }
  // The last four stages have nothing left to fetch for the ring, so their 8 DMA slots per wave carry the RESIDUAL instead:
  // stage nt-4+i brings slab i (32 rows) of the wave's 128 x 128 sub-tile into the ring slot that stage frees, piece u = rows
  // 4u..4u+3 x 128 columns in the lane order the epilogue's row-major side reads them back (wave-private: no barrier needed).
  template <int Q> MMPL_DEV void dma_res() {
    constexpr int q = Q & 3;
    const uint32_t so = res_soff + Q * res_step - 1024u * q;
    if constexpr (q == 0) {
      const uint32_t m = dslot + 2 * wave_off + 4096 * (Q >> 2);
      asm volatile("s_mov_b32 m0, %3\n\ts_nop 3\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(r_voff), "s"(rsrd), "s"(so), "s"(m) : "memory");
    } else {
      asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2 offen offset:%c3 lds" ::"v"(r_voff), "s"(rsrd), "s"(so), "i"(1024 * q) : "memory");
    }
  }
  MMPL_DEV void dma_advance() {
    ksoff += GK * 2;
    dslot = (dslot + G_STAGE) & (G_RING_BYTES - 1);
  }
  // stage t+1 (DMA'd three stages ago) has landed when at most the 16 pieces of the last two stages are outstanding
  MMPL_DEV void barrier() { if constexpr (GEMM_ABL & 8) return; asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory"); }
};

// One k stage: MFMAs on register buffer P, the next stage's fragments into buffer P ^ 1.  Gap placement (dev sweep: -DGEMM_BAR=..):
#ifndef GEMM_BAR
#define GEMM_BAR 8        // gap of the barrier
#endif
#ifndef GEMM_LDS0
#define GEMM_LDS0 (GEMM_BAR + 2)   // first fragment read (16 reads, one per gap)
#endif
#ifndef GEMM_DMA0
#define GEMM_DMA0 (GEMM_BAR + 2)   // first DMA piece
#endif
#ifndef GEMM_DMAS
#define GEMM_DMAS 3       // gaps between DMA pieces
#endif
template <int P, bool RES> MMPL_DEV void gemm_stage(GCtx& k) {
  static_assert(GEMM_LDS0 > GEMM_BAR + 1 && GEMM_LDS0 + 16 <= 32 && GEMM_DMA0 > GEMM_BAR && GEMM_DMA0 + 7 * GEMM_DMAS < 32, "placement");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // buffer P (read during the previous stage)
  sfor<32>([&k](auto gi) {
    constexpr int g = decltype(gi)::value;
    k.template mfma<P, g>();
    if constexpr (g == GEMM_BAR) k.barrier();
    if constexpr (g == GEMM_BAR + 1) k.read_addr();
    if constexpr (g >= GEMM_LDS0 && g < GEMM_LDS0 + 16) k.template lds<P ^ 1, g - GEMM_LDS0>();
    if constexpr (g >= GEMM_DMA0 && (g - GEMM_DMA0) % GEMM_DMAS == 0 && (g - GEMM_DMA0) / GEMM_DMAS < 8) {
      if constexpr (RES) k.template dma_res<(g - GEMM_DMA0) / GEMM_DMAS>();
      else k.template dma<(g - GEMM_DMA0) / GEMM_DMAS>();
    }
  });
  k.dma_advance();
  if constexpr (RES) k.res_soff += 8 * k.res_step;
}

MMPL_DEV u32x4 rows_srd(const bf16_t* first_row, long long rows_left, int ld, int K) {
  // [first_row, end of the last row's K elements): anything past it reads as zero
  const uint64_t p = (uint64_t)first_row;
  const long long bytes = rows_left <= 0 ? 0 : ((rows_left - 1) * (long long)ld + K) * 2;
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((uint32_t)p);
  r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane((uint32_t)(bytes > 0xffffffffll ? 0xffffffffll : bytes));
  r[3] = 0x00020000u;
  return r;
}

template <int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_w64_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_m = (g.M + GM - 1) / GM, tiles_n = (g.N + GN - 1) / GN;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {  // XCD-aware bijective remap, then grouped-M order (as gemm_bf16_v6_kernel)
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int per_group = g.group * tiles_n;
  const int first_m = (bid / per_group) * g.group;
  const int gsz = min(tiles_m - first_m, g.group);
  const int tm = first_m + (bid % per_group) % gsz;
  const int tn = (bid % per_group) / gsz;
  const int m0 = tm * GM, n0 = tn * GN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int hi = lane >> 5, l31 = lane & 31;

  [[maybe_unused]] unsigned long long tick_start = 0, tick_loop = 0, tick_epi = 0;
  if constexpr (GEMM_ABL & 32) tick_start = __builtin_readcyclecounter();
  asm volatile("s_nop 0" ::: MMPL_ALL_AGPRS);
  sfor<256>([](auto ii) { asm volatile("v_accvgpr_write_b32 a[%c0], 0" ::"i"(decltype(ii)::value)); });

  GCtx k;
  k.asrd = rows_srd(g.A + (size_t)m0 * g.lda, (long long)g.M - m0, g.lda, g.K);
  k.wsrd = rows_srd(g.W + (size_t)n0 * g.ldw, (long long)g.N - n0, g.ldw, g.K);
  k.wave_off = wave * 4096;
  {
    const int rp = lane >> 2, cp = lane & 3;                     // row within the piece, chunk position within the LDS row
    const int chunk = cp ^ ((rp >> 2) & 3);                      // the logical chunk that lives there
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 16 * (4 * wave + q) + rp;
      k.a_voff[q] = (uint32_t)(row * g.lda + 8 * chunk) * 2u - 1024u * q;
      k.w_voff[q] = (uint32_t)(row * g.ldw + 8 * chunk) * 2u - 1024u * q;
    }
    const int s = (l31 >> 2) & 3;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      k.a_rd[ks] = (128 * wm + l31) * 64 + 16 * ((2 * ks + hi) ^ s);
      k.w_rd[ks] = G_HALF + (128 * wn + l31) * 64 + 16 * ((2 * ks + hi) ^ s);
    }
  }
  k.ksoff = 0; k.dslot = 0; k.rslot = 0;

  // ---- prologue: stages 0..3 into the ring, stage 0's fragments into buffer 0
#pragma unroll 1
  for (int s = 0; s < G_RING; ++s) {
    sfor<8>([&k](auto q) { k.template dma<decltype(q)::value>(); });
    k.dma_advance();
  }
  asm volatile("s_waitcnt vmcnt(24)\n\ts_barrier" ::: "memory");
  k.read_addr();
  sfor<16>([&k](auto r) { k.template lds<0, decltype(r)::value>(); });

  const int nt = g.K / GK;                                       // even: K % 64 == 0 (checked by the launcher)
  [[maybe_unused]] unsigned long long tick0 = 0;
  if constexpr (GEMM_ABL & (16 | 32)) tick0 = __builtin_readcyclecounter();
  constexpr bool HAS_RES = EPI == EPI_GATE_RES || EPI == EPI_RES;
  const int mw = m0 + 128 * wm, nw = n0 + 128 * wn;             // the wave's sub-tile
  if constexpr (HAS_RES) {
    // [first element of the sub-tile, end of the matrix' last row): rows past M read as zeros
    const long long bytes = (long long)g.M - mw <= 0 ? 0 : (((long long)g.M - mw - 1) * g.ldres + (g.N - nw)) * 2;
    const uint64_t p = (uint64_t)(g.res + (size_t)mw * g.ldres + nw);
    k.rsrd[0] = __builtin_amdgcn_readfirstlane((uint32_t)p);
    k.rsrd[1] = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32) & 0xffffu);
    k.rsrd[2] = __builtin_amdgcn_readfirstlane((uint32_t)(bytes > 0xffffffffll ? 0xffffffffll : (bytes < 0 ? 0 : bytes)));
    k.rsrd[3] = 0x00020000u;
    k.r_voff = (uint32_t)((lane >> 4) * g.ldres + 8 * (lane & 15)) * 2u;
    k.res_step = (uint32_t)g.ldres * 8u;                         // 4 rows, in bytes
    k.res_soff = 0;
  }
#pragma unroll 1
  for (int t = 0; t < nt - 4; t += 2) {
    gemm_stage<0, false>(k);
    gemm_stage<1, false>(k);
  }
  gemm_stage<0, HAS_RES>(k);
  gemm_stage<1, HAS_RES>(k);
  gemm_stage<0, HAS_RES>(k);
  gemm_stage<1, HAS_RES>(k);
  if constexpr (GEMM_ABL & 16) {
    const unsigned long long dt = __builtin_readcyclecounter() - tick0;
    if (lane == 0) {
      float* tp = reinterpret_cast<float*>(g.C) + (blockIdx.x * 4 + wave) * 2;
      tp[0] = (float)dt;
      tp[1] = (float)nt;
    }
    return;
  }
  if constexpr (GEMM_ABL & 32) tick_loop = __builtin_readcyclecounter();
  // every wave is past its last fragment read; the residual slabs (this wave's own pieces) have landed
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");

  // ---- epilogue, one 32-row slab (MFMA tile row i) at a time, everything wave-private.
  // MFMA side: lane (l31, hi) holds row l31, columns 32 j + 8 gq + 4 hi + {0..3} in a[64 i + 16 j + 4 gq ..]: bias, rounding to
  // bf16 and the activation happen here, the 4 values go to the wave's staging slab as one 8-byte write (16-byte chunk c of
  // row r at r*256 + 16*(c ^ (r & 15)): conflict-free both ways).  Row-major side: step u, lane -> row 4 u + (lane >> 4),
  // columns 8 (lane & 15) .. + 7: one 16-byte read of the staged values, one of the residual (DMA'd by the last four stages
  // in exactly this lane order), gate, and a 16-byte store, 256 contiguous bytes per row.
  char* stg = smem + G_RING_BYTES + wave * G_STG;
  const int erow = lane >> 4, ec = lane & 15, n = nw + 8 * ec;
  const bool n_ok = n < g.N;                                       // N % 8 == 0 (launcher)
  uint2 bias_m[16];
#pragma unroll
  for (int x = 0; x < 16; ++x) {
    const int nb = nw + 32 * (x >> 2) + 8 * (x & 3) + 4 * hi;
    bias_m[x] = (g.bias && nb < g.N) ? *reinterpret_cast<const uint2*>(g.bias + nb) : uint2{0u, 0u};
  }
  sfor<4>([&](auto it) {
    constexpr int i = decltype(it)::value;
    float o[64];
    sfor<64>([&o](auto ii) { constexpr int x = decltype(ii)::value; asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(o[x]) : "i"(64 * i + x)); });
#pragma unroll
    for (int x = 0; x < 16; ++x) {                                 // x = 4 j + gq = the 16-byte chunk of the row
      const float b[4] = {__uint_as_float(bias_m[x].x << 16), __uint_as_float(bias_m[x].x & 0xffff0000u),
                          __uint_as_float(bias_m[x].y << 16), __uint_as_float(bias_m[x].y & 0xffff0000u)};
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = rbf(o[4 * x + e] + b[e]);                           // Linear output rounds to bf16
        if (EPI == EPI_BIAS_GELU) v[e] = gelu_tanh(v[e]);
        if (EPI == EPI_BIAS_SILU) v[e] = silu(v[e]);
      }
      *reinterpret_cast<uint2*>(stg + l31 * 256 + ((x ^ (l31 & 15)) << 4) + 8 * hi) = uint2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const char* rslab = smem + ((nt + i) & 3) * G_STAGE + 8192 * wave;       // stage nt-4+i went to ring slot (nt-4+i) mod 4
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int row = 4 * u + erow, m = mw + 32 * i + row;
      const uint4 yv = *reinterpret_cast<const uint4*>(stg + row * 256 + ((ec ^ (row & 15)) << 4));
      uint4 xv = uint4{0u, 0u, 0u, 0u}, ev = uint4{0u, 0u, 0u, 0u};
      if (HAS_RES) xv = *reinterpret_cast<const uint4*>(rslab + 1024 * u + 16 * lane);
      if (m >= g.M || !n_ok) continue;
      if (EPI == EPI_GATE_RES) ev = *reinterpret_cast<const uint4*>(g.gate + (size_t)(m / g.rows_per_frame) * g.gate_frame_stride + n);
      uint4 ov = yv;
      if (HAS_RES) {
        const uint32_t yw[4] = {yv.x, yv.y, yv.z, yv.w}, xw[4] = {xv.x, xv.y, xv.z, xv.w}, ew[4] = {ev.x, ev.y, ev.z, ev.w};
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float y = __uint_as_float((e & 1) ? (yw[e >> 1] & 0xffff0000u) : (yw[e >> 1] << 16));
          const float x = __uint_as_float((e & 1) ? (xw[e >> 1] & 0xffff0000u) : (xw[e >> 1] << 16));
          if (EPI == EPI_GATE_RES) y = rbf(y * __uint_as_float((e & 1) ? (ew[e >> 1] & 0xffff0000u) : (ew[e >> 1] << 16)));   // y * e rounds
          v[e] = x + y;                                            // x + (.) rounds at the pack
        }
        ov.x = pack2bf(v[0], v[1]); ov.y = pack2bf(v[2], v[3]); ov.z = pack2bf(v[4], v[5]); ov.w = pack2bf(v[6], v[7]);
      }
      if (EPI == EPI_BIAS_VPAGES && n >= g.v_col0) {
        const int fr = m / g.rows_per_frame;
        *reinterpret_cast<uint4*>(g.v_dst[fr] + (size_t)(m - fr * g.rows_per_frame) * g.v_ld + (n - g.v_col0)) = ov;
      } else {
        *reinterpret_cast<uint4*>(g.C + (size_t)m * g.ldc + n) = ov;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                           // the next slab's writes stay behind this slab's reads
  });
  if constexpr (GEMM_ABL & 32) {                               // dev: { prologue, k loop, epilogue } shader cycles per wave, over the output
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tick_epi = __builtin_readcyclecounter();
    __syncthreads();
    if (lane == 0) {
      float* tp = reinterpret_cast<float*>(g.C) + (blockIdx.x * 4 + wave) * 4;
      tp[0] = (float)(tick0 - tick_start);
      tp[1] = (float)(tick_loop - tick0);
      tp[2] = (float)(tick_epi - tick_loop);
      tp[3] = (float)nt;
    }
  }
}

template <int EPI>
hipError_t launch_w64(const GemmArgs& g, hipStream_t s) {
  if (hipError_t e = mmpl_dyn_smem_once(reinterpret_cast<const void*>(gemm_w64_kernel<EPI>), G_SMEM); e != hipSuccess) return e;
  const int tiles = ((g.M + GM - 1) / GM) * ((g.N + GN - 1) / GN);
  GemmArgs g2 = g;
  g2.group = 4;
  hipLaunchKernelGGL(gemm_w64_kernel<EPI>, dim3(tiles), dim3(256), G_SMEM, s, g2);
  return hipGetLastError();
}

}  // namespace

// Does the one-wave-per-SIMD kernel take this problem?  (gemm.hip's launcher asks; everything else stays on its kernels.)
bool mmpl_gemm_w64_accepts(const GemmArgs& g) {
  if (g.batch > 1 || g.M < 1024 || g.N < 256 || g.K < 128 || g.K % 64 != 0 || g.N % 8 != 0) return false;
  if (g.epi == EPI_F32_SCALE) return false;
  if (g.ldc % 8 != 0 || g.lda % 8 != 0 || g.ldw % 8 != 0) return false;
  if ((g.epi == EPI_GATE_RES || g.epi == EPI_RES) && g.ldres % 8 != 0) return false;
  if (g.epi == EPI_GATE_RES && g.gate_frame_stride % 8 != 0) return false;
  if (g.epi == EPI_BIAS_VPAGES && (g.v_col0 % 8 != 0 || g.v_ld % 8 != 0)) return false;
  // 32-bit DMA offsets within a 256-row tile
  return (long long)GM * g.lda * 2 < (1ll << 31) && (long long)GN * g.ldw * 2 < (1ll << 31);
}

hipError_t mmpl_launch_gemm_w64(const GemmArgs& g, hipStream_t s) {
  switch (g.epi) {
    case EPI_BIAS: return launch_w64<EPI_BIAS>(g, s);
    case EPI_BIAS_GELU: return launch_w64<EPI_BIAS_GELU>(g, s);
    case EPI_BIAS_SILU: return launch_w64<EPI_BIAS_SILU>(g, s);
    case EPI_GATE_RES: return launch_w64<EPI_GATE_RES>(g, s);
    case EPI_RES: return launch_w64<EPI_RES>(g, s);
    case EPI_BIAS_VPAGES: return launch_w64<EPI_BIAS_VPAGES>(g, s);
    default: return hipErrorInvalidValue;
  }
}
