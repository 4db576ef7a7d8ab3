// Self-attention forward, 64 query rows per wave: 4 waves x 64 rows, ONE wave per SIMD, the whole 512-entry register file.
//
// Same math, page table, LDS images and fragment layouts as attn_pp_kernel (attention.hip; replaces attention.py:139-185 + the
// K/V gather of causal_fps_model.py:219-227), but every K / V fragment read from LDS feeds TWO MFMAs (the wave's two 32-row
// query blocks "A" and "B"), so LDS fragment traffic and LDS-DMA pieces per MFMA are half of the 8 x 32-row kernels'.
//
// Register plan (per lane).  Accumulator half, named literally in the asm below and never seen by the compiler:
//     a[0:63] O_A   a[64:127] O_B   (4 d-blocks x 16 fp32 each)
//     a[128:159] Q_A   a[160:191] Q_B   (8 hd chunks x 4 regs: B operand of S^T = K.Q^T)
//     a[192:255] V fragments of one KV tile (16 x 4 regs, written straight from LDS by ds_read_b64_tr_b16)
// Architectural half (compiler-allocated, every hot instruction is an `asm volatile` so program order == source order):
//     S_A, S_B (2 x 32 fp32), P_A, P_B (2 x 16 packed bf16 pairs), the 16 K fragments of one tile (64), softmax state.
//
// Schedule.  The two query blocks run HALF A TILE APART; per KV tile j two phases of 32 MFMAs, one barrier per tile:
//     A(j): MFMA  S_A(j) = K(j).Q_A [16]  then  O_A += V(j-1).P_A(j-1) [16]      VALU: softmax of S_B(j-1) -> P_B(j-1)
//           LDS-DMA of tiles j+4 (K) / j+3 (V), 8 pieces per wave, after the barrier
//     B(j): MFMA  S_B(j) = K(j).Q_B [16]  then  O_B += V(j-1).P_B(j-1) [16]      VALU: softmax of S_A(j)   -> P_A(j)
//           LDS reads: K(j+1) fragment i right behind the last MFMA that used K(j) fragment i, V(j) likewise
// so each accumulator is finished 16 MFMAs (>= 512 cycles) before the VALU touches it and each P a phase before its MFMA, and
// every gap between two MFMAs carries <= 5 other instructions (MI355X_MICROARCH.md: a 32x32x16 MFMA hides ~5 issues).
//
// Softmax without a row max on the common path: p = exp2(s*c - m_ref*c) against a per-row reference m_ref that only moves on a
// rare path.  The common path just sums the tile's p; if any lane's partial sum is not <= 2^30 (first tile: m_ref = -inf ->
// +inf) the tile is redone on the slow path: exact row max, m_ref = max(m_ref, max), O and l rescaled, p recomputed.  So p <= 2^30
// always, l >= 1 after the first tile, and the result is the exact softmax up to rounding (fp32 exponent range is never at risk).
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int QB = 256, KVB = 64, TILE = KVB * 256;                   // K ring [0, RING*TILE), V ring behind it
constexpr float L_BOUND = 1073741824.f;                               // 2^30

template <int I> using ic = std::integral_constant<int, I>;
template <class F, int... I> MMPL_DEV void sfor_(F&& f, std::integer_sequence<int, I...>) { (f(ic<I>{}), ...); }
template <int N, class F> MMPL_DEV void sfor(F&& f) { sfor_(f, std::make_integer_sequence<int, N>{}); }

#define A10(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
#define ALL_AGPRS                                                                                                                  \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", A10(1), A10(2), A10(3), A10(4), A10(5), A10(6), A10(7), A10(8),     \
      A10(9), A10(10), A10(11), A10(12), A10(13), A10(14), A10(15), A10(16), A10(17), A10(18), A10(19), A10(20), A10(21), A10(22), \
      A10(23), A10(24), "a250", "a251", "a252", "a253", "a254", "a255"

constexpr int AO = 0, AQ = 128, AV = 192;      // accumulator-file map

struct W64 {
  f32x16 S[2][2];      // [query block][kv half]
  u32x4 P[2][4];       // [query block][16-row kv step]: 8 bf16 = B operand of O^T += V^T.P^T
  bf16x8 kf[16];       // K fragments of one tile: i = 2*chunk + half
  float l[2], mref[2], nmc[2];
  float la, lb, p0, p1;
};

// ---- single-instruction helpers (volatile: the hot loop is emitted in source order; the compiler only allocates registers)
template <int X, int G> MMPL_DEV void mfma_qk(W64& w) {   // S_X[h] (+)= K frag G . Q_X[chunk]
  constexpr int c = G >> 1, h = G & 1, qa = AQ + 32 * X + 4 * c;
  if constexpr (c == 0)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(w.S[X][h]) : "v"(w.kf[G]), "i"(qa), "i"(qa + 3));
  else
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(w.S[X][h]) : "v"(w.kf[G]), "i"(qa), "i"(qa + 3));
}
template <int X, int G> MMPL_DEV void mfma_pv(W64& w) {   // O_X[nb] += V frag G . P_X[ks]
  constexpr int ks = G >> 2, nb = G & 3, oa = AO + 64 * X + 16 * nb, va = AV + 4 * G;
  asm volatile("v_mfma_f32_32x32x16_bf16 a[%c0:%c1], a[%c2:%c3], %4, a[%c0:%c1]" ::"i"(oa), "i"(oa + 15), "i"(va), "i"(va + 3),
               "v"(w.P[X][ks]));
}
template <int G> MMPL_DEV void lds_k(W64& w, uint32_t addr) {   // fragment G = (chunk, half): addr already carries the chunk
  asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(w.kf[G]) : "v"(addr), "i"((G & 1) * 32 * 256));
}
template <int G, int IMM> MMPL_DEV void lds_k_next(W64& w, uint32_t& addr) {   // move addr to the next chunk, then read
  asm volatile("v_xor_b32 %1, %2, %1\n\tds_read_b128 %0, %1 offset:%c3" : "=v"(w.kf[G]), "+v"(addr) : "i"(IMM), "i"((G & 1) * 32 * 256));
}
template <int G> MMPL_DEV void lds_v(uint32_t addr) {           // fragment G = (ks, nb): addr already carries nb
  constexpr int va = AV + 4 * G, off = (G >> 2) * 16 * 256;
  asm volatile("ds_read_b64_tr_b16 a[%c1:%c2], %0 offset:%c3\n\tds_read_b64_tr_b16 a[%c4:%c5], %0 offset:%c6" ::"v"(addr), "i"(va),
               "i"(va + 1), "i"(off), "i"(va + 2), "i"(va + 3), "i"(off + 8 * 256));
}
template <int IMM> MMPL_DEV uint32_t v_xor(uint32_t a) { uint32_t r; asm volatile("v_xor_b32 %0, %1, %2" : "=v"(r) : "i"(IMM), "v"(a)); return r; }
// One asm statement per group of dependent VALU ops: hipcc pads a wait state wherever one statement's output feeds the next
// statement, and knows nothing about the instructions inside (the trans -> VALU use distance is kept by the gap structure).
MMPL_DEV void v_scale_exp2(float& p0, float& p1, float s0, float s1, float c, float nmc) {     // p = exp2(s*c + nmc)
  asm volatile("v_fma_f32 %0, %2, %4, %5\n\tv_fma_f32 %1, %3, %4, %5\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %1, %1"
               : "=&v"(p0), "=&v"(p1) : "v"(s0), "v"(s1), "v"(c), "v"(nmc));
}
MMPL_DEV uint32_t v_sum_pack(float& la, float& lb, float p0, float p1) {                        // la += p0, lb += p1, pack(p0, p1)
  uint32_t r;
  asm volatile("v_add_f32 %0, %0, %3\n\tv_add_f32 %1, %1, %4\n\tv_cvt_pk_bf16_f32 %2, %3, %4" : "+v"(la), "+v"(lb), "=v"(r) : "v"(p0), "v"(p1));
  return r;
}
MMPL_DEV uint32_t v_pack(float p0, float p1) { uint32_t r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(p0), "v"(p1)); return r; }

// LDS-DMA of one 1 KiB piece (4 rows x 256 B): M0 = LDS destination (written in the same statement that uses it; nothing the
// compiler emits in this kernel reads M0), 1 wait state between the M0 write and the DMA
template <int OFF> MMPL_DEV void dma16(const void* base, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_add_u32 m0, %2, %c3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_dst), "i"(OFF)
               : "memory", "scc");
}

// ---- one gap's share of the softmax of query block X (tile already masked): pair q = G >> 1 of the 16 register pairs
template <int X, int G> MMPL_DEV void sm_gap(W64& w, float c) {
  constexpr int q = G >> 1, h = q >> 3, e = (q & 7) * 2, ks = 2 * h + (e >> 3), wd = (e & 7) >> 1;
  if constexpr ((G & 1) == 0) {
    v_scale_exp2(w.p0, w.p1, w.S[X][h][e], w.S[X][h][e + 1], c, w.nmc[X]);
  } else if constexpr (q == 0) {
    w.la = w.p0;
    w.lb = w.p1;
    w.P[X][ks][wd] = v_pack(w.p0, w.p1);
  } else {
    w.P[X][ks][wd] = v_sum_pack(w.la, w.lb, w.p0, w.p1);
  }
}

template <int X> MMPL_DEV void mask_tail(W64& w, int valid, int hi) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int kv = 8 * (r >> 2) + 4 * hi + (r & 3);
    if (kv >= valid) w.S[X][0][r] = -INFINITY;
    if (kv + 32 >= valid) w.S[X][1][r] = -INFINITY;
  }
}

// Slow path of one tile of query block X: exact row max, move the reference, rescale O_X and l, recompute P and the tile sum.
template <int X> MMPL_DEV float sm_slow(W64& w, float c) {
  float mx = w.S[X][0][0];
#pragma unroll
  for (int r = 1; r < 16; ++r) mx = fmaxf(mx, w.S[X][0][r]);
#pragma unroll
  for (int r = 0; r < 16; ++r) mx = fmaxf(mx, w.S[X][1][r]);
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float m_new = fmaxf(w.mref[X], mx);
  const float alpha = __builtin_amdgcn_exp2f((w.mref[X] - m_new) * c);      // first tile: exp2(-inf) = 0 (O = l = 0)
  w.mref[X] = m_new;
  w.nmc[X] = -m_new * c;
  w.l[X] *= alpha;
  sfor<64>([&w, alpha](auto ii) {
    constexpr int i = decltype(ii)::value;
    float t;
    asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(t) : "i"(AO + 64 * X + i));
    t *= alpha;
    asm volatile("v_accvgpr_write_b32 a[%c0], %1" ::"i"(AO + 64 * X + i), "v"(t));
  });
  float lt = 0.f;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float a0 = __builtin_amdgcn_exp2f(fmaf(w.S[X][h][8 * cc + 2 * q], c, w.nmc[X]));
        const float a1 = __builtin_amdgcn_exp2f(fmaf(w.S[X][h][8 * cc + 2 * q + 1], c, w.nmc[X]));
        lt += a0 + a1;
        w.P[X][2 * h + cc][q] = pack2bf(a0, a1);
      }
  return lt;
}

// end of a tile of query block X: accept the optimistic sum or redo the tile on the slow path
template <int X> MMPL_DEV void sm_finish(W64& w, float c) {
  float lt = w.la + w.lb;
  if (__any(!(lt <= L_BOUND))) lt = sm_slow<X>(w, c);
  w.l[X] += lt;
}

// RING: tiles per LDS ring = how far ahead the LDS-DMA runs (K(j+RING), V(j+RING-1) are issued during tile j).
// ABL (dev builds only): timing ablations, results are garbage -- 1 no LDS-DMA in the loop, 2 no softmax, 4 no fragment reads.
template <int RING, int ABL, bool SPLIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_w64_kernel(AttnArgs a, int local_base, int sp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;

  const int n_qb = (a.Lq + QB - 1) / QB;
  int head, qb, part = 0, tail_idx = 0;
  if (SPLIT) {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    part = idx % sp;
    tail_idx = idx / sp;
    const int local = local_base + tail_idx;
    head = xcd + 8 * (local / n_qb);
    qb = local % n_qb;
    tail_idx = xcd * (gridDim.x / (8 * sp)) + tail_idx;
  } else if ((a.H & 7) == 0) {
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    head = xcd + 8 * (local / n_qb);
    qb = local % n_qb;
  } else {
    head = blockIdx.x / n_qb;
    qb = blockIdx.x % n_qb;
  }

  // the accumulator file is ours: this statement makes the kernel descriptor allocate all 256 entries
  asm volatile("s_nop 0" ::: ALL_AGPRS);
  sfor<128>([](auto ii) { asm volatile("v_accvgpr_write_b32 a[%c0], 0" ::"i"(AO + decltype(ii)::value)); });

  // ---- Q fragments -> a[128:191]: lane (l31, hi) holds Q[row][16c + 8*hi .. +8] of its row in block A and in block B
  {
    const bf16_t* qbase = a.q + head * 128 + 8 * hi;
#pragma unroll
    for (int X = 0; X < 2; ++X) {
      const int qrow = min(qb * QB + wave * 64 + 32 * X + l31, a.Lq - 1);
      const bf16_t* qp = qbase + (size_t)qrow * a.ldq;
      u32x4 qf[8];
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) qf[cc] = *reinterpret_cast<const u32x4*>(qp + 16 * cc);
      if (X == 0)
        sfor<32>([&qf](auto ii) { constexpr int i = decltype(ii)::value; asm volatile("v_accvgpr_write_b32 a[%c0], %1" ::"i"(AQ + i), "v"(qf[i >> 2][i & 3])); });
      else
        sfor<32>([&qf](auto ii) { constexpr int i = decltype(ii)::value; asm volatile("v_accvgpr_write_b32 a[%c0], %1" ::"i"(AQ + 32 + i), "v"(qf[i >> 2][i & 3])); });
    }
  }

  const int tiles_pp = (a.page_rows + KVB - 1) / KVB;
  const int T_all = a.n_pages * tiles_pp;
  const int t_first = SPLIT ? (int)((long long)part * T_all / sp) : 0;
  const int T = SPLIT ? (int)((long long)(part + 1) * T_all / sp) - t_first : T_all;     // tiles of THIS block

  // ---- LDS-DMA roles: wave w moves pieces w, w+4, w+8, w+12 (4 rows x 256 B each) of every K tile and of every V tile.  The
  // LDS image is lane-linear, so the bank swizzle goes on the per-lane SOURCE chunk (K: chunk ^= row & 15; V: chunk ^=
  // (row & 3) << 2; row & 15 = 4*wave + drow for all four pieces) and again on the fragment reads.
  const int drow = lane >> 4, dchunk = lane & 15;
  const int prow = 4 * wave + drow;                     // row of piece 0; piece k: + 16 k
  uint32_t dko[4], dvo[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    dko[k] = (uint32_t)((prow + 16 * k) * a.ldk + ((dchunk ^ prow) << 3)) * 2u;
    dvo[k] = (uint32_t)((prow + 16 * k) * a.ldv + ((dchunk ^ (drow << 2)) << 3)) * 2u;
  }
  // issue cursors (next K tile / next V tile to fetch); they stop on the block's last tile, which is then simply re-fetched,
  // so every event issues exactly 8 pieces and the counted waits below never change
  struct Cur { const bf16_t* ptr; int row0, pg, t; };
  Cur ck, cv;
  ck.pg = cv.pg = t_first / tiles_pp;
  ck.row0 = cv.row0 = (t_first % tiles_pp) * KVB;
  ck.t = cv.t = 0;
  ck.ptr = a.k_pages[ck.pg] + (size_t)ck.row0 * a.ldk + head * 128;
  cv.ptr = a.v_pages[cv.pg] + (size_t)cv.row0 * a.ldv + head * 128;
  uint32_t kslot = wave * 1024, vslot = RING * TILE + wave * 1024;          // LDS byte address of this wave's piece 0 in the ring slot
  auto issue_k = [&](auto kk) {                         // piece k of the K tile under the cursor
    constexpr int k = decltype(kk)::value;
    uint32_t off = dko[k];
    if (ck.row0 + KVB > a.page_rows)                    // ragged last tile of a page: clamp rows (masked in the softmax)
      off = (uint32_t)(min(prow + 16 * k, a.page_rows - 1 - ck.row0) * a.ldk + ((dchunk ^ prow) << 3)) * 2u;
    dma16<4096 * k>(ck.ptr, off, kslot);
  };
  auto issue_v = [&](auto kk) {
    constexpr int k = decltype(kk)::value;
    uint32_t off = dvo[k];
    if (cv.row0 + KVB > a.page_rows)
      off = (uint32_t)(min(prow + 16 * k, a.page_rows - 1 - cv.row0) * a.ldv + ((dchunk ^ (drow << 2)) << 3)) * 2u;
    dma16<4096 * k>(cv.ptr, off, vslot);
  };
  auto advance = [&](Cur& cu, const bf16_t* const* pages, int ld, uint32_t& slot, uint32_t ring0) {
    slot += TILE;
    if (slot >= ring0 + RING * TILE) slot -= RING * TILE;
    if (cu.t + 1 < T) {
      ++cu.t;
      cu.row0 += KVB;
      cu.ptr += (size_t)KVB * ld;
      if (cu.row0 >= a.page_rows) {
        cu.row0 = 0;
        ++cu.pg;
        cu.ptr = pages[cu.pg] + head * 128;
      }
    }
    asm volatile("" : "+s"(cu.ptr));
  };

  W64 w;
  w.l[0] = w.l[1] = 0.f;
  w.mref[0] = w.mref[1] = -INFINITY;
  w.nmc[0] = w.nmc[1] = INFINITY;
  const float c = a.scale * 1.4426950408889634f;

  // ---- per-lane fragment read offsets (swizzled); chunk / d-block enter by XOR: koff(cs) = koff(0) ^ 32 cs, voff(nb) = voff(0) ^ 64 nb
  const uint32_t kbase = l31 * 256 + 32 * ((l31 & 15) >> 1) + 16 * (hi ^ (l31 & 1));
  const int i16 = lane & 15, g16 = (lane >> 4) & 1;
  const uint32_t vbase = RING * TILE + (4 * hi + (i16 >> 2)) * 256 + 64 * (i16 >> 2) + 32 * g16 + 8 * (i16 & 3);

  // ---- prologue: events -RING .. -1 of the DMA stream (event e = { K(e+RING), V(e+RING-1) }), then the K(0) fragments
  sfor<4>(issue_k);
  advance(ck, a.k_pages, a.ldk, kslot, 0);
#pragma unroll 1
  for (int e = 1; e < RING; ++e) {
    sfor<4>(issue_k);
    sfor<4>(issue_v);
    advance(ck, a.k_pages, a.ldk, kslot, 0);
    advance(cv, a.v_pages, a.ldv, vslot, RING * TILE);
  }
  asm volatile("s_waitcnt vmcnt(%c0)\n\ts_barrier" ::"i"(8 * (RING - 1)) : "memory");
  sfor<16>([&w, kbase](auto gi) { constexpr int g = decltype(gi)::value; lds_k<g>(w, kbase ^ (32 * (g >> 1))); });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  uint32_t rk = TILE, rv = 0;                       // ring offsets of the tiles the NEXT B phase reads: K(j+1), V(j)
  int crow[2];
  crow[0] = crow[1] = (t_first % tiles_pp) * KVB;   // first kv row (within its page) of the tile each block's softmax is at
  uint32_t vaddr[4];

  // ---- phase A(j): MFMAs of query block A, softmax of query block B
  auto phase_a = [&](auto has_qk, auto has_pv) {
    constexpr bool QK = decltype(has_qk)::value, PV = decltype(has_pv)::value;
    if constexpr (PV) {
      const int valid = a.page_rows - crow[1];
      if (valid < KVB) mask_tail<1>(w, valid, hi);
      crow[1] += KVB;
      if (crow[1] >= a.page_rows) crow[1] = 0;
    }
    sfor<32>([&](auto gi) {
      constexpr int g = decltype(gi)::value;
      if constexpr (g < 16) { if constexpr (QK) mfma_qk<0, g>(w); }
      else { if constexpr (PV) mfma_pv<0, g - 16>(w); }
      if constexpr (PV && !(ABL & 2)) sm_gap<1, g>(w, c);
      if constexpr (PV && !(ABL & 4) && g == 0) lds_v<15>(vaddr[3]);
      if constexpr (g == 3) {
        if constexpr (QK) asm volatile("s_waitcnt vmcnt(%c0) lgkmcnt(0)\n\ts_barrier" ::"i"(8 * (RING - 2)) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      if constexpr (QK && !(ABL & 1) && g >= 5 && g <= 11 && (g & 1)) issue_k(ic<(g - 5) / 2>{});
      if constexpr (QK && !(ABL & 1) && g >= 13 && g <= 19 && (g & 1)) issue_v(ic<(g - 13) / 2>{});
      if constexpr (QK && g == 21) {
        advance(ck, a.k_pages, a.ldk, kslot, 0);
        advance(cv, a.v_pages, a.ldv, vslot, RING * TILE);
      }
    });
    if constexpr (PV && !(ABL & 2)) sm_finish<1>(w, c);
  };
  // ---- phase B(j): MFMAs of query block B, softmax of query block A, fragment reads of K(j+1) and V(j)
  auto phase_b = [&](auto has_qk, auto has_pv) {
    constexpr bool QK = decltype(has_qk)::value, PV = decltype(has_pv)::value;
    uint32_t kaddr = 0;
    if constexpr (QK) {
      const int valid = a.page_rows - crow[0];
      if (valid < KVB) mask_tail<0>(w, valid, hi);
      crow[0] += KVB;
      if (crow[0] >= a.page_rows) crow[0] = 0;
      kaddr = kbase + rk;
      vaddr[0] = vbase + rv;
    }
    sfor<32>([&](auto gi) {
      constexpr int g = decltype(gi)::value;
      if constexpr (g < 16) { if constexpr (QK) mfma_qk<1, g>(w); }
      else { if constexpr (PV) mfma_pv<1, g - 16>(w); }
      if constexpr (QK) {
        if constexpr (!(ABL & 2)) sm_gap<0, g>(w, c);
        if constexpr (!(ABL & 4) && g >= 1 && g <= 16) {
          constexpr int f = g - 1;                       // K(j) fragment f was last read by MFMA f of this phase
          if constexpr ((f & 1) == 0 && f > 0) lds_k_next<f, (32 * (f >> 1)) ^ (32 * ((f >> 1) - 1))>(w, kaddr);
          else lds_k<f>(w, kaddr);
        }
        if constexpr (g >= 13 && g <= 15) vaddr[g - 12] = v_xor<64 * (g - 12)>(vaddr[0]);
        if constexpr (g == 22) asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory");     // the 16 K reads are older than the 10 V reads since
        if constexpr (!(ABL & 4) && g >= 17) lds_v<g - 17>(vaddr[(g - 17) & 3]);
      }
    });
    if constexpr (QK) {
      if constexpr (!(ABL & 2)) sm_finish<0>(w, c);
      rk += TILE; if (rk >= RING * TILE) rk = 0;
      rv += TILE; if (rv >= RING * TILE) rv = 0;
    }
  };
  using std::true_type;
  using std::false_type;

  phase_a(true_type{}, false_type{});
  phase_b(true_type{}, false_type{});
#pragma unroll 1
  for (int j = 1; j < T; ++j) {
    phase_a(true_type{}, true_type{});
    phase_b(true_type{}, true_type{});
  }
  phase_a(false_type{}, true_type{});
  phase_b(false_type{}, true_type{});
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

  // ---- epilogue: lane (q = l31, hi) holds O_X[q][32*nb + 8*g + 4*hi + {0..3}] in a[64 X + 16 nb + 4 g ..+3]
#pragma unroll
  for (int X = 0; X < 2; ++X) {
    const float l_tot = w.l[X] + __shfl_xor(w.l[X], 32, 64);
    float o[64];
    if (X == 0) sfor<64>([&o](auto ii) { constexpr int i = decltype(ii)::value; asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(o[i]) : "i"(AO + i)); });
    else sfor<64>([&o](auto ii) { constexpr int i = decltype(ii)::value; asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(o[i]) : "i"(AO + 64 + i)); });
    const int rr = wave * 64 + 32 * X + l31;
    if (SPLIT) {
      // partial of this KV range: O (fp32, relative to m_ref), then m, l per row -- [tail block][part][256 rows][128 + 2]
      float* pbase = a.split_ws + ((size_t)tail_idx * sp + part) * (QB * 130);
      float* op = pbase + (size_t)rr * 128 + 4 * hi;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(op + 32 * nb + 8 * g) = f32x4{o[16 * nb + 4 * g], o[16 * nb + 4 * g + 1], o[16 * nb + 4 * g + 2], o[16 * nb + 4 * g + 3]};
      if (hi == 0) {
        pbase[QB * 128 + rr] = w.mref[X];
        pbase[QB * 129 + rr] = l_tot;
      }
    } else {
      const float inv = 1.0f / l_tot;
      const int q_out = qb * QB + rr;
      if (q_out < a.Lq) {
        bf16_t* op = a.o + (size_t)q_out * a.ldo + head * 128 + 4 * hi;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            uint2 ww;
            ww.x = pack2bf(o[16 * nb + 4 * g] * inv, o[16 * nb + 4 * g + 1] * inv);
            ww.y = pack2bf(o[16 * nb + 4 * g + 2] * inv, o[16 * nb + 4 * g + 3] * inv);
            *reinterpret_cast<uint2*>(op + 32 * nb + 8 * g) = ww;
          }
      }
    }
  }
}

}  // namespace

namespace {
template <int RING, int ABL>
const void* w64_sym(int split) {
  return split ? reinterpret_cast<const void*>(attn_w64_kernel<RING, ABL, true>) : reinterpret_cast<const void*>(attn_w64_kernel<RING, ABL, false>);
}
template <int RING, int ABL>
void w64_launch(const AttnArgs& a, int blocks, int local_base, int sp, bool split, hipStream_t s) {
  if (split) hipLaunchKernelGGL((attn_w64_kernel<RING, ABL, true>), dim3(blocks), dim3(256), 2 * RING * TILE, s, a, local_base, sp);
  else hipLaunchKernelGGL((attn_w64_kernel<RING, ABL, false>), dim3(blocks), dim3(256), 2 * RING * TILE, s, a, local_base, sp);
}
// dev knobs, read once per process: MMPL_W64_RING (2..4), MMPL_W64_ABL (timing ablations, garbage results)
int w64_ring() { static const int r = getenv("MMPL_W64_RING") ? atoi(getenv("MMPL_W64_RING")) : 4; return r; }
int w64_abl() { static const int r = getenv("MMPL_W64_ABL") ? atoi(getenv("MMPL_W64_ABL")) : 0; return r; }
#define W64_DISPATCH(expr)                                                   \
  do {                                                                       \
    const int r_ = w64_ring(), b_ = w64_abl();                               \
    if (b_ == 1) { constexpr int RING = 2, ABL = 1; expr; }                  \
    else if (b_ == 2) { constexpr int RING = 2, ABL = 2; expr; }             \
    else if (b_ == 4) { constexpr int RING = 2, ABL = 4; expr; }             \
    else if (b_ == 6) { constexpr int RING = 2, ABL = 6; expr; }             \
    else if (b_ == 7) { constexpr int RING = 2, ABL = 7; expr; }             \
    else if (r_ == 2) { constexpr int RING = 2, ABL = 0; expr; }             \
    else if (r_ == 3) { constexpr int RING = 3, ABL = 0; expr; }             \
    else { constexpr int RING = 4, ABL = 0; expr; }                          \
  } while (0)
}  // namespace

int mmpl_attention_w64_smem() { int v = 0; W64_DISPATCH(v = 2 * RING * TILE; (void)ABL); return v; }
const void* mmpl_attention_w64_symbol(int split) { const void* p = nullptr; W64_DISPATCH(p = (w64_sym<RING, ABL>(split))); return p; }
void mmpl_launch_attention_w64(const AttnArgs& a, int blocks, int local_base, int sp, bool split, hipStream_t s) {
  W64_DISPATCH((w64_launch<RING, ABL>(a, blocks, local_base, sp, split, s)));
}
