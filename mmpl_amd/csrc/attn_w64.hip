// Self-attention forward, 64 query rows per wave: 4 waves x 64 rows, ONE wave per SIMD, the whole 512-entry register file.
//
// Same math, page table and fragment layouts as attn_fwd_kernel (attention.hip; replaces attention.py:139-185 + the
// K/V gather of causal_fps_model.py:219-227), but every K / V fragment read from LDS feeds TWO MFMAs (the wave's two 32-row
// query blocks "A" and "B"), so LDS fragment traffic and LDS-DMA pieces per MFMA are half of the 8 x 32-row kernels'.
//
// Register plan (per lane).  Accumulator half, named literally in the asm below and never seen by the compiler:
//     a[0:63] O_A   a[64:127] O_B   (4 d-blocks x 16 fp32 each)
//     a[128:159] Q_A   a[160:191] Q_B   (8 hd chunks x 4 regs: B operand of S^T = K.Q^T; PRESCALED by scale*log2(e))
//     a[192:255] V fragments of one KV tile (16 x 4 regs, written straight from LDS by ds_read_b64_tr_b16)
// Architectural half (compiler-allocated; every hot instruction is an `asm volatile`, so program order == source order):
//     S_A, S_B (2 x 32 fp32), P_A, P_B (2 x 16 packed bf16 pairs), the 16 K fragments of one tile (64), softmax state.
//
// Schedule.  The two query blocks run HALF A TILE APART; per KV tile j two phases of 32 MFMAs, one barrier per tile:
//     A(j): S_A(j) = K(j).Q_A [16 MFMAs]  then  O_A += V(j-1).P_A(j-1) [16]     + LDS-DMA of K(j+4) / V(j+3), 8 pieces per wave
//     B(j): S_B(j) = K(j).Q_B [16]        then  O_B += V(j-1).P_B(j-1) [16]     + fragment reads of K(j+1) and V(j), each right
//                                                                                  behind the last MFMA that used its register
// Everything that is not an MFMA is placed by tools/gen_attn_w64.py (attn_w64_sched.inc) into the gaps between them, <= ~5
// issue slots per gap (MI355X_MICROARCH.md: that is what one 32x32x16 MFMA hides from the wave that issued it): the softmax
// of S_B(j-1) streams through A(j) and the head of B(j), the softmax of S_A(j) through the tail of A(j) and B(j), so each
// accumulator is finished >= 2 MFMAs before the VALU touches it and each P >= 2 gaps before its MFMA.
//
// Softmax without a row max on the common path.  Q is prescaled, so S is in log2 units and p = exp2(S - m_ref) against a per-row
// reference m_ref.  FAST pass: m_ref is FIXED for the whole pass and costs nothing in the loop -- it is the C operand of each score
// tile's first MFMA (the register tile that also carries the ragged-tile mask holds -m_ref instead of 0), so S arrives already
// shifted: one v_exp, one add and half a cvt_pk per score, and NOTHING per tile that could branch.  m_ref = 64 + the largest score of
// the block's first FOUR KV tiles (the ones the prologue has in the ring before the pipeline starts: 128 extra MFMAs per 256-row
// block; a lane's two query rows share one value).
// Every p is >= 0, so the row sums only grow: one look at them after the last tile tells whether any exponential overflowed
// (2^-100 <= l <= 2^100 is required, NaN fails) -- i.e. whether some later score beat the first tile's maximum by more than ~150
// (log2 units; ~100 nats).  Round 3 used m_ref = 0, which holds for unit-gain synthetic weights only: with QK-norm gains x8 91 % of
// the blocks failed it and ran twice (profiles/r04a_bench_14B_720p_heavy_tail_before_row_reference.json); with this reference 0 %
// at gains x3, 0.5 % at x5, 26.5 % at x8 (one tile instead of four: 32.8 %; profiles/r04e_attn_ref_tiles_ab.log).  If any row of the block
// fails, the whole block is redone by the GENERAL pass: p = exp2(S - m_ref) with a RUNNING reference (one more VALU per score), a
// tile is accepted when every lane's partial row sum is <= 2^30, otherwise it is redone on a slow path (exact row max,
// m_ref = max(m_ref, max), O and l rescaled through v_accvgpr moves, p recomputed).  So the result is the exact softmax up to rounding
// for any input (tests: spiked scores far beyond both bounds); inputs that a FAST pass cannot hold cost that block two passes.
//
// Nothing in the steady loop branches except its back edge.  Everything that happens once per page -- the K / V cursors
// switching buffer descriptors, the ragged last tile's mask, a cursor parking on the block's last tile -- is decided between
// RUNS of identical iterations (Ctx::plan): a run ends where the next such event is due.  The mask is the C operand of each
// score tile's first MFMA: a register tile that is all zero except during a page's ragged last tile (-inf on the rows past
// the page's end, whose K and V rows the LDS-DMA's range check zero-filled).
#include <stdlib.h>

#include "common.h"
#include "dev_knobs.h"
#include "kernels.h"
#include "w64_util.h"

namespace {

constexpr int QB = 256, KVB = 64, TILE = KVB * 256, RING = 4;         // K ring [0, 64 KiB), V ring [64 KiB, 128 KiB)
constexpr int W64_SMEM = 2 * RING * TILE + 64;     // + the block's redo flag
constexpr float BOUND_GEN = 1073741824.f;                              // 2^30: a GENERAL tile's partial row sums
constexpr float pow2f(int e) { return e == 0 ? 1.f : (e > 0 ? 2.f * pow2f(e - 1) : 0.5f * pow2f(e + 1)); }
// Build parameters and timing ablations: dev_knobs.h.  A product build has the shipped constants and W64_ABL == 0, so every
// `if constexpr (ABL(bits))` below is dead code the compiler drops.
#define ABL(bits) ((W64_ABL & (bits)) != 0)
// 2^-100 <= l <= 2^100: a FAST pass's final row sums.  The window is where the fp32 exponent range puts it: above, O <= l |v| must stay finite;
// below, every p under 2^-126 is flushed to zero -- N keys lose at most N 2^-126, which is <= 2^-9.8 of l (under a bf16 ulp of the output)
// for N = 75 600 keys exactly when l >= 2^-100.  Lowering the bound to 2^-124 cut the heavy-tail x8 redo rate from 26.7 to 18.1 % of the
// blocks (profiles/r05c_attn_fast_window_offsets.log) but voids that guarantee; moving the offset only trades one side for the other.
constexpr float FAST_L_MIN = pow2f(-W64_LMIN_EXP), FAST_L_MAX = pow2f(W64_LMAX_EXP);
constexpr float FAST_REF_OFFSET = W64_REF_OFFSET;   // FAST pass: m_ref = (largest score of the block's first FAST_REF_TILES KV tiles) + this
constexpr int FAST_REF_TILES = W64_REF_TILES;  // 1 .. 4 (= RING: the tiles the prologue fetches before the pipeline starts)

using w64::sfor;
#define ALL_AGPRS MMPL_ALL_AGPRS

constexpr int AO = 0, AQ = 128, AV = 192;      // accumulator-file map

// All state of a wave.  Passed by reference through always-inlined members, so every field ends up in a register (VGPR or,
// when provably wave-uniform, SGPR); arrays are only ever indexed with compile-time constants.
struct Ctx {
  // ---- vector state
  f32x16 S[2][2];        // [query block][kv half]
  f32x16 M[2];           // [kv half] C operand of a score tile's first MFMA: 0, or -inf on the rows past a page's end
  u32x4 P[2][4];         // [query block][16-row kv step]: 8 bf16 = B operand of O^T += V^T.P^T
  bf16x8 kf[16];         // K fragments of one tile: i = 2*chunk + half
  float l[2], mref[2];   // running row sum (this lane's 32 kv columns of every tile) and reference (log2 units)
  float mbase;           // what an unmasked entry of the C-operand tile M holds: -m_ref in a FAST pass, 0 in a GENERAL one
  float la[2], lb[2];    // the tile's partial sums (even / odd register of each pair)
  float t[2][2][2];      // exp results in flight: [stream][pair parity][element]
  uint32_t dko[4], dvo[4];   // per-piece LDS-DMA source offsets within a tile (constant)
  uint32_t kbase, vbase, kaddr, vaddr[4];
  int hi;
  // ---- wave-uniform state
  const bf16_t* const* k_pages; const bf16_t* const* v_pages;
  int ldk, ldv, head, T;
  const int* rows_each;                        // rows of every page (kernel argument array)
  u32x4 ksrd, vsrd;                            // buffer descriptors of the cursors' pages (this head's 256-byte column)
  uint32_t ksoff, vsoff, tile_bytes_k, tile_bytes_v;   // byte offset of the cursor tile's first row within its page
  uint32_t kstep, vstep;                       // what an iteration adds to ksoff / vsoff: the tile's bytes, 0 once parked
  int t_first, masked;                         // masked: M is not all zero
  uint32_t kslot, vslot;                       // LDS address of this wave's piece 0 in the slot the cursor tile goes to
  uint32_t rk, rv;                             // ring offsets of the tiles the next B phase reads (K(j+1), V(j))
  int first[2];                                // GENERAL pass: stream has not finished its first tile yet
  int use_mem;                                 // FAST pass: take m_ref from mem_ref (the block's history) instead of sampling the first tiles
  float mem_ref;                               // ... this lane's remembered reference (log2 units; an integer value)
  int prow, drow, dchunk;
  uint32_t wave_slot;                          // this wave's first piece within a ring slot

  // ---------------------------------------------------------------- MFMAs
  template <int X, int G> MMPL_DEV void mfma_qk() {      // S_X[h] (+)= K frag G . Q_X[chunk]
    constexpr int c = G >> 1, h = G & 1, qa = AQ + 32 * X + 4 * c;
    if constexpr (c == 0)
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %4" : "=&v"(S[X][h]) : "v"(kf[G]), "i"(qa), "i"(qa + 3), "v"(M[h]));
    else
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(S[X][h]) : "v"(kf[G]), "i"(qa), "i"(qa + 3));
  }
  template <int X, int G> MMPL_DEV void mfma_pv() {      // O_X[nb] += V frag G . P_X[ks]
    constexpr int ks = G >> 2, nb = G & 3, oa = AO + 64 * X + 16 * nb, va = AV + 4 * G;
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c0:%c1], a[%c2:%c3], %4, a[%c0:%c1]" ::"i"(oa), "i"(oa + 15), "i"(va), "i"(va + 3),
                 "v"(P[X][ks]));
  }
  // ---------------------------------------------------------------- LDS fragment reads
  MMPL_DEV void addr_k() { if constexpr (ABL(4)) return; asm volatile("v_add_u32 %0, %1, %2" : "=v"(kaddr) : "s"(rk), "v"(kbase)); }
  MMPL_DEV void addr_v() {
    if constexpr (ABL(4)) return;
    asm volatile("v_add_u32 %0, %4, %5\n\tv_xor_b32 %1, 64, %0\n\tv_xor_b32 %2, 0x80, %0\n\tv_xor_b32 %3, 0xc0, %0"
                 : "=&v"(vaddr[0]), "=&v"(vaddr[1]), "=&v"(vaddr[2]), "=&v"(vaddr[3]) : "s"(rv), "v"(vbase));
  }
  template <int G> MMPL_DEV void lds_k() {               // fragment G = (chunk, half); the chunk enters the address by XOR
    if constexpr (ABL(4)) return;
    constexpr int off = (G & 1) * 32 * 256, cs = G >> 1;
    if constexpr ((G & 1) == 0 && G > 0)
      asm volatile("v_xor_b32 %1, %2, %1\n\tds_read_b128 %0, %1 offset:%c3" : "=v"(kf[G]), "+v"(kaddr) : "i"((32 * cs) ^ (32 * (cs - 1))), "i"(off));
    else
      asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(kf[G]) : "v"(kaddr), "i"(off));
  }
  template <int G> MMPL_DEV void lds_v() {               // fragment G = (ks, nb) -> a[192 + 4G ..]
    if constexpr (ABL(4)) return;
    constexpr int va = AV + 4 * G, off = (G >> 2) * 16 * 256;
    asm volatile("ds_read_b64_tr_b16 a[%c1:%c2], %0 offset:%c3\n\tds_read_b64_tr_b16 a[%c4:%c5], %0 offset:%c6" ::"v"(vaddr[G & 3]),
                 "i"(va), "i"(va + 1), "i"(off), "i"(va + 2), "i"(va + 3), "i"(off + 8 * 256));
  }
  template <int N> MMPL_DEV void wait_lgkm() { if constexpr (ABL(8)) return; asm volatile("s_waitcnt lgkmcnt(%c0)" ::"i"(N) : "memory"); }
  MMPL_DEV void wait_lgkm0() { if constexpr (ABL(8)) return; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
  // K(j+1) and V(j) (DMA event j-3) have landed when at most the 16 pieces of events j-1 and j-2 are outstanding -- two whole
  // tile times of latency cover (with one, vmcnt(8), the wait costs ~150 cycles per tile: measured); all of this wave's
  // fragment reads of the slots event j is about to overwrite are complete (lgkmcnt 0)
  MMPL_DEV void barrier() { if constexpr (ABL(8)) return; asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

  // ---------------------------------------------------------------- LDS-DMA (event j = { K(j+4), V(j+3) })
  // A piece = 4 rows x 256 B = one buffer_load_dwordx4 ... lds.  Wave w moves pieces 4w..4w+3 of a tile; M0 (the LDS
  // destination of piece 4w) is written once per tile and the instruction offset -- which the hardware adds to BOTH addresses
  // -- steps the LDS side by 1 KiB per piece (the per-piece source offsets are pre-compensated by - 1024 k).  The source is
  // a buffer descriptor over the cursor's PAGE (base = page + this head's column, num_records ends inside the page's last
  // row) plus the tile's byte offset in the scalar offset: rows past the end of the page fail the hardware range check
  // (voffset + soffset + inst_offset >= num_records) and arrive in LDS as ZEROS -- a page's ragged last tile needs no
  // clamped addresses (tools/probe_bufdma.hip: semantics verified on MI355X).  Nothing the compiler emits in this kernel
  // touches M0 (tests/test_isa_audit.py).  Piece 0's "s_mov m0 + s_nop 3" is also the 5 wait states gfx9 wants between a
  // scalar write of the offset / descriptor registers (the cursor advance, which hipcc may place right before this asm
  // without knowing what is inside it) and the VMEM instruction that reads them.
  template <int K> MMPL_DEV void dma_k() {
    if constexpr (ABL(1)) return;
    if constexpr (K == 0)
      asm volatile("s_mov_b32 m0, %3\n\ts_nop 3\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(dko[0]), "s"(ksrd), "s"(ksoff), "s"(kslot) : "memory");
    else
      asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%c3 lds" ::"v"(dko[K]), "s"(ksrd), "s"(ksoff), "i"(1024 * K) : "memory");
  }
  template <int K> MMPL_DEV void dma_v() {
    if constexpr (ABL(1)) return;
    if constexpr (K == 0)
      asm volatile("s_mov_b32 m0, %3\n\ts_nop 3\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(dvo[0]), "s"(vsrd), "s"(vsoff), "s"(vslot) : "memory");
    else
      asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%c3 lds" ::"v"(dvo[K]), "s"(vsrd), "s"(vsoff), "i"(1024 * K) : "memory");
  }
  MMPL_DEV static u32x4 page_srd(const bf16_t* page, int head_, int rows, int ld) {
    const uint64_t p = (uint64_t)(page + head_ * 128);
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)p);
    r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane((uint32_t)rows * (uint32_t)ld * 2u - ((uint32_t)ld * 2u - 256u));
    r[3] = 0x00020000u;
    return r;
  }
  // Cursor arithmetic.  Inside a run an iteration only adds kstep / vstep to the scalar offsets (no branch); plan(j), called
  // between runs, places both cursors for iteration j from scratch (K at tile j + 4, V at j + 3, each parked on the block's last
  // tile once it gets there -- that tile is simply re-fetched, so every event issues exactly 8 pieces and the counted waits
  // never change), sets or clears the mask tile for tile j, and returns how many iterations may run before the next event.
  // tile index `at` of the block's page list -> its page, the tile's index inside the page, the page's tile count (pages may
  // differ in length: merged runs of cache slots; <= 24 pages, scalar code that runs between runs only)
  MMPL_DEV void locate(int at, int& pg, int& pos, int& tiles_pg) const {
    pg = 0;
    int start = 0, tp = (rows_each[0] + KVB - 1) / KVB;
    while (at >= start + tp) {
      start += tp;
      ++pg;
      tp = (rows_each[pg] + KVB - 1) / KVB;
    }
    pos = at - start;
    tiles_pg = tp;
  }
  MMPL_DEV void seek_k(int t, int& pos, int& tiles_pg) {
    int pg;
    locate(t_first + t, pg, pos, tiles_pg);
    ksrd = page_srd(k_pages[pg], head, rows_each[pg], ldk);
    ksoff = (uint32_t)pos * tile_bytes_k;
  }
  MMPL_DEV void seek_v(int t, int& pos, int& tiles_pg) {
    int pg;
    locate(t_first + t, pg, pos, tiles_pg);
    vsrd = page_srd(v_pages[pg], head, rows_each[pg], ldv);
    vsoff = (uint32_t)pos * tile_bytes_v;
  }
  MMPL_DEV void set_mask(int valid) {
    // register r of half h holds kv row 32 h + 8 (r >> 2) + (r & 3) + 4 hi: one compare of 4 hi against a scalar per register
    // (written as asm so that the 32 compares do not all stay live in SGPR pairs at once)
    const int hi4 = 4 * hi;
    const float ninf = -INFINITY, zero = mbase;
    Ctx* self = this;
    sfor<32>([self, hi4, ninf, zero, valid](auto ii) {
      constexpr int i = decltype(ii)::value, h = i >> 4, r = i & 15;
      const int thr = valid - (32 * h + 8 * (r >> 2) + (r & 3));          // masked iff 4 hi >= thr
      asm volatile("v_cmp_le_i32 vcc, %2, %1\n\tv_cndmask_b32 %0, %4, %3, vcc" : "=v"(self->M[h][r]) : "v"(hi4), "s"(thr), "v"(ninf), "v"(zero) : "vcc");
    });
    masked = 1;
  }
  MMPL_DEV void clear_mask() {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) M[h][r] = mbase;
    masked = 0;
  }
  MMPL_DEV int plan(int j) {
    int n = T - j, pos, tp;
    const int tk = min(j + 4, T - 1), tv = min(j + 3, T - 1);
    seek_k(tk, pos, tp);
    kstep = 0;
    if (tk < T - 1) { kstep = tile_bytes_k; n = min(n, min(tp - pos, T - tk)); }
    seek_v(tv, pos, tp);
    vstep = 0;
    if (tv < T - 1) { vstep = tile_bytes_v; n = min(n, min(tp - pos, T - tv)); }
    int pg_j, pos_j, tp_j;
    locate(t_first + j, pg_j, pos_j, tp_j);
    const int valid = rows_each[pg_j] - (tp_j - 1) * KVB;                // rows of this page's last tile
    if (valid < KVB && pos_j == tp_j - 1) {
      set_mask(valid);
      n = 1;
    } else {
      if (masked) clear_mask();
      n = min(n, (valid < KVB ? tp_j - 1 : tp_j) - pos_j);               // a run ends with tile j's page (before its ragged last tile)
    }
    return __builtin_amdgcn_readfirstlane(n);
  }
  MMPL_DEV void advance_k() {
    kslot = (kslot + TILE) & (RING * TILE - 1);
    ksoff += kstep;
  }
  MMPL_DEV void advance_v() {
    vslot = RING * TILE + ((vslot + TILE) & (RING * TILE - 1));
    vsoff += vstep;
  }
  // Wait states between the score MFMAs (hand-issued: the compiler's hazard recogniser does not see them) and compiler-generated VALU
  // reads of S.  The S tiles are in-out operands of the pad: a read of S is then data-dependent on it and cannot be scheduled above it.
  // (Without that tie the `v_max` chain of score_max() was free to move in front of the nops -- it did, differently from build to
  // build: the sampled reference then came from accumulators still in flight.  Harmless for the result -- any reference inside the
  // window gives the exact softmax -- but it made the stateless kernel's BITS depend on the compiler's schedule: round 6, found by
  // hashing the outputs of two builds, profiles/r06z_attn_hash_r05_vs_r06.log.)
  MMPL_DEV void mfma_write_pad() {
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(S[0][0]), "+v"(S[0][1]), "+v"(S[1][0]), "+v"(S[1][1])::"memory");
  }
  MMPL_DEV void rotate() {                               // end of B(j): the next B phase reads the next ring slots
    rk = (rk + TILE) & (RING * TILE - 1);
    rv = (rv + TILE) & (RING * TILE - 1);
  }

  // ---------------------------------------------------------------- softmax streams (placement: attn_w64_sched.inc)
  // pair q of stream X: registers e, e+1 of S_X[h]; packed into word wd of P_X[ks]
  template <int MODE, int X, int Q, int EL> MMPL_DEV void sm_e() {
    if constexpr (ABL(2 | 32)) return;
    constexpr int h = Q >> 3, e = (Q & 7) * 2 + EL;
    if constexpr (MODE == 0)
      asm volatile("v_exp_f32 %0, %1" : "=v"(t[X][Q & 1][EL]) : "v"(S[X][h][e]));
    else
      asm volatile("v_sub_f32 %0, %1, %2\n\tv_exp_f32 %0, %0" : "=&v"(t[X][Q & 1][EL]) : "v"(S[X][h][e]), "v"(mref[X]));
  }
  template <int MODE, int X, int Q> MMPL_DEV void sm_e0() { sm_e<MODE, X, Q, 0>(); }
  template <int MODE, int X, int Q> MMPL_DEV void sm_e1() { sm_e<MODE, X, Q, 1>(); }
  template <int MODE, int X, int Q> MMPL_DEV void sm_a0() {
    if constexpr (ABL(2 | 64)) return;
    if constexpr (Q == 0 && MODE == 1) la[X] = t[X][0][0];
    else asm volatile("v_add_f32 %0, %0, %1" : "+v"(la[X]) : "v"(t[X][Q & 1][0]));
  }
  template <int MODE, int X, int Q> MMPL_DEV void sm_a1() {
    if constexpr (ABL(2 | 64)) return;
    if constexpr (Q == 0 && MODE == 1) lb[X] = t[X][0][1];
    else asm volatile("v_add_f32 %0, %0, %1" : "+v"(lb[X]) : "v"(t[X][Q & 1][1]));
  }
  template <int MODE, int X, int Q> MMPL_DEV void sm_c() {
    if constexpr (ABL(2 | 128)) return;
    constexpr int ks = Q >> 2, wd = Q & 3;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(P[X][ks][wd]) : "v"(t[X][Q & 1][0]), "v"(t[X][Q & 1][1]));
  }
  // GENERAL pass, slow path of one tile of stream X (see the header): returns the tile's partial row sum.
  template <int X> MMPL_DEV float slow(float lt) {
    // (the reads of S below must stay behind every statement of the schedule that precedes this call -- the MFMAs issued since S_X was
    // finished ARE its wait states: an empty statement that "modifies" S pins them there)
    asm volatile("" : "+v"(S[X][0]), "+v"(S[X][1]));
    float mx = S[X][0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, S[X][0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[X][1][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (first[X]) {
      first[X] = 0;
      mref[X] = mx;                                   // O = l = 0: nothing to rescale
    } else {
      const float m_new = fmaxf(mref[X], mx);
      const float alpha = __builtin_amdgcn_exp2f(mref[X] - m_new);
      mref[X] = m_new;
      l[X] *= alpha;
      sfor<64>([alpha](auto ii) {
        constexpr int i = decltype(ii)::value;
        float v;
        asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(v) : "i"(AO + 64 * X + i));
        v *= alpha;
        asm volatile("v_accvgpr_write_b32 a[%c0], %1" ::"i"(AO + 64 * X + i), "v"(v));
      });
    }
    lt = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int h = q >> 3, e = (q & 7) * 2;
      const float a0 = __builtin_amdgcn_exp2f(S[X][h][e] - mref[X]);
      const float a1 = __builtin_amdgcn_exp2f(S[X][h][e + 1] - mref[X]);
      lt += a0 + a1;
      P[X][q >> 2][q & 3] = pack2bf(a0, a1);
    }
    asm volatile("s_nop 1" ::: "memory");             // VALU-written P -> MFMA operand
    return lt;
  }
  MMPL_DEV float score_max() {                           // largest of this lane's 64 scores (both query blocks, both kv halves)
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(fmaxf(mx, fmaxf(S[0][0][r], S[0][1][r])), fmaxf(S[1][0][r], S[1][1][r]));
    return mx;
  }
  template <int MODE, int X> MMPL_DEV void finish() {
    if constexpr (MODE == 0 || ABL(2 | 4 | 32 | 64 | 128)) return;     // FAST: la / lb run on; timing ablations: no slow path
    float lt = la[X] + lb[X];
    if (__builtin_expect(first[X] || __any(!(lt <= BOUND_GEN)), 0)) lt = slow<X>(lt);
    l[X] += lt;
  }
};

#include "attn_w64_sched.inc"

// One pass over the block's KV tiles in softmax mode MODE (0 FAST, 1 GENERAL): O in a[0:127], row sums in la / lb (FAST) or l
// (GENERAL), references in mref.
template <int MODE> MMPL_DEV void w64_pass(Ctx& k) {
  const int T = k.T;
  sfor<128>([](auto ii) { asm volatile("v_accvgpr_write_b32 a[%c0], 0" ::"i"(AO + decltype(ii)::value)); });
  k.l[0] = k.l[1] = 0.f;
  k.la[0] = k.la[1] = k.lb[0] = k.lb[1] = 0.f;
  k.mref[0] = k.mref[1] = 0.f;
  k.mbase = 0.f;
  k.first[0] = k.first[1] = 1;
  k.clear_mask();
  k.kslot = k.wave_slot; k.vslot = RING * TILE + k.wave_slot;
  k.rk = TILE; k.rv = 0;
  k.kstep = k.vstep = 0;

  // ---- prologue: DMA events -4 .. -1 (event e = { K(e+4), V(e+3) }; V(-1) does not exist), then the K(0) fragments
  int pos, tpg;
  k.seek_k(0, pos, tpg);
  sfor<4>([&k](auto kk) { k.template dma_k<decltype(kk)::value>(); });
  k.advance_k();
#pragma unroll 1
  for (int e = 1; e < 4; ++e) {
    k.seek_k(min(e, T - 1), pos, tpg);
    k.seek_v(min(e - 1, T - 1), pos, tpg);
    sfor<4>([&k](auto kk) { k.template dma_k<decltype(kk)::value>(); });
    sfor<4>([&k](auto kk) { k.template dma_v<decltype(kk)::value>(); });
    k.advance_k();
    k.advance_v();
  }
  asm volatile("s_waitcnt vmcnt(24)\n\ts_barrier" ::: "memory");        // K(0): the first 4 of the 28 pieces
  k.kaddr = k.kbase;
  sfor<16>([&k](auto gi) { k.template lds_k<decltype(gi)::value>(); });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  k.plan(0);
  if constexpr (MODE == 0 && !ABL(~0)) {
    float ref;
    if (k.use_mem) {
      // FAST pass reference from the block's history: the mean log-sum-exp of this lane's two query rows on the previous launch of
      // this attention (kernel epilogue below) -- consecutive denoise steps see the same K / V and nearly the same q, so the row sums
      // come out near 2^0, in the middle of the window, wherever the scores' range is (no sampling: the pipeline starts at once)
      ref = k.mem_ref;
    } else {
    // FAST pass reference (header): the scores of KV tile 0 for both query blocks, their largest per lane (= two query rows) + offset
    // becomes m_ref; it enters every later score through the C operand of the tile's first MFMA, so the pipeline below is untouched.
    sfor<16>([&k](auto gi) { k.template mfma_qk<0, decltype(gi)::value>(); });
    sfor<16>([&k](auto gi) { k.template mfma_qk<1, decltype(gi)::value>(); });
    k.mfma_write_pad();
    float mx = k.score_max();
    if (T > FAST_REF_TILES) {
      // ... and of KV tiles 1 .. 3, which the prologue above already sent on their way into ring slots 1 .. 3: four times the
      // sample for ~0.2 % more MFMAs per block (all full tiles strictly before the block's last one; T is block-uniform)
      sfor<FAST_REF_TILES - 1>([&k, &mx](auto ei) {
        constexpr int e = decltype(ei)::value + 1;
        // in flight, oldest first: K0 | K1 V0 | K2 V1 | K3 V2 (4 pieces each): K(e) has landed when at most 24 - 8 e are outstanding
        asm volatile("s_waitcnt vmcnt(%c0)\n\ts_barrier" ::"i"(24 - 8 * e) : "memory");
        k.kaddr = k.kbase + e * TILE;
        sfor<16>([&k](auto gi) { k.template lds_k<decltype(gi)::value>(); });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        sfor<16>([&k](auto gi) { k.template mfma_qk<0, decltype(gi)::value>(); });
        sfor<16>([&k](auto gi) { k.template mfma_qk<1, decltype(gi)::value>(); });
        k.mfma_write_pad();
        mx = fmaxf(mx, k.score_max());
      });
      k.kaddr = k.kbase;                                 // the K(0) fragments again: the pipeline starts from them
      sfor<16>([&k](auto gi) { k.template lds_k<decltype(gi)::value>(); });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    mx = fminf(fmaxf(mx, -1e30f), 1e30f);              // (a NaN / inf score: the end-of-pass check fails and GENERAL takes over)
    ref = mx + FAST_REF_OFFSET;
    }
    k.mref[0] = k.mref[1] = ref;
    k.mbase = -ref;
    k.masked = 1;                                      // make plan() rewrite the C-operand tile (mask of tile 0 included) on the new base
    k.plan(0);
  }
  w64_phase_a<MODE, true, false, true, false>(k);
  w64_phase_b<MODE, true, false, true, false>(k);
  // runs of identical, branch-free iterations; whatever happens once per page is decided in between (Ctx::plan)
#pragma unroll 1
  for (int j = 1; j < T;) {
    const int n = k.plan(j);
    j += n;
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
      w64_phase_a<MODE, true, true, true, true>(k);
      w64_phase_b<MODE, true, true, true, true>(k);
    }
  }
  w64_phase_a<MODE, false, true, false, true>(k);
  w64_phase_b<MODE, false, true, false, true>(k);
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
}

template <bool SPLIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_w64_kernel(AttnArgs a, int local_base, int sp) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;

  const int n_qb = (a.Lq + QB - 1) / QB;
  int head, qb, part = 0, tail_idx = 0;
  {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    int local = idx;
    if (SPLIT) {
      part = idx % sp;
      tail_idx = idx / sp;
      local = local_base + tail_idx;
      tail_idx = xcd * (gridDim.x / (8 * sp)) + tail_idx;
    }
    if (!mmpl_attn_item(a.H, n_qb, xcd, local, head, qb)) return;     // grid padding (head counts that are no multiple of 8)
  }

  // the accumulator file is ours: this statement makes the kernel descriptor allocate all 256 entries
  asm volatile("s_nop 0" ::: ALL_AGPRS);

  // ---- Q fragments -> a[128:191], prescaled: lane (l31, hi) holds Q[row][16c + 8*hi .. +8] * scale * log2(e).  The DiT forward
  // folds that factor into q where q is produced (qknorm_kernel, before the rounding to bf16: q_prescaled); a raw q is scaled
  // here, at the price of a second rounding.
  const float c = a.scale * 1.4426950408889634f;
  const float qmul = a.q_prescaled ? 1.0f : c;
  {
    const bf16_t* qbase = a.q + head * 128 + 8 * hi;
#pragma unroll
    for (int X = 0; X < 2; ++X) {
      const int qrow = min(qb * QB + wave * 64 + 32 * X + l31, a.Lq - 1);
      const bf16_t* qp = qbase + (size_t)qrow * a.ldq;
      u32x4 qf[8];
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) {
        qf[cc] = *reinterpret_cast<const u32x4*>(qp + 16 * cc);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          qf[cc][j] = pack2bf(__uint_as_float(qf[cc][j] << 16) * qmul, __uint_as_float(qf[cc][j] & 0xffff0000u) * qmul);
      }
      if (X == 0)
        sfor<32>([&qf](auto ii) { constexpr int i = decltype(ii)::value; asm volatile("v_accvgpr_write_b32 a[%c0], %1" ::"i"(AQ + i), "v"(qf[i >> 2][i & 3])); });
      else
        sfor<32>([&qf](auto ii) { constexpr int i = decltype(ii)::value; asm volatile("v_accvgpr_write_b32 a[%c0], %1" ::"i"(AQ + 32 + i), "v"(qf[i >> 2][i & 3])); });
    }
  }

  int T_all = 0;
  for (int p = 0; p < a.n_pages; ++p) T_all += (a.page_rows_each[p] + KVB - 1) / KVB;
  const int t_first = SPLIT ? (int)((long long)part * T_all / sp) : 0;
  const int T = SPLIT ? (int)((long long)(part + 1) * T_all / sp) - t_first : T_all;     // tiles of THIS block

  Ctx k;
  k.hi = hi;
  k.k_pages = a.k_pages; k.v_pages = a.v_pages;
  k.ldk = a.ldk; k.ldv = a.ldv; k.rows_each = a.page_rows_each; k.head = head; k.T = T;
  k.drow = lane >> 4; k.dchunk = lane & 15;
  k.prow = 16 * wave + k.drow;                        // LDS row of piece 4w; piece 4w + k: + 4 k
  k.t_first = t_first;
  k.tile_bytes_k = (uint32_t)KVB * a.ldk * 2u;
  k.tile_bytes_v = (uint32_t)KVB * a.ldv * 2u;
  k.wave_slot = wave * 4096;
  // the bank swizzle is keyed on the LDS row: K chunk ^= row & 15, V chunk ^= (row & 3) << 2
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    k.dko[kk] = (uint32_t)((k.prow + 4 * kk) * a.ldk + ((k.dchunk ^ ((k.prow + 4 * kk) & 15)) << 3)) * 2u - 1024u * kk;
    k.dvo[kk] = (uint32_t)((k.prow + 4 * kk) * a.ldv + ((k.dchunk ^ (k.drow << 2)) << 3)) * 2u - 1024u * kk;
  }
  // per-lane fragment read offsets (swizzled): koff(cs) = kbase ^ 32 cs, voff(nb) = vbase ^ 64 nb
  k.kbase = l31 * 256 + 32 * ((l31 & 15) >> 1) + 16 * (hi ^ (l31 & 1));
  {
    const int i16 = lane & 15, g16 = (lane >> 4) & 1;
    k.vbase = RING * TILE + (4 * hi + (i16 >> 2)) * 256 + 64 * (i16 >> 2) + 32 * g16 + 8 * (i16 & 3);
  }

  // ---- the FAST pass; if any row of the block cannot be held by it (see the header), the GENERAL pass from scratch.
  // With a history (AttnArgs.history / history_mem: per (head, query block, split part) one state byte and one int16 per lane pair,
  // owned by the caller, carried from one launch of the same (layer, CFG branch, stage) to the next -- consecutive denoise steps see the
  // same K / V and nearly the same q) a block whose FAST pass failed does not keep paying for both passes:
  //   * from the first failure on every pass leaves, per lane, the mean log-sum-exp of its two query rows (log2 units, rounded to an
  //     integer), and the next FAST pass takes THAT as its reference instead of sampling the first four KV tiles: the row sums then
  //     come out near 2^0 whatever the range of the scores (1.0 instead of 2.66 FAST-pass times);
  //   * a block whose FAST pass fails even so (two rows of a lane further apart than the window is wide) goes straight to GENERAL on the
  //     following launches (1.66): FAST is tried again on the 8th, after another failure on the 16th, then every 31st launch.
  // State byte: bit 7 = the lane references are valid; bits 5-6 = back-off level; bits 0-4 = countdown (> 1: GENERAL, decrement;
  // 1: FAST is tried again; 0: FAST).  0 = no history: the stateless kernel, bit for bit (blocks that never fail never leave 0).
  // Either pass is the exact softmax up to rounding, so any contents give a correct result; WHICH pass ran, and against which reference,
  // decides the rounding: with a history the bits depend on the launches before (identical sequences of launches give identical bits).
  extern __shared__ __attribute__((aligned(16))) char w64_smem[];
  volatile int* redo = reinterpret_cast<volatile int*>(w64_smem + 2 * RING * TILE);
  const size_t hidx = (((size_t)head * n_qb + qb) << 2) | (SPLIT ? part : 0);
  unsigned char* hist = a.history ? a.history + hidx : nullptr;
  short* hmem = a.history ? a.history_mem + hidx * 128 + wave * 32 + l31 : nullptr;
  int hstate = 0;
  if (hist) hstate = __builtin_amdgcn_readfirstlane((int)*reinterpret_cast<volatile unsigned char*>(hist));
  const bool try_fast = (hstate & 31) <= 1;
  const bool have_mem = (hstate & 128) != 0;
  k.use_mem = have_mem;
  k.mem_ref = 0.f;
  if (have_mem && try_fast) k.mem_ref = (float)*reinterpret_cast<volatile short*>(hmem);
  int redo_block = 1;
  bool bad = false;
  if (try_fast) {
    if (tid == 0) *redo = 0;                             // ordered before the vote by the pass's barriers
    w64_pass<0>(k);
#pragma unroll
    for (int X = 0; X < 2; ++X) {
      k.l[X] = k.la[X] + k.lb[X];
      const float l_tot = k.l[X] + __shfl_xor(k.l[X], 32, 64);
      bad |= !(l_tot >= FAST_L_MIN && l_tot <= FAST_L_MAX);
    }
    if constexpr (ABL(~0)) bad = false;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");        // the parked cursors' last re-fetches
    if (__any(bad) && lane == 0) *redo = 1;
    __syncthreads();
    redo_block = *redo;
  }
  if (a.redo_stats) {
    if (tid == 0) {
      atomicAdd(a.redo_stats, 1ull);
      if (!try_fast) atomicAdd(a.redo_stats + 3, 1ull);                    // straight to GENERAL
      else if (redo_block) atomicAdd(a.redo_stats + 1, 1ull);              // paid for both passes
      else if (have_mem) atomicAdd(a.redo_stats + 4, 1ull);                // FAST held on the remembered references
    }
    if (__any(bad) && lane == 0) atomicAdd(a.redo_stats + 2, 1ull);        // waves (64 query rows) that held a failing row themselves
  }
  if (redo_block) w64_pass<1>(k);
  // The history is rewritten only HERE: every wave of the block has passed a barrier since it read the state byte (the vote's, or the
  // GENERAL pass's own), so no late wave can see the new value and take the other branch (a countdown going 2 -> 1 flips the decision).
  if (hist) {
    int ns;
    if (!try_fast) ns = hstate - 1;                                        // (keeps bit 7 and the level)
    else if (!redo_block) ns = hstate & 128;
    else if (!have_mem) ns = 128;                                          // first failure: the references below, FAST again next time
    else {
      const int level = (hstate & 31) ? min(((hstate >> 5) & 3) + 1, 2) : 0;
      ns = 128 | (level << 5) | min(8 << level, 31);
    }
    if (ns & 128) {
      // mean log-sum-exp of the lane's two query rows relative to nothing: m_ref + log2(l) (both passes keep l relative to m_ref)
      float lse = 0.f;
#pragma unroll
      for (int X = 0; X < 2; ++X) {
        const float l_tot = k.l[X] + __shfl_xor(k.l[X], 32, 64);
        lse += 0.5f * (k.mref[X] + __builtin_amdgcn_logf(l_tot));          // v_log_f32 = log2
      }
      lse = fminf(fmaxf(lse, -32000.f), 32000.f);                          // (a NaN -- only ever from non-finite scores -- becomes -32000)
      if (hi == 0) *hmem = (short)__builtin_rintf(lse);
    }
    if (tid == 0) *hist = (unsigned char)ns;
  }

  // ---- epilogue: lane (q = l31, hi) holds O_X[q][32*nb + 8*g + 4*hi + {0..3}] in a[64 X + 16 nb + 4 g ..+3]
#pragma unroll
  for (int X = 0; X < 2; ++X) {
    const float l_tot = k.l[X] + __shfl_xor(k.l[X], 32, 64);
    float o[64];
    if (X == 0) sfor<64>([&o](auto ii) { constexpr int i = decltype(ii)::value; asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(o[i]) : "i"(AO + i)); });
    else sfor<64>([&o](auto ii) { constexpr int i = decltype(ii)::value; asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(o[i]) : "i"(AO + 64 + i)); });
    const int rr = wave * 64 + 32 * X + l31;
    if (SPLIT) {
      // partial of this KV range: O (fp32, relative to m_ref), then m (raw score units, as attn_merge_kernel expects), l per
      // row -- [tail block][part][256 rows][128 + 2]
      float* pbase = a.split_ws + ((size_t)tail_idx * sp + part) * (QB * 130);
      float* op = pbase + (size_t)rr * 128 + 4 * hi;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(op + 32 * nb + 8 * g) = f32x4{o[16 * nb + 4 * g], o[16 * nb + 4 * g + 1], o[16 * nb + 4 * g + 2], o[16 * nb + 4 * g + 3]};
      if (hi == 0) {
        pbase[QB * 128 + rr] = k.mref[X] / c;
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

int mmpl_attention_w64_smem() { return W64_SMEM; }
const void* mmpl_attention_w64_symbol(int split) {
  return split ? reinterpret_cast<const void*>(attn_w64_kernel<true>) : reinterpret_cast<const void*>(attn_w64_kernel<false>);
}
void mmpl_launch_attention_w64(const AttnArgs& a, int blocks, int local_base, int sp, bool split, hipStream_t s) {
  if (split) hipLaunchKernelGGL(attn_w64_kernel<true>, dim3(blocks), dim3(256), W64_SMEM, s, a, local_base, sp);
  else hipLaunchKernelGGL(attn_w64_kernel<false>, dim3(blocks), dim3(256), W64_SMEM, s, a, local_base, sp);
}
