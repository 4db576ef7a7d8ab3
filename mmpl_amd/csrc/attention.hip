// Flash-attention forward over a frame-slot page table (no mask), head_dim 128, bf16 in / fp32 softmax.
//
// Replaces the reference's "fancy-index gather of K and V + flash_attn_varlen" (causal_fps_model.py:219-227,
// attention.py:139-185): K/V are read IN PLACE from the per-layer KV cache through a table of page (frame
// slot) base pointers, so no gather copy exists.
//
// Three kernels share the math, the page table and the fragment layouts; mmpl_launch_attention (end of this file) picks:
//   attn_w64_kernel (attn_w64.hip)  the DiT forward's self-attention: one wave per SIMD, 64 query rows per wave (default)
//   attn_pp_kernel  (below)         round 1's default; today a raw-q caller of the attention() seam, and MMPL_ATTN_PP=1
//   attn_fwd_kernel (below)         the lock-step original: text / image cross-attention, the CLIP tower, MMPL_ATTN_V1=1
//
// attn_fwd_kernel:
// 512 threads = 8 waves, each wave owns 32 query rows (Q fragments live in registers);
// KV tiles of 64 rows, page-aligned (a frame's ragged tail tile is masked).  K/V tiles are staged
// global -> registers -> LDS (row-padded: conflict-free ds_read_b128 for K, ds_read_b64_tr_b16 for V), the
// next tile's global loads are in flight under the current tile's MFMAs, one barrier per tile.
// "Swapped" formulation: S^T = K.Q^T and O^T = V^T.P^T with v_mfma_f32_32x32x16_bf16, so each lane owns one
// query column: softmax statistics are lane-local (one cross-half shuffle per tile) and the P^T fragment for
// the PV MFMA is just 8 consecutive accumulator registers packed to bf16 (no LDS round trip for P).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int NW = 8, QW = 32, QB = NW * QW, KVB = 64;
constexpr int K_STRIDE = 272;               // bytes per K row in LDS (256 + 16 pad)
constexpr int V_STRIDE = 320;               // bytes per V row in LDS (256 + 64 pad): 4 rows -> 4 bank windows
constexpr int K_TILE = KVB * K_STRIDE;      // 17408
constexpr int V_TILE = KVB * V_STRIDE;      // 20480
constexpr int BUF = K_TILE + V_TILE;        // 37888
constexpr int SMEM = 2 * BUF;               // 75776

typedef __attribute__((ext_vector_type(4))) short s16x4;

MMPL_DEV bf16x8 tr_pair(const char* p0, const char* p1) {
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p1));
  return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// CROSS only tags the symbol (attn_fwd_kernel<0> = self-attention over cache pages, <1> = text cross-attention) so that
// profiles report the two launch populations separately; the code is identical.
template <int CROSS>
__global__ __launch_bounds__(512, 2) void attn_fwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;

  // block -> (head, q block): keep a head's blocks on one XCD so its K/V stream is shared in that L2
  const int n_qb = (a.Lq + QB - 1) / QB;
  int head, qb;
  if ((a.H & 7) == 0) {
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    head = xcd + 8 * (local / n_qb);
    qb = local % n_qb;
  } else {
    head = blockIdx.x / n_qb;
    qb = blockIdx.x % n_qb;
  }

  // ---- Q fragments (B operand of S^T = K.Q^T): lane holds Q[q = l31][16c + 8*hi .. +8]
  const int qrow = min(qb * QB + wave * QW + l31, a.Lq - 1);
  bf16x8 qf[8];
  {
    const bf16_t* qp = a.q + (size_t)qrow * a.ldq + head * 128 + 8 * hi;
#pragma unroll
    for (int c = 0; c < 8; ++c) qf[c] = *reinterpret_cast<const bf16x8*>(qp + 16 * c);
  }

  const int tiles_pp = (a.page_rows + KVB - 1) / KVB;
  const int total = a.n_pages * tiles_pp;

  // ---- staging roles: thread moves 2x16 B of K and 2x16 B of V per tile
  const int srow = tid >> 4, schunk = tid & 15;
  u32x4 rk[2], rv[2];
  auto stage_load = [&](int t) {
    const int p = t / tiles_pp, row0 = (t - p * tiles_pp) * KVB;
    const bf16_t* kp = a.k_pages[p] + head * 128 + schunk * 8;
    const bf16_t* vp = a.v_pages[p] + head * 128 + schunk * 8;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = min(row0 + srow + 32 * j, a.page_rows - 1);
      rk[j] = *reinterpret_cast<const u32x4*>(kp + (size_t)r * a.ldk);
      rv[j] = *reinterpret_cast<const u32x4*>(vp + (size_t)r * a.ldv);
    }
  };
  auto stage_write = [&](int buf) {
    char* kb = smem + buf * BUF;
    char* vb = kb + K_TILE;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      *reinterpret_cast<u32x4*>(kb + (srow + 32 * j) * K_STRIDE + schunk * 16) = rk[j];
      *reinterpret_cast<u32x4*>(vb + (srow + 32 * j) * V_STRIDE + schunk * 16) = rv[j];
    }
  };

  f32x16 o[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[nb][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float c = a.scale * 1.4426950408889634f;  // fold log2(e): p = exp2(s*c - m*c)

  // per-lane LDS read offsets
  const int k_off = l31 * K_STRIDE + 16 * hi;                                          // + 32*c bytes, + 32 rows for half 1
  const int i16 = lane & 15, g16 = (lane >> 4) & 1;
  const int v_off = (4 * hi + (i16 >> 2)) * V_STRIDE + (16 * g16 + 4 * (i16 & 3)) * 2;  // + kv/d block offsets

  stage_load(0);
  stage_write(0);
  __syncthreads();

  for (int t = 0; t < total; ++t) {
    const int cur = t & 1;
    if (t + 1 < total) stage_load(t + 1);
    const char* kb = smem + cur * BUF;
    const char* vb = kb + K_TILE;

    // ---- S^T = K . Q^T, one 32-row kv half at a time so that the vector work on half 0 (mask, row max) issues in the
    // shadow of half 1's MFMAs; likewise P.V of half 0 runs under the exponentials of half 1.
    const int pg = t / tiles_pp, row0 = (t - pg * tiles_pp) * KVB;
    const int valid = a.page_rows - row0;
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
    // K fragments are read 4 ahead of the MFMA that consumes them (a rotating 4-deep register window): without this
    // hipcc issues read -> wait -> MFMA one at a time and every MFMA eats a full LDS round trip
    {
      auto kfrag = [&](int i) {  // i = 0..15: half = i >> 3, hd chunk = i & 7
        return *reinterpret_cast<const bf16x8*>(kb + k_off + (i >> 3) * (32 * K_STRIDE) + 32 * (i & 7));
      };
      bf16x8 w0 = kfrag(0), w1 = kfrag(1), w2 = kfrag(2), w3 = kfrag(3);
      __builtin_amdgcn_sched_barrier(0);   // hipcc otherwise sinks each read down to its MFMA
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 16; i += 4) {
        if (i < 8) s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, qf[i & 7], s0, 0, 0, 0);
        else s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, qf[i & 7], s1, 0, 0, 0);
        if (i + 4 < 16) w0 = kfrag(i + 4);
        __builtin_amdgcn_sched_barrier(0);
        if (i < 8) s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, qf[(i + 1) & 7], s0, 0, 0, 0);
        else s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, qf[(i + 1) & 7], s1, 0, 0, 0);
        if (i + 5 < 16) w1 = kfrag(i + 5);
        __builtin_amdgcn_sched_barrier(0);
        if (i < 8) s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, qf[(i + 2) & 7], s0, 0, 0, 0);
        else s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, qf[(i + 2) & 7], s1, 0, 0, 0);
        if (i + 6 < 16) w2 = kfrag(i + 6);
        __builtin_amdgcn_sched_barrier(0);
        if (i < 8) s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3, qf[(i + 3) & 7], s0, 0, 0, 0);
        else s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3, qf[(i + 3) & 7], s1, 0, 0, 0);
        if (i + 7 < 16) w3 = kfrag(i + 7);
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_setprio(0);
    }
    // lane holds kv_local = 32*half + 8*(r>>2) + 4*hi + (r&3) for its query column
    if (valid < KVB) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kv = 8 * (r >> 2) + 4 * hi + (r & 3);
        if (kv >= valid) s0[r] = -INFINITY;
        if (kv + 32 >= valid) s1[r] = -INFINITY;
      }
    }
    float mx = s0[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s0[r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s1[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    // exact lazy rescale: only when some row's running max actually grew (rare after the first tiles)
    if (__any(m_new > m_run)) {
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      l_run *= alpha;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[nb][r] *= alpha;
      m_run = m_new;
    }
    const float mc = m_run * c;
    float ls0 = 0.f, ls1 = 0.f;
    bf16x8 pb[2][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = __builtin_amdgcn_exp2f(s0[r] * c - mc); ls0 += s0[r]; }
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      union { uint32_t u[4]; bf16x8 v; } x0;
#pragma unroll
      for (int j = 0; j < 4; ++j) x0.u[j] = pack2bf(s0[8 * cc + 2 * j], s0[8 * cc + 2 * j + 1]);
      pb[0][cc] = x0.v;
    }
    // ---- O^T += V^T . P^T, kv half 0 (A operand element j of lane (d, hi) is V[32*half + 16*cc + 8*(j>>2) + 4*hi + (j&3)][d])
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const char* vp = vb + v_off + (16 * cc) * V_STRIDE + 64 * nb;
        const bf16x8 vf = tr_pair(vp, vp + 8 * V_STRIDE);
        o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[0][cc], o[nb], 0, 0, 0);
      }
#pragma unroll
    for (int r = 0; r < 16; ++r) { s1[r] = __builtin_amdgcn_exp2f(s1[r] * c - mc); ls1 += s1[r]; }
    l_run += ls0 + ls1;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      union { uint32_t u[4]; bf16x8 v; } x1;
#pragma unroll
      for (int j = 0; j < 4; ++j) x1.u[j] = pack2bf(s1[8 * cc + 2 * j], s1[8 * cc + 2 * j + 1]);
      pb[1][cc] = x1.v;
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const char* vp = vb + v_off + (32 + 16 * cc) * V_STRIDE + 64 * nb;
        const bf16x8 vf = tr_pair(vp, vp + 8 * V_STRIDE);
        o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[1][cc], o[nb], 0, 0, 0);
      }
    __builtin_amdgcn_s_setprio(0);

    if (t + 1 < total) stage_write(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane (q = l31, hi) holds O[q][32*nb + 8*g + 4*hi + {0..3}] in o[nb][4g..4g+3]
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int q_out = qb * QB + wave * QW + l31;
  if (q_out < a.Lq) {
    bf16_t* op = a.o + (size_t)q_out * a.ldo + head * 128 + 4 * hi;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 w;
        w.x = pack2bf(o[nb][4 * g] * inv, o[nb][4 * g + 1] * inv);
        w.y = pack2bf(o[nb][4 * g + 2] * inv, o[nb][4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(op + 32 * nb + 8 * g) = w;
      }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// v2 "ping-pong" (round 1's default for self-attention): same math and fragment layouts as attn_fwd_kernel, but the 8 waves of
// the (single) resident block are split into two groups that run half an iteration apart.  Per KV tile j a wave runs
//   M_j = { O += V(j-1).P(j-1) ; S(j) = K(j).Q }   32 MFMAs, LDS fragments requested 6 ahead, 4 LDS-DMA ops
//   V_j = { online softmax of S(j) -> P(j) }        ~200 VALU (32 exp2)
// and while the waves of one group are in M the co-resident waves of the other group (same SIMDs: waves w and w+4)
// are in V, so every SIMD has one wave feeding the matrix pipe and one on the VALU instead of all eight contending
// for the same pipe in lock-step.  One raw s_barrier per segment boundary keeps the groups in opposite phase.
// K/V tiles go global -> LDS by DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write) into rings of 4 (K) and
// 5 (V) tiles, issued three tiles ahead and ordered with counted s_waitcnt vmcnt; the LDS image is lane-linear, so
// the bank-conflict swizzle is applied to the per-lane SOURCE chunk (K: chunk ^= row & 15; V: chunk ^= (row & 3) << 2)
// and again on the fragment reads (SQ_LDS_BANK_CONFLICT = 0).
//
// Measured on MI355X (profiles/r01f_*, same box, 14B/720p shapes): +6 % (s1) ... +14 % (s3) over attn_fwd_kernel.
// Cycle counters inside the kernel (s_memtime per segment) say where the rest goes: M = 1650-2000 cycles for 1024
// cycles of MFMA (each LDS-DMA op stalls its in-order wave ~110 cycles at issue: the LDS-DMA path delivers ~17 B/clk/CU
// and this kernel asks for 70 % of that; with the DMA ops removed M = 1200, with the fragment reads removed too 1034),
// V = 1250-1400.  Tried and rejected on this structure: {QK+softmax | PV} split (-1 %), exponentials interleaved
// under the PV MFMAs (-10 %), DMA ops issued from the vector segment (-8 %) or staggered by wave (-9 %), register
// staging instead of DMA (-4 %, 250 VGPRs), static / per-group priorities (-2..-5 %).
constexpr int PP_RK = 4, PP_RV = 5, PP_TILE = KVB * 256;
constexpr int PP_SMEM = (PP_RK + PP_RV) * PP_TILE;      // 147456

// LDS-DMA issued from inline asm: hipcc cannot disambiguate LDS-DMA writes from later LDS reads and would put an
// s_waitcnt vmcnt(0) in front of the next ds_read (draining the prefetch it was meant to overlap); ordering is done
// by the explicit counted waits below instead.  (M0 = wave-uniform LDS destination; 1 wait state after the M0 write.
// M0 is a reserved register that hipcc only ever sets immediately before an instruction of its own that needs it, and
// this kernel has no such instruction, so it is not listed as a clobber.)
MMPL_DEV void glds16(const void* base, uint32_t voff, char* lds) {   // base: wave-uniform; voff: per-lane byte offset
  const uint32_t dst = (uint32_t)(size_t)((__attribute__((address_space(3))) char*)lds);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(dst) : "memory");
}

// SPLIT (the tail round, see mmpl_launch_attention): the block handles KV tiles [part*T/sp, (part+1)*T/sp) of query block
// `local_base + idx / sp` and writes an un-normalised fp32 partial (O, m, l) for attn_merge_kernel.
template <bool SPLIT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_pp_kernel(AttnArgs a, int local_base, int sp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;            // waves w and w+4 share a SIMD (dispatch order cycles over the 4 SIMDs)
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
    // any other head count (Wan 1.3B: 12): the hardware deals blocks round-robin to the 8 XCDs, so XCD x is given the x-th
    // contiguous chunk of the head-major (head, query block) list -- the blocks sharing one L2 work on at most a few heads;
    // the grid is padded to 8 chunks (launcher), the padding blocks leave at once
    const int total = n_qb * a.H, per = (total + 7) >> 3;
    const int item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (item >= total) return;
    head = item / n_qb;
    qb = item % n_qb;
  }

  const int qrow = min(qb * QB + wave * QW + l31, a.Lq - 1);
  bf16x8 qf[8];
  {
    const bf16_t* qp = a.q + (size_t)qrow * a.ldq + head * 128 + 8 * hi;
#pragma unroll
    for (int c = 0; c < 8; ++c) qf[c] = *reinterpret_cast<const bf16x8*>(qp + 16 * c);
    // complete the Q loads HERE in the compiler's view too: otherwise it waits for them lazily inside the main loop
    // with vmcnt(7..0), which also drains the LDS-DMA prefetch (invisible to its counters) every iteration
#pragma unroll
    for (int c = 0; c < 8; ++c) asm volatile("" ::"v"(qf[c]));
  }

  const int tiles_pp = (a.page_rows + KVB - 1) / KVB;
  const int T_all = a.n_pages * tiles_pp;
  const int t_first = SPLIT ? (int)((long long)part * T_all / sp) : 0;
  const int T = SPLIT ? (int)((long long)(part + 1) * T_all / sp) - t_first : T_all;     // tiles of THIS block
  char* const kring = smem;
  char* const vring = smem + PP_RK * PP_TILE;

  // ---- DMA roles: wave w moves pieces w and w + 8 (4 rows x 256 B each) of the K tile and of the V tile.  A DMA op is
  // global_load_lds_dwordx4 voff, s[base]: the per-lane byte offsets (row * ld + swizzled chunk) are computed once,
  // the wave-uniform base (page, first row of the tile, head) per tile; only a page's ragged last tile re-clamps rows.
  const int drow = lane >> 4, dchunk = lane & 15;
  uint32_t dko[2], dvo[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = 4 * (wave + 8 * h) + drow;
    dko[h] = (uint32_t)(row * a.ldk + ((dchunk ^ (row & 15)) << 3)) * 2u;
    dvo[h] = (uint32_t)(row * a.ldv + ((dchunk ^ ((row & 3) << 2)) << 3)) * 2u;
  }
  int ipg = t_first / tiles_pp, irow0 = (t_first % tiles_pp) * KVB, it = 0;          // next tile to issue
  // wave-uniform source pointers of that tile (first row, this head).  They are advanced / re-read from the page table
  // in issue_advance() and PINNED there: a scalar load first used inside the matrix segment would put an
  // s_waitcnt lgkmcnt(0) in front of the DMA op, draining the wave's whole LDS fragment prefetch window each time
  const bf16_t* kcur = a.k_pages[ipg] + (size_t)irow0 * a.ldk + head * 128;
  const bf16_t* vcur = a.v_pages[ipg] + (size_t)irow0 * a.ldv + head * 128;
  asm volatile("" : "+s"(kcur), "+s"(vcur));
  // piece k of the tile being issued: 0 = K rows 0..31 share, 1 = V, 2 = K rows 32..63 share, 3 = V
  auto issue_piece = [&](int k) {
    const int h = k >> 1;
    const bool isv = k & 1;
    const int ld = isv ? a.ldv : a.ldk;
    char* dst = (isv ? vring + (it % PP_RV) * PP_TILE : kring + (it % PP_RK) * PP_TILE) + (wave + 8 * h) * 1024;
    uint32_t off = isv ? dvo[h] : dko[h];
    if (irow0 + KVB > a.page_rows) {             // ragged tail: clamp to the page's last row (masked in the softmax)
      const int row = 4 * (wave + 8 * h) + drow, r = min(row, a.page_rows - 1 - irow0);
      off = (uint32_t)(r * ld + ((dchunk ^ (isv ? ((row & 3) << 2) : (row & 15))) << 3)) * 2u;
    }
    glds16(isv ? vcur : kcur, off, dst);
  };
  auto issue_advance = [&]() {
    ++it;
    irow0 += KVB;
    kcur += (size_t)KVB * a.ldk;
    vcur += (size_t)KVB * a.ldv;
    if (irow0 >= a.page_rows) {
      irow0 = 0;
      ++ipg;
      if (ipg < a.n_pages) {
        kcur = a.k_pages[ipg] + head * 128;
        vcur = a.v_pages[ipg] + head * 128;
      }
    }
    asm volatile("" : "+s"(kcur), "+s"(vcur));
  };
  auto issue_tile = [&]() {
#pragma unroll
    for (int k = 0; k < 4; ++k) issue_piece(k);
    issue_advance();
  };

  f32x16 o[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[nb][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float c = a.scale * 1.4426950408889634f;

  // ---- per-lane fragment read offsets (swizzled)
  int koff[8];
#pragma unroll
  for (int cs = 0; cs < 8; ++cs) koff[cs] = l31 * 256 + 32 * (cs ^ ((l31 & 15) >> 1)) + 16 * (hi ^ (l31 & 1));
  const int i16 = lane & 15, g16 = (lane >> 4) & 1;
  int voff[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) voff[nb] = (4 * hi + (i16 >> 2)) * 256 + 64 * (nb ^ (i16 >> 2)) + 32 * g16 + 8 * (i16 & 3);

  // ---- prologue: tiles 0..2 in flight, tile 0 landed
  issue_tile();
  if (T > 1) issue_tile();
  if (T > 2) issue_tile();
  if (T > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (T > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();      // group B runs half an iteration late

  f32x16 s0, s1;
  bf16x8 pb[2][2];
  int crow0 = (t_first % tiles_pp) * KVB;           // first kv row (within its page) of compute tile j
  constexpr int PD = 6;                             // LDS fragments requested ahead of the MFMA that consumes them

  // ---- M_j: 16 PV MFMAs of tile j-1, then 16 QK^T MFMAs of tile j (two accumulator chains, alternating).  The LDS
  // fragment of step i+PD is requested right after the MFMA of step i (a rotating register window pinned with
  // sched_barrier: left alone hipcc reads one fragment ahead and every MFMA eats an LDS round trip).  The four DMA ops
  // of tile j+3 go out one per quarter of the segment so the address path drains in between.
  auto m_segment = [&](auto do_pv, auto do_qk, int j) {
    constexpr bool PV = decltype(do_pv)::value, QK = decltype(do_qk)::value;
    constexpr int I0 = PV ? 0 : 16, I1 = QK ? 32 : 16;
    constexpr int SH = PV && QK ? 3 : 2, MASK = (1 << SH) - 1;
    const bool do_issue = j + 3 < T;
    const char* kb = kring + (j % PP_RK) * PP_TILE;
    const char* vb = vring + ((j + PP_RV - 1) % PP_RV) * PP_TILE;      // tile j-1
    auto frag = [&](int i) -> bf16x8 {
      if (i < 16) {                               // V^T fragment: i = 8*h + 4*cc + nb
        const char* vp = vb + voff[i & 3] + (32 * (i >> 3) + 16 * ((i >> 2) & 1)) * 256;
        return tr_pair(vp, vp + 8 * 256);
      }
      const int q = i - 16;                       // K fragment: kv half = q & 1, hd chunk = q >> 1
      return *reinterpret_cast<const bf16x8*>(kb + koff[q >> 1] + (q & 1) * (32 * 256));
    };
    f32x16 zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
    bf16x8 w[PD];
#pragma unroll
    for (int i = 0; i < PD; ++i) w[i] = frag(I0 + i);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = I0; i < I1; ++i) {
      const bf16x8 f = w[(i - I0) % PD];
      if (i < 16) o[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, pb[i >> 3][(i >> 2) & 1], o[i & 3], 0, 0, 0);
      else if ((i & 1) == 0) s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, qf[(i - 16) >> 1], i == 16 ? zero : s0, 0, 0, 0);
      else s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, qf[(i - 16) >> 1], i == 17 ? zero : s1, 0, 0, 0);
      if (i + PD < I1) w[(i - I0) % PD] = frag(i + PD);
      if (((i - I0) & MASK) == 1 && do_issue) issue_piece((i - I0) >> SH);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (do_issue) issue_advance();
    __builtin_amdgcn_s_setprio(0);
  };
  // ---- V_j: online softmax of S(j) -> P(j) (bf16 B-fragments in pb)
  auto v_segment = [&]() {
    // S(j) is consumed HERE, P(j) is complete at the end of this segment: without the two pins hipcc moves the vector
    // work across the (memory-only) barriers into the neighbouring matrix segments, where it serialises with the MFMAs
    asm volatile("" : "+v"(s0), "+v"(s1));
    const int valid = a.page_rows - crow0;
    if (valid < KVB) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kv = 8 * (r >> 2) + 4 * hi + (r & 3);
        if (kv >= valid) s0[r] = -INFINITY;
        if (kv + 32 >= valid) s1[r] = -INFINITY;
      }
    }
    crow0 += KVB;
    if (crow0 >= a.page_rows) crow0 = 0;
    float mx = s0[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s0[r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s1[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    // exact lazy rescale: only when some row's running max actually grew (rare after the first tiles)
    if (__any(m_new > m_run)) {
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      l_run *= alpha;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[nb][r] *= alpha;
      m_run = m_new;
    }
    const float mc = m_run * c;
    float ls0 = 0.f, ls1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = __builtin_amdgcn_exp2f(s0[r] * c - mc); ls0 += s0[r]; }
#pragma unroll
    for (int r = 0; r < 16; ++r) { s1[r] = __builtin_amdgcn_exp2f(s1[r] * c - mc); ls1 += s1[r]; }
    l_run += ls0 + ls1;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      union { uint32_t u[4]; bf16x8 v; } x0, x1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        x0.u[q] = pack2bf(s0[8 * cc + 2 * q], s0[8 * cc + 2 * q + 1]);
        x1.u[q] = pack2bf(s1[8 * cc + 2 * q], s1[8 * cc + 2 * q + 1]);
      }
      pb[0][cc] = x0.v;
      pb[1][cc] = x1.v;
    }
    asm volatile("" : "+v"(pb[0][0]), "+v"(pb[0][1]), "+v"(pb[1][0]), "+v"(pb[1][1]), "+v"(l_run));
  };
  // Before the barrier that precedes the first reader of tile j+1 every wave's four DMA ops of it must have landed;
  // tile j+2 and (if already issued: group A waits after its M_j, group B one segment earlier) tile j+3 stay in flight.
  auto dma_wait = [&](int j, bool issued_j3) {
    const int younger = (j + 2 < T ? 4 : 0) + (issued_j3 && j + 3 < T ? 4 : 0);
    if (younger == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  using std::true_type;
  using std::false_type;

  m_segment(false_type{}, true_type{}, 0);
  if (grp == 1) dma_wait(0, true);
  __builtin_amdgcn_s_barrier();
  v_segment();
  if (grp == 0) dma_wait(0, true);
  __builtin_amdgcn_s_barrier();
  for (int j = 1; j < T; ++j) {
    m_segment(true_type{}, true_type{}, j);
    if (grp == 1) dma_wait(j, true);
    __builtin_amdgcn_s_barrier();
    v_segment();
    if (grp == 0) dma_wait(j, true);
    __builtin_amdgcn_s_barrier();
  }
  m_segment(true_type{}, false_type{}, T);
  if (grp == 0) __builtin_amdgcn_s_barrier();

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (SPLIT) {
    // partial of this KV range: O (fp32, relative to m_run), then m, l per row -- [tail block][part][256 rows][128 + 2]
    float* pbase = a.split_ws + ((size_t)tail_idx * sp + part) * (QB * 130);
    const int rr = wave * QW + l31;
    float* op = pbase + (size_t)rr * 128 + 4 * hi;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(op + 32 * nb + 8 * g) = f32x4{o[nb][4 * g], o[nb][4 * g + 1], o[nb][4 * g + 2], o[nb][4 * g + 3]};
    if (hi == 0) {
      pbase[QB * 128 + rr] = m_run;
      pbase[QB * 129 + rr] = l_tot;
    }
    return;
  }
  const float inv = 1.0f / l_tot;
  const int q_out = qb * QB + wave * QW + l31;
  if (q_out < a.Lq) {
    bf16_t* op = a.o + (size_t)q_out * a.ldo + head * 128 + 4 * hi;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 w;
        w.x = pack2bf(o[nb][4 * g] * inv, o[nb][4 * g + 1] * inv);
        w.y = pack2bf(o[nb][4 * g + 2] * inv, o[nb][4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(op + 32 * nb + 8 * g) = w;
      }
  }
}

// Combine the `sp` KV-range partials of each tail query block: O = sum_p w_p O_p / sum_p w_p l_p, w_p = 2^((m_p - m) c).
__global__ __launch_bounds__(256) void attn_merge_kernel(AttnArgs a, int local_base, int sp, int tb) {
  __shared__ float wgt[QB][4];
  const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
  const int n_qb = (a.Lq + QB - 1) / QB;
  const int local = local_base + k, head = xcd + 8 * (local / n_qb), qb = local % n_qb;
  const float* base = a.split_ws + (size_t)(xcd * tb + k) * sp * (QB * 130);
  const float c = a.scale * 1.4426950408889634f;
  {
    const int r = threadIdx.x;
    float m = -INFINITY;
    for (int p = 0; p < sp; ++p) m = fmaxf(m, base[(size_t)p * (QB * 130) + QB * 128 + r]);
    float l = 0.f, w[4] = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < sp; ++p) {
      w[p] = __builtin_amdgcn_exp2f((base[(size_t)p * (QB * 130) + QB * 128 + r] - m) * c);
      l += w[p] * base[(size_t)p * (QB * 130) + QB * 129 + r];
    }
    const float inv = 1.0f / l;
    for (int p = 0; p < 4; ++p) wgt[r][p] = w[p] * inv;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < QB * 32; i += 256) {          // 16-byte chunks, coalesced
    const int r = i >> 5, ch = i & 31, q_out = qb * QB + r;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < sp; ++p) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)p * (QB * 130) + (size_t)r * 128 + ch * 4);
      const float w = wgt[r][p];
      acc[0] += w * v[0]; acc[1] += w * v[1]; acc[2] += w * v[2]; acc[3] += w * v[3];
    }
    if (q_out < a.Lq) {
      uint2 o2;
      o2.x = pack2bf(acc[0], acc[1]);
      o2.y = pack2bf(acc[2], acc[3]);
      *reinterpret_cast<uint2*>(a.o + (size_t)q_out * a.ldo + head * 128 + ch * 4) = o2;
    }
  }
}

}  // namespace

size_t mmpl_attention_split_ws_bytes() { return (size_t)256 * QB * 130 * sizeof(float); }   // <= one block per CU in the tail round

namespace {
bool env_flag(const char* name) { const char* v = getenv(name); return v && atoi(v); }
}  // namespace

// MMPL_ATTN_V1=1 / MMPL_ATTN_PP=1: A/B runs of a whole forward on the lock-step / ping-pong kernel (read once per process)
int mmpl_attention_self_variant() {
  static const int v = env_flag("MMPL_ATTN_V1") ? ATTN_LOCKSTEP : env_flag("MMPL_ATTN_PP") ? ATTN_PINGPONG : ATTN_W64;
  return v;
}

namespace {
// Pages that lie back to back in memory (the KV-cache slots of one layer are one allocation; so are a stage's scratch pages) are
// presented to the kernel as fewer, longer pages: every page ends in a ragged 64-row tile (3600 = 56 * 64 + 16, 1560 = 24 * 64 + 24)
// whose MFMAs mostly multiply masked rows -- 1.3 % of the tiles at 720p, 2.5 % at 480p.  The kernel takes ONE page length, so
// the maximal runs are cut into pages of gcd(run lengths) x page_rows (s0: 2 slots -> 1 page; s2: 13 -> 1; s3: 15 + 6 -> 7
// pages of 3; s1's 4 + 5 stay 9).  Softmax does not care about the order of the keys; only fp32 summation order changes.
AttnArgs merge_contiguous_pages(const AttnArgs& a) {
  int order[MMPL_MAX_PAGES];
  for (int i = 0; i < a.n_pages; ++i) order[i] = i;
  for (int i = 1; i < a.n_pages; ++i)                      // insertion sort by K address
    for (int j = i; j > 0 && a.k_pages[order[j]] < a.k_pages[order[j - 1]]; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
  const size_t kstride = (size_t)a.page_rows * a.ldk, vstride = (size_t)a.page_rows * a.ldv;
  int run_start[MMPL_MAX_PAGES], run_len[MMPL_MAX_PAGES], n_runs = 0, g = 0;
  for (int i = 0; i < a.n_pages;) {
    int r = 1;
    while (i + r < a.n_pages && a.k_pages[order[i + r]] == a.k_pages[order[i + r - 1]] + kstride &&
           a.v_pages[order[i + r]] == a.v_pages[order[i + r - 1]] + vstride)
      ++r;
    run_start[n_runs] = i; run_len[n_runs++] = r;
    int x = g, y = r;
    while (y) { const int t = x % y; x = y; y = t; }
    g = x;
    i += r;
  }
  // (the kernels address a page with 32-bit byte offsets / buffer descriptors)
  while (g > 1 && ((size_t)g * kstride * 2 >= (1ull << 31) || (size_t)g * vstride * 2 >= (1ull << 31))) {
    int d = 2;
    while (g % d) ++d;
    g /= d;
  }
  if (g <= 1) return a;
  AttnArgs m = a;
  m.page_rows = a.page_rows * g;
  m.n_pages = 0;
  for (int r = 0; r < n_runs; ++r)
    for (int j = 0; j < run_len[r]; j += g) {
      m.k_pages[m.n_pages] = a.k_pages[order[run_start[r] + j]];
      m.v_pages[m.n_pages++] = a.v_pages[order[run_start[r] + j]];
    }
  return m;
}
}  // namespace

hipError_t mmpl_launch_attention(const AttnArgs& a_in, hipStream_t s) {
  if (a_in.Lq <= 0) return hipSuccess;
  if (a_in.n_pages <= 0 || a_in.n_pages > MMPL_MAX_PAGES) return hipErrorInvalidValue;
  static const bool no_merge = env_flag("MMPL_ATTN_NO_MERGE");
  const bool w64_bound = a_in.variant == ATTN_W64 || (a_in.variant == ATTN_AUTO && !a_in.cross && a_in.q_prescaled && mmpl_attention_self_variant() == ATTN_W64);
  const AttnArgs a = (!no_merge && w64_bound && a_in.n_pages > 1) ? merge_contiguous_pages(a_in) : a_in;
  if (a.n_pages <= 0 || a.n_pages > MMPL_MAX_PAGES || a.page_rows <= 0 || (a.ldq % 8) || (a.ldk % 8) || (a.ldv % 8) ||
      (a.ldo % 4) || a.variant < ATTN_AUTO || a.variant > ATTN_W64)
    return hipErrorInvalidValue;
  // Kernel choice.  The DiT forward's self-attention (q prescaled by its producer) -> the 64-rows-per-wave kernel
  // (attn_w64.hip); the 8-tile text cross-attention -> the lock-step kernel (its prologue is the shortest); a raw-q launch
  // (the attention() seam) -> the ping-pong kernel (see mmpl_attention_self_variant).  AttnArgs.variant (C ABI:
  // mmpl_attn_fwd_variant) selects one explicitly.
  static const bool no_split = env_flag("MMPL_ATTN_NOSPLIT");
  int variant = a.variant;
  if (variant == ATTN_AUTO) {
    variant = a.cross ? ATTN_LOCKSTEP : mmpl_attention_self_variant();
    if (variant == ATTN_W64 && !a.q_prescaled) variant = ATTN_PINGPONG;
  }
  if (a.q_prescaled && variant != ATTN_W64) return hipErrorInvalidValue;
  const int n_qb = (a.Lq + QB - 1) / QB;
  if (variant == ATTN_LOCKSTEP) {
    const void* f = a.cross ? reinterpret_cast<const void*>(attn_fwd_kernel<1>) : reinterpret_cast<const void*>(attn_fwd_kernel<0>);
    if (hipError_t e = mmpl_dyn_smem_once(f, SMEM); e != hipSuccess) return e;
    if (a.cross) hipLaunchKernelGGL(attn_fwd_kernel<1>, dim3(n_qb * a.H), dim3(512), SMEM, s, a);
    else hipLaunchKernelGGL(attn_fwd_kernel<0>, dim3(n_qb * a.H), dim3(512), SMEM, s, a);
    return hipGetLastError();
  }
  const bool w64 = variant == ATTN_W64;
  if (hipError_t e = w64 ? mmpl_dyn_smem_once(mmpl_attention_w64_symbol(0), mmpl_attention_w64_smem())
                         : mmpl_dyn_smem_once(reinterpret_cast<const void*>(attn_pp_kernel<false>), PP_SMEM);
      e != hipSuccess)
    return e;
  if (hipError_t e = w64 ? mmpl_dyn_smem_once(mmpl_attention_w64_symbol(1), mmpl_attention_w64_smem())
                         : mmpl_dyn_smem_once(reinterpret_cast<const void*>(attn_pp_kernel<true>), PP_SMEM);
      e != hipSuccess)
    return e;
  auto run = [&](int blocks, int local_base, int sp, bool split) {
    if (w64) mmpl_launch_attention_w64(a, blocks, local_base, sp, split, s);
    else if (split) hipLaunchKernelGGL(attn_pp_kernel<true>, dim3(blocks), dim3(512), PP_SMEM, s, a, local_base, sp);
    else hipLaunchKernelGGL(attn_pp_kernel<false>, dim3(blocks), dim3(512), PP_SMEM, s, a, local_base, sp);
  };
  // Tail round: with one block per CU and b = n_qb*H/8 query blocks per XCD (32 CUs), the last b mod 32 blocks of every
  // XCD would run alone for a whole block time.  They are launched instead as `sp` blocks each over 1/sp of the KV tiles
  // (fp32 partials in split_ws) followed by a small merge kernel, so the tail round lasts ~1/sp block times.
  const int per_xcd = mmpl_cus_per_xcd();
  const int tiles = a.n_pages * ((a.page_rows + KVB - 1) / KVB);
  int sp = 1, tb = 0, b = 0;
  if (!no_split && a.split_ws && (a.H & 7) == 0) {
    b = n_qb * (a.H / 8);
    tb = b % per_xcd;
    if (b > per_xcd && tb > 0 && per_xcd / tb >= 2) sp = per_xcd / tb > 4 ? 4 : per_xcd / tb;
    if (tiles / sp < 8 || (size_t)8 * tb * sp * QB * 130 * sizeof(float) > a.split_ws_bytes) sp = 1;
  }
  if (sp == 1) {
    run((a.H & 7) == 0 ? n_qb * a.H : 8 * ((n_qb * a.H + 7) / 8), 0, 1, false);        // other head counts: grid padded to 8 XCD chunks
  } else {
    run(8 * (b - tb), 0, 1, false);
    run(8 * tb * sp, b - tb, sp, true);
    hipLaunchKernelGGL(attn_merge_kernel, dim3(8 * tb), dim3(256), 0, s, a, b - tb, sp, tb);
  }
  return hipGetLastError();
}
