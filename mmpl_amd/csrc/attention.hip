// Flash-attention forward over a frame-slot page table (no mask), head_dim 128, bf16 in / fp32 softmax.
//
// Replaces the reference's "fancy-index gather of K and V + flash_attn_varlen" (causal_fps_model.py:219-227,
// attention.py:139-185): K/V are read IN PLACE from the per-layer KV cache through a table of page (frame
// slot) base pointers, so no gather copy exists.
//
// Three kernels share the math, the page table and the fragment layouts; mmpl_launch_attention (end of this file) picks:
//   attn_w64_kernel (attn_w64.hip)  the DiT forward's self-attention: one wave per SIMD, 64 query rows per wave (default)
//   attn_cross_kernel (below)       text cross-attention over <= 128 keys (the padded tail collapsed): K / V resident in LDS over a
//                                   run of query blocks, the next block's q in flight -- the launch is a q + o stream
//   attn_fwd_kernel (below)         the lock-step original: longer contexts, the image stream, the CLIP tower, raw-q callers of the
//                                   attention() seam, MMPL_ATTN_V1=1
// (round 1's ping-pong kernel, attn_pp_kernel -- 8 waves, two groups half an iteration apart, 1010-1060 TFLOP/s -- was removed in
// round 3: attn_w64_kernel superseded it on the hot path and nothing else needed its speed; DESIGN.md section 3.1 keeps its record)
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
#include "mmpl_config.h"

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

// One KV tile of 64 rows (kb / vb: its K and V in LDS) against the wave's 32 query rows (qf): S^T = K.Q^T, online softmax,
// O^T += V^T.P^T.  `valid` = rows of the tile that exist; last_bias: see the page's-last-row comment below.
struct AttnState { f32x16 o[4]; float m_run, l_run; };
MMPL_DEV void attn_tile(const char* kb, const char* vb, const bf16x8 (&qf)[8], int valid, float last_bias, float c, int k_off, int v_off,
                        int hi, AttnState& st) {
#pragma clang fp contract(off)          // the one fma (the exponent's argument) is written out: the same bits whatever this is inlined into
  f32x16 (&o)[4] = st.o;
  float& m_run = st.m_run;
  float& l_run = st.l_run;
  // ---- S^T = K . Q^T, one 32-row kv half at a time so that the vector work on half 0 (mask, row max) issues in the
  // shadow of half 1's MFMAs; likewise P.V of half 0 runs under the exponentials of half 1.
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
  // the page's last row stands for `last_row_copies` identical keys (the zero-padded tail of the text context, api.hip):
  // copies * exp(scale * s) = exp(scale * (s + ln(copies) / scale))
  if (last_bias != 0.f && valid <= KVB) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kv = 8 * (r >> 2) + 4 * hi + (r & 3);
      if (kv == valid - 1) s0[r] += last_bias;
      if (kv + 32 == valid - 1) s1[r] += last_bias;
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
  for (int r = 0; r < 16; ++r) { s0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], c, -mc)); ls0 += s0[r]; }
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
  for (int r = 0; r < 16; ++r) { s1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], c, -mc)); ls1 += s1[r]; }
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

  AttnState st;
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) st.o[nb][r] = 0.f;
  st.m_run = -INFINITY;
  st.l_run = 0.f;
  const float c = a.scale * 1.4426950408889634f;  // fold log2(e): p = exp2(s*c - m*c)
  const float last_bias = a.last_row_copies > 1 ? __logf((float)a.last_row_copies) / a.scale : 0.f;

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

    const int pg = t / tiles_pp, row0 = (t - pg * tiles_pp) * KVB;
    attn_tile(kb, vb, qf, a.page_rows - row0, last_bias, c, k_off, v_off, hi, st);

    if (t + 1 < total) stage_write(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane (q = l31, hi) holds O[q][32*nb + 8*g + 4*hi + {0..3}] in o[nb][4g..4g+3].  Straight from there a store
  // instruction would write 16 B per query row (32 rows x 2 lanes x 8 B): 16 partial-sector writes per row.  The wave's 32 x 128
  // tile goes through its own 8 KiB of the (now free: the tile loop ended on a barrier) K/V buffers instead and leaves as full
  // 256-byte rows, 16 B per lane.  Row pitch 264 B: the 8-byte writes of lanes 0..31 fall on 64 different banks.
  const float l_tot = st.l_run + __shfl_xor(st.l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  constexpr int O_PITCH = 264;
  static_assert(NW * QW * O_PITCH <= SMEM, "output staging fits the K/V buffers");
  char* ob = smem + wave * (QW * O_PITCH);
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint2 w;
      w.x = pack2bf(st.o[nb][4 * g] * inv, st.o[nb][4 * g + 1] * inv);
      w.y = pack2bf(st.o[nb][4 * g + 2] * inv, st.o[nb][4 * g + 3] * inv);
      *reinterpret_cast<uint2*>(ob + l31 * O_PITCH + (32 * nb + 8 * g + 4 * hi) * 2) = w;
    }
  const int orow = lane >> 4, ochunk = lane & 15;                 // 4 rows x 16 chunks of 16 B per instruction
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int r = orow + 4 * k, q_out = qb * QB + wave * QW + r;
    const uint2 lo = *reinterpret_cast<const uint2*>(ob + r * O_PITCH + ochunk * 16);
    const uint2 hi2 = *reinterpret_cast<const uint2*>(ob + r * O_PITCH + ochunk * 16 + 8);
    if (q_out < a.Lq) *reinterpret_cast<uint4*>(a.o + (size_t)q_out * a.ldo + head * 128 + ochunk * 8) = uint4{lo.x, lo.y, hi2.x, hi2.y};
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// attn_cross_kernel: the same math for a context of <= 2 KV tiles (text cross-attention after the padded-tail collapse, a short
// prompt; 128 keys at most).  There the lock-step kernel is one block per CU (233 registers x 8 waves) doing load q -> stage K/V
// -> 32-64 MFMAs -> store o strictly one after the other, ~15 rounds of it per launch: nothing overlaps, 2.0 TB/s of q + o.
// Here a block keeps ITS head's K / V tiles in LDS for a whole run of query blocks and every wave has the next block's q
// fragments in flight while it works on the current one (the prefetch of the row passes, elementwise.hip); o leaves through a
// separate staging area, so the loop has no barrier at all.  Bit-identical to attn_fwd_kernel (same attn_tile, same order).
constexpr int O_PITCH_X = 264;
constexpr int SMEM_X = 2 * BUF + NW * QW * O_PITCH_X;            // 143360 B: one block per CU, as the registers dictate anyway

template <int NT>
__global__ __launch_bounds__(512, 2) void attn_cross_kernel(AttnArgs a, int qb_per_block, int blocks_per_head) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int n_qb = (a.Lq + QB - 1) / QB;
  int head, part;                                                // a head's blocks on one XCD, as attn_fwd_kernel
  if ((a.H & 7) == 0) {
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    head = xcd + 8 * (local / blocks_per_head);
    part = local % blocks_per_head;
  } else {
    head = blockIdx.x / blocks_per_head;
    part = blockIdx.x % blocks_per_head;
  }
  int qb = part * qb_per_block;
  const int qb_end = min(qb + qb_per_block, n_qb);
  if (qb >= qb_end) return;

  auto load_q = [&](bf16x8 (&f)[8], int qblock) {
    const int qrow = min(qblock * QB + wave * QW + l31, a.Lq - 1);
    const bf16_t* qp = a.q + (size_t)qrow * a.ldq + head * 128 + 8 * hi;
#pragma unroll
    for (int c = 0; c < 8; ++c) f[c] = *reinterpret_cast<const bf16x8*>(qp + 16 * c);
  };
  bf16x8 qf[8], qn[8];
  load_q(qf, qb);
  {  // the head's K / V, once
    const int srow = tid >> 4, schunk = tid & 15;
    const bf16_t* kp = a.k_pages[0] + head * 128 + schunk * 8;
    const bf16_t* vp = a.v_pages[0] + head * 128 + schunk * 8;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = min(t * KVB + srow + 32 * j, a.page_rows - 1);
        const u32x4 rk = *reinterpret_cast<const u32x4*>(kp + (size_t)r * a.ldk);
        const u32x4 rv = *reinterpret_cast<const u32x4*>(vp + (size_t)r * a.ldv);
        *reinterpret_cast<u32x4*>(smem + t * BUF + (srow + 32 * j) * K_STRIDE + schunk * 16) = rk;
        *reinterpret_cast<u32x4*>(smem + t * BUF + K_TILE + (srow + 32 * j) * V_STRIDE + schunk * 16) = rv;
      }
  }
  __syncthreads();
  const float c = a.scale * 1.4426950408889634f;
  const float last_bias = a.last_row_copies > 1 ? __logf((float)a.last_row_copies) / a.scale : 0.f;
  const int k_off = l31 * K_STRIDE + 16 * hi;
  const int i16 = lane & 15, g16 = (lane >> 4) & 1;
  const int v_off = (4 * hi + (i16 >> 2)) * V_STRIDE + (16 * g16 + 4 * (i16 & 3)) * 2;
  char* ob = smem + 2 * BUF + wave * (QW * O_PITCH_X);
  const int orow = lane >> 4, ochunk = lane & 15;

  for (;; ++qb) {
    const bool more = qb + 1 < qb_end;                           // block-uniform
    if (more) load_q(qn, qb + 1);
    asm volatile("" ::: "memory");                               // the prefetch is issued here, not where it is consumed
    AttnState st;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) st.o[nb][r] = 0.f;
    st.m_run = -INFINITY;
    st.l_run = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
      attn_tile(smem + t * BUF, smem + t * BUF + K_TILE, qf, a.page_rows - t * KVB, last_bias, c, k_off, v_off, hi, st);
    const float l_tot = st.l_run + __shfl_xor(st.l_run, 32, 64);
    const float inv = 1.0f / l_tot;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 w;
        w.x = pack2bf(st.o[nb][4 * g] * inv, st.o[nb][4 * g + 1] * inv);
        w.y = pack2bf(st.o[nb][4 * g + 2] * inv, st.o[nb][4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(ob + l31 * O_PITCH_X + (32 * nb + 8 * g + 4 * hi) * 2) = w;
      }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int r = orow + 4 * k, q_out = qb * QB + wave * QW + r;
      const uint2 lo = *reinterpret_cast<const uint2*>(ob + r * O_PITCH_X + ochunk * 16);
      const uint2 hi2 = *reinterpret_cast<const uint2*>(ob + r * O_PITCH_X + ochunk * 16 + 8);
      if (q_out < a.Lq) *reinterpret_cast<uint4*>(a.o + (size_t)q_out * a.ldo + head * 128 + ochunk * 8) = uint4{lo.x, lo.y, hi2.x, hi2.y};
    }
    if (!more) break;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      qf[i] = qn[i];
      asm volatile("" : "+v"(qf[i]));                            // "used" here: the wait for the prefetch lands before the next one is issued
    }
  }
}


// Combine the `sp` KV-range partials of each tail query block: O = sum_p w_p O_p / sum_p w_p l_p, w_p = 2^((m_p - m) c).
__global__ __launch_bounds__(256) void attn_merge_kernel(AttnArgs a, int local_base, int sp, int tb) {
  __shared__ float wgt[QB][4];
  const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
  const int n_qb = (a.Lq + QB - 1) / QB;
  int head, qb;
  if (!mmpl_attn_item(a.H, n_qb, xcd, local_base + k, head, qb)) return;
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

// MMPL_ATTN_V1=1: A/B runs of a whole forward on the lock-step kernel (mmpl_config.h)
int mmpl_attention_self_variant() { return mmpl_config().attn_v1 ? ATTN_LOCKSTEP : ATTN_W64; }

namespace {
// Pages that lie back to back in memory (the KV-cache slots of one layer are one allocation; so are a stage's scratch pages) are
// presented to attn_w64_kernel as fewer, longer pages: every page ends in a ragged 64-row tile (3600 = 56 * 64 + 16, 1560 =
// 24 * 64 + 24) whose MFMAs mostly multiply masked rows -- 1.3 % of the tiles at 720p, 2.5 % at 480p.  The kernel takes a row
// count per page (AttnArgs.page_rows_each), so every maximal run becomes ONE page (s0: 2 slots -> 1 page; s1: 4 + 5 -> 2;
// s2: 13 -> 1; s3: 15 + 6 -> 2).  Softmax does not care about the order of the keys; only fp32 summation order changes.
AttnArgs merge_contiguous_pages(const AttnArgs& a) {
  int order[MMPL_MAX_PAGES];
  for (int i = 0; i < a.n_pages; ++i) order[i] = i;
  auto before = [&a](int x, int y) {                      // (allocation group, K address): see AttnArgs.page_group
    return a.page_group[x] != a.page_group[y] ? a.page_group[x] < a.page_group[y] : a.k_pages[x] < a.k_pages[y];
  };
  for (int i = 1; i < a.n_pages; ++i)                      // insertion sort
    for (int j = i; j > 0 && before(order[j], order[j - 1]); --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
  const size_t kstride = (size_t)a.page_rows * a.ldk, vstride = (size_t)a.page_rows * a.ldv;
  // (the kernel addresses a page with 32-bit byte offsets / a buffer descriptor: a merged page stays below 2 GiB)
  const size_t per_page = 2 * (kstride > vstride ? kstride : vstride);
  const int max_run = per_page ? (int)(((1ull << 31) - 1) / per_page) : 1;
  AttnArgs m = a;
  m.n_pages = 0;
  for (int i = 0; i < a.n_pages;) {
    int r = 1;
    while (i + r < a.n_pages && r < max_run && a.page_group[order[i + r]] == a.page_group[order[i]] &&
           a.k_pages[order[i + r]] == a.k_pages[order[i + r - 1]] + kstride && a.v_pages[order[i + r]] == a.v_pages[order[i + r - 1]] + vstride)
      ++r;
    m.k_pages[m.n_pages] = a.k_pages[order[i]];
    m.v_pages[m.n_pages] = a.v_pages[order[i]];
    m.page_rows_each[m.n_pages++] = r * a.page_rows;
    i += r;
  }
  return m;
}
}  // namespace

hipError_t mmpl_launch_attention(const AttnArgs& a_in, hipStream_t s) {
  if (a_in.Lq <= 0) return hipSuccess;
  if (a_in.n_pages <= 0 || a_in.n_pages > MMPL_MAX_PAGES) return hipErrorInvalidValue;
  const bool no_merge = mmpl_config().attn_no_merge;
  const bool w64_bound = a_in.variant == ATTN_W64 || (a_in.variant == ATTN_AUTO && !a_in.cross && a_in.q_prescaled && mmpl_attention_self_variant() == ATTN_W64);
  AttnArgs a = a_in;
  for (int p = 0; p < a.n_pages; ++p) a.page_rows_each[p] = a.page_rows;
  if (!no_merge && w64_bound && a.n_pages > 1) a = merge_contiguous_pages(a);
  if (a.n_pages <= 0 || a.n_pages > MMPL_MAX_PAGES || a.page_rows <= 0 || (a.ldq % 8) || (a.ldk % 8) || (a.ldv % 8) ||
      (a.ldo % 8) || ((uintptr_t)a.o & 15) || (a.variant != ATTN_AUTO && a.variant != ATTN_LOCKSTEP && a.variant != ATTN_W64))
    return hipErrorInvalidValue;
  // Kernel choice.  The DiT forward's self-attention (q prescaled by its producer) -> the 64-rows-per-wave kernel
  // (attn_w64.hip); the 8-tile text cross-attention -> the lock-step kernel (its prologue is the shortest); a raw-q launch
  // (the attention() seam) -> the lock-step kernel too (exact online softmax on the caller's q; attn_w64 would round q a second
  // time, kernels.h).  AttnArgs.variant (C ABI: mmpl_attn_fwd_variant) selects one explicitly.
  const bool no_split = mmpl_config().attn_nosplit;
  int variant = a.variant;
  if (variant == ATTN_AUTO) {
    variant = a.cross ? ATTN_LOCKSTEP : mmpl_attention_self_variant();
    if (variant == ATTN_W64 && !a.q_prescaled) variant = ATTN_LOCKSTEP;
  }
  if (a.q_prescaled && variant != ATTN_W64) return hipErrorInvalidValue;
  if (a.last_row_copies > 1 && (variant != ATTN_LOCKSTEP || a.n_pages != 1)) return hipErrorInvalidValue;
  const int n_qb = (a.Lq + QB - 1) / QB;
  if (variant == ATTN_LOCKSTEP && a.cross && a.n_pages == 1 && a.page_rows <= 2 * KVB) {
    // <= 2 KV tiles: the head's K / V stay in LDS over a run of query blocks; every CU gets one block (the registers allow no more),
    // the blocks of a head share its query blocks evenly
    const int nt = (a.page_rows + KVB - 1) / KVB;
    const void* f = nt == 1 ? reinterpret_cast<const void*>(attn_cross_kernel<1>) : reinterpret_cast<const void*>(attn_cross_kernel<2>);
    if (hipError_t e = mmpl_dyn_smem_once(f, SMEM_X); e != hipSuccess) return e;
    const int cus = 8 * mmpl_cus_per_xcd();
    int bph = cus / a.H < 1 ? 1 : cus / a.H;                    // blocks per head (40 heads, 256 CUs: 6)
    if (bph > n_qb) bph = n_qb;
    const int qpb = (n_qb + bph - 1) / bph;
    bph = (n_qb + qpb - 1) / qpb;
    if (nt == 1) hipLaunchKernelGGL(attn_cross_kernel<1>, dim3(bph * a.H), dim3(512), SMEM_X, s, a, qpb, bph);
    else hipLaunchKernelGGL(attn_cross_kernel<2>, dim3(bph * a.H), dim3(512), SMEM_X, s, a, qpb, bph);
    return hipGetLastError();
  }
  if (variant == ATTN_LOCKSTEP) {
    const void* f = a.cross ? reinterpret_cast<const void*>(attn_fwd_kernel<1>) : reinterpret_cast<const void*>(attn_fwd_kernel<0>);
    if (hipError_t e = mmpl_dyn_smem_once(f, SMEM); e != hipSuccess) return e;
    if (a.cross) hipLaunchKernelGGL(attn_fwd_kernel<1>, dim3(n_qb * a.H), dim3(512), SMEM, s, a);
    else hipLaunchKernelGGL(attn_fwd_kernel<0>, dim3(n_qb * a.H), dim3(512), SMEM, s, a);
    return hipGetLastError();
  }
  if (a.history) a.history_mem = reinterpret_cast<short*>(a.history + mmpl_attention_history_state_bytes(a.Lq, a.H));
  if (hipError_t e = mmpl_dyn_smem_once(mmpl_attention_w64_symbol(0), mmpl_attention_w64_smem()); e != hipSuccess) return e;
  if (hipError_t e = mmpl_dyn_smem_once(mmpl_attention_w64_symbol(1), mmpl_attention_w64_smem()); e != hipSuccess) return e;
  auto run = [&](int blocks, int local_base, int sp, bool split) { mmpl_launch_attention_w64(a, blocks, local_base, sp, split, s); };
  // Tail round: with one block per CU and b = n_qb*H/8 query blocks per XCD (32 CUs), the last b mod 32 blocks of every
  // XCD would run alone for a whole block time.  They are launched instead as `sp` blocks each over 1/sp of the KV tiles
  // (fp32 partials in split_ws) followed by a small merge kernel, so the tail round lasts ~1/sp block times.
  const int per_xcd = mmpl_cus_per_xcd();
  int tiles = 0;
  for (int p = 0; p < a.n_pages; ++p) tiles += (a.page_rows_each[p] + KVB - 1) / KVB;
  // b = work items per XCD (kernels.h: mmpl_attn_item), tb = those of the partial last round.  Only a tail that fits ONE round of
  // parts is split (tb * sp <= CUs per XCD): tails of 3 shorter rounds (tb = 17 -> 3 parts, 23-24 -> 4) and launches with fewer
  // items than CUs measured 0.6-8 % SLOWER than leaving them alone (profiles/r03Q_*).
  const int total = n_qb * a.H;
  const int b = (a.H & 7) == 0 ? total / 8 : (total + 7) / 8, tb = b % per_xcd;
  int sp = 1;
  if (!no_split && a.split_ws && b > per_xcd && tb > 0 && per_xcd / tb >= 2) {
    sp = per_xcd / tb > 4 ? 4 : per_xcd / tb;
    if (tiles / sp < 8 || (size_t)8 * tb * sp * QB * 130 * sizeof(float) > a.split_ws_bytes) sp = 1;
  }
  if (sp == 1) {
    run(8 * b, 0, 1, false);
  } else {
    run(8 * (b - tb), 0, 1, false);
    run(8 * tb * sp, b - tb, sp, true);
    hipLaunchKernelGGL(attn_merge_kernel, dim3(8 * tb), dim3(256), 0, s, a, b - tb, sp, tb);
  }
  return hipGetLastError();
}
