// Flash-attention forward over a frame-slot page table (no mask), head_dim 128, bf16 in / fp32 softmax.
//
// Replaces the reference's "fancy-index gather of K and V + flash_attn_varlen" (causal_fps_model.py:219-227,
// attention.py:139-185): K/V are read IN PLACE from the per-layer KV cache through a table of page (frame
// slot) base pointers, so no gather copy exists.  The same kernel serves text cross-attention (one page of
// 512 rows) and the generic attention() seam.
//
// Structure (CDNA4): 512 threads = 8 waves, each wave owns 32 query rows (Q fragments live in registers);
// KV tiles of 64 rows, page-aligned (a frame's ragged tail tile is masked).  K/V tiles are staged
// global -> registers -> LDS (row-padded: conflict-free ds_read_b128 for K, ds_read_b64_tr_b16 for V), the
// next tile's global loads are in flight under the current tile's MFMAs, one barrier per tile.
// "Swapped" formulation: S^T = K.Q^T and O^T = V^T.P^T with v_mfma_f32_32x32x16_bf16, so each lane owns one
// query column: softmax statistics are lane-local (one cross-half shuffle per tile) and the P^T fragment for
// the PV MFMA is just 8 consecutive accumulator registers packed to bf16 (no LDS round trip for P).
#include <stdlib.h>

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


}  // namespace

hipError_t mmpl_launch_attention(const AttnArgs& a, hipStream_t s) {
  if (a.Lq <= 0) return hipSuccess;
  if (a.n_pages <= 0 || a.n_pages > MMPL_MAX_PAGES || a.page_rows <= 0 || (a.ldq % 8) || (a.ldk % 8) || (a.ldv % 8) ||
      (a.ldo % 4))
    return hipErrorInvalidValue;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<0>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int n_qb = (a.Lq + QB - 1) / QB;
  if (a.cross) hipLaunchKernelGGL(attn_fwd_kernel<1>, dim3(n_qb * a.H), dim3(512), SMEM, s, a);
  else hipLaunchKernelGGL(attn_fwd_kernel<0>, dim3(n_qb * a.H), dim3(512), SMEM, s, a);
  return hipGetLastError();
}
