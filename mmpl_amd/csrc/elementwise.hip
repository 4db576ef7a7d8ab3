// HBM-bound kernels of the Wan DiT forward: LayerNorm+modulate, full-dim QK RMSNorm + 3-axis RoPE + KV page
// write, patchify / unpatchify, timestep sinusoid, modulation prep, CFG + UniPC step.
// All of them move 16 B per lane, a wave works on one token row at a time (the two big ones walk a range of rows with the next
// row's loads in flight), fp32 statistics, and round to bf16 exactly where the reference's bf16 module boundaries round
// (DESIGN.md "numerics"); every fp32 expression whose rounding matters is written with explicit fmaf under contract(off).
#include "common.h"
#include "kernels.h"
#include "mmpl_config.h"

namespace {

MMPL_DEV void unpack8(const uint4& u, float (&f)[8]) {
  f[0] = bf2f(u.x & 0xffff); f[1] = bf2f(u.x >> 16); f[2] = bf2f(u.y & 0xffff); f[3] = bf2f(u.y >> 16);
  f[4] = bf2f(u.z & 0xffff); f[5] = bf2f(u.z >> 16); f[6] = bf2f(u.w & 0xffff); f[7] = bf2f(u.w >> 16);
}
MMPL_DEV uint4 pack8(const float (&f)[8]) {
  uint4 u;
  u.x = pack2bf(f[0], f[1]); u.y = pack2bf(f[2], f[3]); u.z = pack2bf(f[4], f[5]); u.w = pack2bf(f[6], f[7]);
  return u;
}

// ------------------------------------------------------------------ LayerNorm (+ modulation | affine)
// one wave per row; the row lives in registers (NIT chunks of 8 per lane) -> exact two-pass mean / variance.
template <int NIT>
__global__ __launch_bounds__(256) void layernorm_kernel(LnArgs a) {
#pragma clang fp contract(off)                                    // one fma, written out (the affine form): the same bits as layernorm_pipelined_kernel
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  const int lane = threadIdx.x & 63;
  const int nchunk = a.d >> 3;
  const bf16_t* xp = a.x + (size_t)row * a.ldx;
  float v[NIT][8];
  float sum = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int ch = lane + 64 * it;
    if (ch < nchunk) {
      const uint4 u = *reinterpret_cast<const uint4*>(xp + ch * 8);
      unpack8(u, v[it]);
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[it][j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[it][j] = 0.f;
    }
  }
  const float mean = wave_sum(sum) / (float)a.d;
  float sq = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int ch = lane + 64 * it;
    if (ch < nchunk) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float dlt = v[it][j] - mean; sq += dlt * dlt; }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)a.d + a.eps);
  bf16_t* yp = a.y + (size_t)row * a.ldy;
  const int frame = a.w ? 0 : row / a.rows_per_frame;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int ch = lane + 64 * it;
    if (ch >= nchunk) continue;
    float o[8];
    if (a.w) {
      float w[8], b[8];
      unpack8(*reinterpret_cast<const uint4*>(a.w + ch * 8), w);
      unpack8(*reinterpret_cast<const uint4*>(a.b + ch * 8), b);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = __builtin_fmaf((v[it][j] - mean) * rstd, w[j], b[j]);
    } else {
      float sc[8], sh[8];
      unpack8(*reinterpret_cast<const uint4*>(a.scale + (size_t)frame * a.mod_frame_stride + ch * 8), sc);
      unpack8(*reinterpret_cast<const uint4*>(a.shift + (size_t)frame * a.mod_frame_stride + ch * 8), sh);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float n = rbf((v[it][j] - mean) * rstd);   // norm output is a bf16 tensor
        const float s1 = rbf(1.0f + sc[j]);              // (1 + e) is a bf16 tensor
        o[j] = rbf(n * s1) + sh[j];                      // product rounds, sum rounds at pack
      }
    }
    *reinterpret_cast<uint4*>(yp + ch * 8) = pack8(o);
  }
}

// ------------------------------------------------------------------ pipelined row passes: LayerNorm for the big stages, QK RMSNorm (+RoPE, +KV page write)
// The one-row-per-wave form above runs load -> statistics -> output -> store in lock step on every wave of the chip (a launch
// is ~6 rounds of resident blocks), so HBM idles while the VALU works and vice versa: 3.9 TB/s on 25 200 x 5120, and the q / k
// pass -- two rows per wave, one after the other, 172 registers -- 2.8 TB/s.  Here a block owns a contiguous range of 4-row
// groups and every wave has the NEXT row's 16 B loads in flight while it works on the current one; the row stays PACKED in
// registers (NIT x 16 B per lane) and is unpacked once per pass; q rows and k rows go to different blocks (grid.y).  vmcnt is
// in order on gfx9, so nothing the current row needs may be loaded from global memory behind that prefetch: the per-column
// vectors (modulation / affine / gain) are staged in LDS once per block (and per frame), RoPE factors ride with the prefetch.
// Same arithmetic, rounding for rounding (hashes of the outputs: profiles/r05r_*): 4.4-4.7 / 4.6 TB/s.  (The q / k pass
// of round 4 -- q, then k, in one wave, rows unpacked to floats -- is gone; its record is profiles/r05p_*.)
MMPL_DEV void keep_packed(uint4& u) { asm volatile("" : "+v"(u.x), "+v"(u.y), "+v"(u.z), "+v"(u.w)); }

// FULL: d == 512 NIT and rows % 4 == 0 (every model shape): no per-chunk / per-row predicate anywhere, so the number of
// memory operations between a prefetch and its use is a compile-time constant and the compiler's s_waitcnt vmcnt(N) lets the
// younger ones stay in flight (with predicated loads it has to fall back to vmcnt(0), which would wait for the prefetch too).
template <int NIT, bool FULL>
MMPL_DEV void load_row(uint4 (&u)[NIT], const bf16_t* p, int lane, int nchunk, bool live) {
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int ch = lane + 64 * it;
    if (FULL) u[it] = *reinterpret_cast<const uint4*>(p + ch * 8);
    else u[it] = (live && ch < nchunk) ? *reinterpret_cast<const uint4*>(p + ch * 8) : uint4{0u, 0u, 0u, 0u};
  }
}

template <int NIT, bool FULL>
MMPL_DEV void layernorm_row(const LnArgs& a, uint4 (&cur)[NIT], const uint4 (*sm)[NIT * 64], int staged, int row, int lane, int nchunk) {
#pragma clang fp contract(off)                                    // as layernorm_kernel is compiled: no fma but the affine one
  float sum = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    float v[8];
    unpack8(cur[it], v);
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += v[j];                      // the padding chunks add +0
  }
  const float mean = wave_sum(sum) / (float)a.d;
  float sq = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    keep_packed(cur[it]);
    const bool ok = FULL || lane + 64 * it < nchunk;
    float v[8];
    unpack8(cur[it], v);
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float dlt = ok ? v[j] - mean : 0.f; sq += dlt * dlt; }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)a.d + a.eps);
  bf16_t* yp = a.y + (size_t)row * a.ldy;
  const int frame = a.w ? 0 : row / a.rows_per_frame;             // != staged only where a group straddles two frames
  const bf16_t* p0 = a.w ? a.w : a.scale + (size_t)frame * a.mod_frame_stride;
  const bf16_t* p1 = a.w ? a.b : a.shift + (size_t)frame * a.mod_frame_stride;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int ch = lane + 64 * it;
    if (!FULL && ch >= nchunk) continue;
    keep_packed(cur[it]);
    float v[8], c0[8], c1[8], o[8];
    unpack8(cur[it], v);
    if (FULL || frame == staged) {                                // FULL: rows_per_frame % 4 == 0, a group never straddles
      unpack8(sm[0][ch], c0);
      unpack8(sm[1][ch], c1);
    } else {
      unpack8(*reinterpret_cast<const uint4*>(p0 + ch * 8), c0);
      unpack8(*reinterpret_cast<const uint4*>(p1 + ch * 8), c1);
    }
    if (a.w) {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = __builtin_fmaf((v[j] - mean) * rstd, c0[j], c1[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float n = rbf((v[j] - mean) * rstd);                // norm output is a bf16 tensor
        const float s1 = rbf(1.0f + c0[j]);                       // (1 + e) is a bf16 tensor
        o[j] = rbf(n * s1) + c1[j];                               // product rounds, sum rounds at pack
      }
    }
    *reinterpret_cast<uint4*>(yp + ch * 8) = pack8(o);
  }
}

template <int NIT, bool FULL>
__global__ __launch_bounds__(256) void layernorm_pipelined_kernel(LnArgs a, int groups_per_block) {
  __shared__ uint4 sm[2][NIT * 64];                               // scale | w and shift | b of the staged frame
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // wave-uniform: row, frame, page pointers stay scalar
  const int nchunk = a.d >> 3;
  const int ngroups = (a.rows + 3) >> 2;
  int g = blockIdx.x * groups_per_block;
  const int g_end = min(g + groups_per_block, ngroups);
  if (g >= g_end) return;
  int staged = -1;
  uint4 cur[NIT], nxt[NIT];
  load_row<NIT, FULL>(cur, a.x + (size_t)(4 * g + wv) * a.ldx, lane, nchunk, 4 * g + wv < a.rows);
  for (;; ++g) {
    const int row = 4 * g + wv;
    const bool more = g + 1 < g_end;                              // block-uniform
    const int gframe = a.w ? 0 : (4 * g) / a.rows_per_frame;      // block-uniform
    if (gframe != staged) {                                       // (before the prefetch: a conditional load between the prefetch and
      __syncthreads();                                            //  its use would make the compiler wait with vmcnt(0) on every path)
      const bf16_t* p0 = a.w ? a.w : a.scale + (size_t)gframe * a.mod_frame_stride;
      const bf16_t* p1 = a.w ? a.b : a.shift + (size_t)gframe * a.mod_frame_stride;
      for (int i = threadIdx.x; i < nchunk; i += 256) {
        sm[0][i] = *reinterpret_cast<const uint4*>(p0 + i * 8);
        sm[1][i] = *reinterpret_cast<const uint4*>(p1 + i * 8);
      }
      __syncthreads();
      staged = gframe;
    }
    if (more) load_row<NIT, FULL>(nxt, a.x + (size_t)(row + 4) * a.ldx, lane, nchunk, row + 4 < a.rows);
    asm volatile("" ::: "memory");                                // the prefetch is issued here, not where it is consumed
    if (FULL || row < a.rows) layernorm_row<NIT, FULL>(a, cur, sm, staged, row, lane, nchunk);
    if (!more) break;
#pragma unroll
    for (int it = 0; it < NIT; ++it) { cur[it] = nxt[it]; keep_packed(cur[it]); }   // "used" here: the wait for the prefetch lands before the next one is issued
  }
}

struct RopeRow { float cs[4], sn[4]; int lf, tok; };
MMPL_DEV void rope_row(const QkNormArgs& a, int row, int lane, bool live, RopeRow& r) {
  r.lf = a.rope ? row / a.rows_per_frame : 0;                     // local frame
  r.tok = a.rope ? row - r.lf * a.rows_per_frame : row;           // token inside the frame
  if (a.rope && live) {
    const int gy = r.tok / a.grid_w, gx = r.tok - gy * a.grid_w, ft = a.frame_ids[r.lf];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p = 4 * (lane & 15) + j;                          // rotary pair index inside the head, 0..63
      const int pos = p < 22 ? ft : (p < 43 ? gy : gx);           // split (22, 21, 21): causal_fps_model.py:31
      r.cs[j] = a.cos_tab[pos * 64 + p];
      r.sn[j] = a.sin_tab[pos * 64 + p];
    }
  }
}

template <int NIT, bool FULL>
MMPL_DEV void qknorm_row(const QkNormArgs& a, uint4 (&cur)[NIT], const uint4* sg, const RopeRow& rc, bf16_t* dst, float qs, int lane, int nchunk) {
#pragma clang fp contract(off)                                    // the two fmas of the rotation are written out below
  float sq = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    float v[8];
    unpack8(cur[it], v);
#pragma unroll
    for (int j = 0; j < 8; ++j) sq += v[j] * v[j];                // the padding chunks add +0
  }
  const float rr = rsqrtf(wave_sum(sq) / (float)a.d + a.eps);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int ch = lane + 64 * it;
    if (!FULL && ch >= nchunk) continue;
    keep_packed(cur[it]);
    float v[8], w[8], o[8];
    unpack8(cur[it], v);
    unpack8(sg[ch], w);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = rbf(rbf(v[j] * rr) * w[j]);       // norm.type_as(x) * weight
    if (a.rope) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // which product of each sum is rounded on its own is what hipcc's contraction happened to give the one-row-per-wave
        // kernel this one replaces (read off its ISA: per 16-byte chunk one unfused difference, then fmas) -- written out so
        // that q and the K pages stay bit-identical to what every recorded parity number was measured on
        const float re = o[2 * j], im = o[2 * j + 1], cs = rc.cs[j], sn = rc.sn[j];
        o[2 * j] = j == 0 ? re * cs - im * sn : __builtin_fmaf(re, cs, -(im * sn));
        o[2 * j + 1] = j == 3 ? __builtin_fmaf(re, sn, im * cs) : __builtin_fmaf(im, cs, re * sn);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] *= qs;
    *reinterpret_cast<uint4*>(dst + ch * 8) = pack8(o);
  }
}

template <int NIT, bool FULL>
__global__ __launch_bounds__(256) void qknorm_kernel(QkNormArgs a, int groups_per_block) {
  __shared__ uint4 sg[NIT * 64];                                  // the gain vector of this block's matrix
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // wave-uniform: row, frame, page pointers stay scalar
  const int nchunk = a.d >> 3;
  const int which = blockIdx.y;                                   // 0 q, 1 k, 2 v
  const int ngroups = (a.rows + 3) >> 2;
  int g = blockIdx.x * groups_per_block;
  const int g_end = min(g + groups_per_block, ngroups);
  if (g >= g_end) return;
  if (which == 2) {                                               // V is copied unchanged into its page
    for (; g < g_end; ++g) {
      const int row = 4 * g + wv;
      if (row >= a.rows) break;
      const int lf = a.rope ? row / a.rows_per_frame : 0, tok = a.rope ? row - lf * a.rows_per_frame : row;
      const bf16_t* src = a.v + (size_t)row * a.ldv;
      bf16_t* dst = a.v_dst[lf] + (size_t)tok * a.d;
      for (int ch = lane; ch < nchunk; ch += 64)
        *reinterpret_cast<uint4*>(dst + ch * 8) = *reinterpret_cast<const uint4*>(src + ch * 8);
    }
    return;
  }
  const bf16_t* gain = which == 0 ? a.wq : a.wk;
  for (int i = threadIdx.x; i < nchunk; i += 256) sg[i] = *reinterpret_cast<const uint4*>(gain + i * 8);
  const bf16_t* base = which == 0 ? a.q : a.k;
  const int ld = which == 0 ? a.ldq : a.ldk;
  const float qs = (which == 0 && a.q_scale != 0.f) ? a.q_scale : 1.0f;   // softmax scale folded into q while it is still fp32 (attn_w64.hip)
  uint4 cur[NIT], nxt[NIT];
  RopeRow rc, rn;
  rope_row(a, 4 * g + wv, lane, FULL || 4 * g + wv < a.rows, rc);
  load_row<NIT, FULL>(cur, base + (size_t)(4 * g + wv) * ld, lane, nchunk, 4 * g + wv < a.rows);
  __syncthreads();
  for (;; ++g) {
    const int row = 4 * g + wv;
    const bool more = g + 1 < g_end;                              // block-uniform
    if (more) {
      rope_row(a, row + 4, lane, FULL || row + 4 < a.rows, rn);   // older than the row's loads: back first
      load_row<NIT, FULL>(nxt, base + (size_t)(row + 4) * ld, lane, nchunk, row + 4 < a.rows);
    }
    asm volatile("" ::: "memory");                                // the prefetch is issued here, not where it is consumed
    if (FULL || row < a.rows)
      qknorm_row<NIT, FULL>(a, cur, sg, rc, which == 0 ? a.q + (size_t)row * a.ldq : (a.k_dst[rc.lf] + (size_t)rc.tok * a.d), qs, lane, nchunk);
    if (!more) break;
#pragma unroll
    for (int it = 0; it < NIT; ++it) { cur[it] = nxt[it]; keep_packed(cur[it]); }   // as in layernorm_pipelined_kernel
    rc = rn;
  }
}

// ------------------------------------------------------------------ small kernels
__global__ void modulation_kernel(const bf16_t* mod, size_t mod_layer_stride, const bf16_t* e, int e_frame_stride, int bcast,
                                  bf16_t* emod, int n_layers, int n_frames, int nmod, int d) {
  const size_t per_frame = (size_t)nmod * d;
  const size_t total = (size_t)n_layers * n_frames * per_frame;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t within = i % per_frame;
    const size_t lf = i / per_frame;
    const int f = (int)(lf % n_frames), l = (int)(lf / n_frames);
    const size_t ei = (size_t)f * e_frame_stride + (bcast ? within % d : within);
    emod[i] = f2bf(bf2f(mod[(size_t)l * mod_layer_stride + within]) + bf2f(e[ei]));
  }
}

__global__ void patchify_kernel(const bf16_t* x, bf16_t* a, int lda, int F, int C, int h, int w) {
  const int gh = h >> 1, gw = w >> 1;
  const size_t total = (size_t)F * gh * gw * lda;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % lda);
    const size_t tokg = i / lda;
    const int c = col >> 2, ph = (col >> 1) & 1, pw = col & 1;
    const int gx = (int)(tokg % gw), gy = (int)((tokg / gw) % gh), f = (int)(tokg / ((size_t)gw * gh));
    a[i] = c < C ? x[(((size_t)f * C + c) * h + (2 * gy + ph)) * w + 2 * gx + pw] : (bf16_t)0;      // columns >= 4 C: K padding
  }
}

__global__ void unpatchify_kernel(const bf16_t* y, int ldy, bf16_t* out, int F, int C, int h, int w) {
  const int gh = h >> 1, gw = w >> 1;
  const size_t total = (size_t)F * C * h * w;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int xx = (int)(i % w), yy = (int)((i / w) % h), c = (int)((i / ((size_t)w * h)) % C),
              f = (int)(i / ((size_t)w * h * C));
    const size_t tok = ((size_t)f * gh + (yy >> 1)) * gw + (xx >> 1);
    out[i] = y[tok * ldy + ((yy & 1) * 2 + (xx & 1)) * C + c];       // 'fhwpqrc->cfphqwr'
  }
}

__global__ void sinusoid_kernel(const float* t, bf16_t* out, int F, int freq_dim) {
  const int half = freq_dim >> 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F * half) return;
  const int f = i / half, k = i - f * half;
  const double ang = (double)t[f] * pow(10000.0, -(double)k / (double)half);   // model.py:19-24 (float64)
  out[(size_t)f * freq_dim + k] = f2bf((float)cos(ang));
  out[(size_t)f * freq_dim + half + k] = f2bf((float)sin(ang));
}

__global__ void silu_kernel(const bf16_t* x, bf16_t* y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = f2bf(silu(bf2f(x[i])));
}

// ------------------------------------------------------------------ CFG + FlowUniPC (order <= 2, bh2, predict_x0)
// Every tensor op of fm_solvers_unipc.py:315-331 / 486-626 / 350-484 on a bf16 tensor rounds to bf16; this kernel
// performs the same chain per element with the same rounding points.  Scalars come from the host scheduler.
MMPL_DEV void unipc_body(const UniPCArgs& a) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
    float flow = bf2f(a.flow_c[i]);
    if (a.flow_u) {
      const float fu = bf2f(a.flow_u[i]);
      flow = rbf(fu + rbf(a.guidance * rbf(flow - fu)));            // casual_fps_inference.py:366-367
    }
    float x = bf2f(a.x[i]);
    const float m_conv = rbf(x - rbf(a.sigma_cur * flow));           // convert_model_output
    float m0 = bf2f(a.m0[i]), m1 = bf2f(a.m1[i]);
    if (a.use_corrector) {
      const float ls = bf2f(a.last_sample[i]);
      const float xt_ = rbf(rbf(a.c_c1 * ls) - rbf(a.c_c2 * m0));
      const float d1t = rbf(m_conv - m0);
      float acc = rbf(a.c_rho_last * d1t);
      if (a.corr_order == 2) {
        const float d1 = rbf(rbf(m1 - m0) * a.c_inv_rk);
        acc = rbf(rbf(a.c_rho0 * d1) + acc);
      }
      x = rbf(xt_ - rbf(a.c_c3 * acc));
    }
    m1 = m0;
    m0 = m_conv;
    a.m1[i] = f2bf(m1);
    a.m0[i] = f2bf(m0);
    a.last_sample[i] = f2bf(x);
    float xt = rbf(rbf(a.p_c1 * x) - rbf(a.p_c2 * m0));
    if (a.pred_order == 2) {
      const float d1 = rbf(rbf(m1 - m0) * a.p_inv_rk);
      xt = rbf(xt - rbf(a.p_c3 * rbf(0.5f * d1)));
    }
    a.x[i] = f2bf(xt);
  }
}
__global__ void unipc_kernel(UniPCArgs a) { unipc_body(a); }
__global__ void unipc_table_kernel(UniPCArgs a, const UniPCStepDev* table, const int* step, int n_steps) {
  if (*step >= n_steps) return;          // a replay beyond the uploaded table must not apply garbage coefficients
  const UniPCStepDev st = table[*step];
  a.guidance = st.guidance; a.sigma_cur = st.sigma_cur; a.use_corrector = st.use_corrector; a.corr_order = st.corr_order;
  a.c_c1 = st.c_c1; a.c_c2 = st.c_c2; a.c_c3 = st.c_c3; a.c_inv_rk = st.c_inv_rk; a.c_rho0 = st.c_rho0; a.c_rho_last = st.c_rho_last;
  a.pred_order = st.pred_order; a.p_c1 = st.p_c1; a.p_c2 = st.p_c2; a.p_c3 = st.p_c3; a.p_inv_rk = st.p_inv_rk;
  unipc_body(a);
}
__global__ void unipc_advance_kernel(int* step, float* t_out, const float* t_tab, int n_t, int n_steps) {
  if (*step >= n_steps) return;
  const int nxt = *step + 1;
  __syncthreads();
  if (threadIdx.x == 0) *step = nxt;
  const float t = t_tab[nxt < n_steps ? nxt : n_steps - 1];
  for (int i = threadIdx.x; i < n_t; i += blockDim.x) t_out[i] = t;
}

inline int grid_for(size_t n, int block = 256) {
  size_t g = (n + block - 1) / block;
  return (int)(g > 2048 * 4 ? 2048 * 4 : (g == 0 ? 1 : g));
}

}  // namespace

#define DISPATCH_NIT(NITV, KERNEL, ARGS, ROWS, STREAM)                                                     \
  switch (NITV) {                                                                                          \
    case 1: hipLaunchKernelGGL(KERNEL<1>, dim3(((ROWS) + 3) / 4), dim3(256), 0, STREAM, ARGS); break;      \
    case 2: hipLaunchKernelGGL(KERNEL<2>, dim3(((ROWS) + 3) / 4), dim3(256), 0, STREAM, ARGS); break;      \
    case 3: hipLaunchKernelGGL(KERNEL<3>, dim3(((ROWS) + 3) / 4), dim3(256), 0, STREAM, ARGS); break;      \
    case 4: hipLaunchKernelGGL(KERNEL<4>, dim3(((ROWS) + 3) / 4), dim3(256), 0, STREAM, ARGS); break;      \
    case 5: case 6: hipLaunchKernelGGL(KERNEL<6>, dim3(((ROWS) + 3) / 4), dim3(256), 0, STREAM, ARGS); break; \
    case 7: case 8: hipLaunchKernelGGL(KERNEL<8>, dim3(((ROWS) + 3) / 4), dim3(256), 0, STREAM, ARGS); break; \
    case 9: case 10: hipLaunchKernelGGL(KERNEL<10>, dim3(((ROWS) + 3) / 4), dim3(256), 0, STREAM, ARGS); break; \
    default: return hipErrorInvalidValue;                                                                  \
  }
// pipelined kernels: all blocks resident at once (blocks per CU from the occupancy query, cached per instantiation), each with
// a contiguous range of 4-row groups
template <typename K>
static int resident_blocks(K kernel, int* cache) {
  if (!*cache) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    *cache = per_cu * 8 * mmpl_cus_per_xcd();
  }
  return *cache;
}
template <typename A>
static void launch_pipelined(void (*kernel)(A, int), int* cache, const A& a, int rows, int ny, hipStream_t s) {
  const int ngroups = (rows + 3) / 4, resident = resident_blocks(kernel, cache);
  int gpb = (ngroups + resident - 1) / resident;
  hipLaunchKernelGGL(kernel, dim3((ngroups + gpb - 1) / gpb, ny), dim3(256), 0, s, a, gpb);
}
#define PIPELINED_CASE(N, KERNEL, ARGS, ROWS, NY, STREAM, FULLV)                                           \
  { static int c0 = 0, c1 = 0;                                                                             \
    if (FULLV) launch_pipelined(KERNEL<N, true>, &c1, ARGS, ROWS, NY, STREAM);                             \
    else launch_pipelined(KERNEL<N, false>, &c0, ARGS, ROWS, NY, STREAM); }
#define DISPATCH_PIPELINED(NITV, KERNEL, ARGS, ROWS, NY, STREAM, FULLV)                                    \
  switch (NITV) {                                                                                          \
    case 1: PIPELINED_CASE(1, KERNEL, ARGS, ROWS, NY, STREAM, FULLV) break;                                \
    case 2: PIPELINED_CASE(2, KERNEL, ARGS, ROWS, NY, STREAM, FULLV) break;                                \
    case 3: PIPELINED_CASE(3, KERNEL, ARGS, ROWS, NY, STREAM, FULLV) break;                                \
    case 4: PIPELINED_CASE(4, KERNEL, ARGS, ROWS, NY, STREAM, FULLV) break;                                \
    case 5: case 6: PIPELINED_CASE(6, KERNEL, ARGS, ROWS, NY, STREAM, (FULLV) && (NITV) == 6) break;       \
    case 7: case 8: PIPELINED_CASE(8, KERNEL, ARGS, ROWS, NY, STREAM, (FULLV) && (NITV) == 8) break;       \
    case 9: case 10: PIPELINED_CASE(10, KERNEL, ARGS, ROWS, NY, STREAM, (FULLV) && (NITV) == 10) break;    \
    default: return hipErrorInvalidValue;                                                                  \
  }

hipError_t mmpl_launch_layernorm(const LnArgs& a, hipStream_t s) {
  if (a.rows <= 0) return hipSuccess;
  if (a.d % 8 || a.d > 5120 || a.ldx % 8 || a.ldy % 8) return hipErrorInvalidValue;
  const int nit = (a.d / 8 + 63) / 64;
  // measured (profiles/r05q_*, r05r_*, r05x_*): the pipeline pays from ~6 row groups per block on rows of >= 6 KB (25 200 x 5120: 131 -> 110 us,
  // 21 600: 110 -> 98); below that its prologue / tail cost more than the overlap returns (10 920 x 5120: 44 vs 46 us, 9360: 40 vs 45,
  // 7200: 33 vs 35; 1536-wide rows: 16 vs 18).  MMPL_LN_PIPELINE_MIN_ROWS moves the threshold (mmpl_config.h).
  if (nit >= 6 && a.rows >= mmpl_config().ln_pipeline_min_rows) {
    const bool full = a.d == 512 * nit && a.rows % 4 == 0 && (a.w || a.rows_per_frame % 4 == 0);
    DISPATCH_PIPELINED(nit, layernorm_pipelined_kernel, a, a.rows, 1, s, full);
  } else {
    DISPATCH_NIT(nit, layernorm_kernel, a, a.rows, s);
  }
  return hipGetLastError();
}

hipError_t mmpl_launch_qknorm(const QkNormArgs& a, hipStream_t s) {
  if (a.rows <= 0) return hipSuccess;
  if (a.d % 128 || a.d > 5120 || a.ldq % 8 || a.ldk % 8 || a.ldv % 8) return hipErrorInvalidValue;
  if (a.v && !a.k) return hipErrorInvalidValue;
  const int nit = (a.d / 8 + 63) / 64;
  const bool full = a.d == 512 * nit && a.rows % 4 == 0;
  DISPATCH_PIPELINED(nit, qknorm_kernel, a, a.rows, 1 + (a.k ? 1 : 0) + (a.v ? 1 : 0), s, full);
  return hipGetLastError();
}

hipError_t mmpl_launch_rmsnorm(bf16_t* x, int ldx, const bf16_t* w, int rows, int d, float eps, hipStream_t s, float out_scale) {
  QkNormArgs a = {};
  a.q = x; a.ldq = ldx; a.wq = w; a.rows = rows; a.d = d; a.eps = eps; a.rope = 0; a.rows_per_frame = rows > 0 ? rows : 1;
  a.q_scale = out_scale;
  a.grid_w = 1;
  return mmpl_launch_qknorm(a, s);
}

hipError_t mmpl_launch_modulation(const bf16_t* mod, size_t mod_layer_stride, const bf16_t* e, int e_frame_stride, int bcast,
                                  bf16_t* emod, int n_layers, int n_frames, int nmod, int d, hipStream_t s) {
  const size_t n = (size_t)n_layers * n_frames * nmod * d;
  hipLaunchKernelGGL(modulation_kernel, dim3(grid_for(n)), dim3(256), 0, s, mod, mod_layer_stride, e, e_frame_stride, bcast, emod,
                     n_layers, n_frames, nmod, d);
  return hipGetLastError();
}
hipError_t mmpl_launch_patchify(const bf16_t* x, bf16_t* a, int lda, int F, int C, int h, int w, hipStream_t s) {
  const size_t n = (size_t)F * (h / 2) * (w / 2) * lda;
  hipLaunchKernelGGL(patchify_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, a, lda, F, C, h, w);
  return hipGetLastError();
}
hipError_t mmpl_launch_unpatchify(const bf16_t* y, int ldy, bf16_t* out, int F, int C, int h, int w, hipStream_t s) {
  const size_t n = (size_t)F * C * h * w;
  hipLaunchKernelGGL(unpatchify_kernel, dim3(grid_for(n)), dim3(256), 0, s, y, ldy, out, F, C, h, w);
  return hipGetLastError();
}
// flags[r] = 1 iff row r of x ([rows, d] bf16, d % 8 == 0) is bitwise identical to row rows - 1 (one wave per row)
__global__ __launch_bounds__(64) void rows_equal_last_kernel(const bf16_t* x, int ld, int rows, int d, int* flags) {
  const int r = blockIdx.x;
  const u32x4* a = reinterpret_cast<const u32x4*>(x + (size_t)r * ld);
  const u32x4* b = reinterpret_cast<const u32x4*>(x + (size_t)(rows - 1) * ld);
  bool same = true;
  for (int i = threadIdx.x; i < d / 8; i += 64) {
    const u32x4 u = a[i], v = b[i];
    same = same && u[0] == v[0] && u[1] == v[1] && u[2] == v[2] && u[3] == v[3];
  }
  const bool all = __all(same);
  if (threadIdx.x == 0) flags[r] = all ? 1 : 0;
}
hipError_t mmpl_launch_rows_equal_last(const bf16_t* x, int ld, int rows, int d, int* flags, hipStream_t s) {
  if (d % 8 || ld % 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(rows_equal_last_kernel, dim3(rows), dim3(64), 0, s, x, ld, rows, d, flags);
  return hipGetLastError();
}
// ---- MMPL_CHECK_SHARE=1 (api.hip): position-sensitive 128-bit fingerprint of a list of equally sized pages: fp[0] += sum of the
// 32-bit words, fp[1] += sum of word * (global word index + 1), both mod 2^64 (integer sums: independent of the order of the adds)
__global__ __launch_bounds__(256) void pages_fingerprint_kernel(PageList pl, size_t words_per_page, unsigned long long* fp) {
  const size_t total = (size_t)pl.n * words_per_page;
  unsigned long long s1 = 0, s2 = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t pg = i / words_per_page, off = i - pg * words_per_page;
    const unsigned long long w = reinterpret_cast<const uint32_t*>(pl.p[pg])[off];
    s1 += w;
    s2 += w * (unsigned long long)(i + 1);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    s1 += __shfl_xor(s1, m, 64);
    s2 += __shfl_xor(s2, m, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(fp, s1);
    atomicAdd(fp + 1, s2);
  }
}
// chk = {producer fp[2], consumer fp[2], mismatches}: which = 0 zeroes the producer's pair, 1 the consumer's
__global__ void share_check_zero_kernel(unsigned long long* chk, int which) {
  if (threadIdx.x < 2) chk[2 * which + threadIdx.x] = 0;
}
__global__ void share_check_compare_kernel(unsigned long long* chk) {
  if (threadIdx.x == 0 && (chk[0] != chk[2] || chk[1] != chk[3])) chk[4] += 1;
}
hipError_t mmpl_launch_pages_fingerprint(const PageList& pl, size_t bytes_per_page, unsigned long long* chk, int which, hipStream_t s) {
  if (pl.n < 1 || pl.n > MMPL_MAX_PAGES || bytes_per_page % 4) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pages_fingerprint_kernel, dim3(2048), dim3(256), 0, s, pl, bytes_per_page / 4, chk + 2 * which);
  return hipGetLastError();
}
hipError_t mmpl_launch_share_check_zero(unsigned long long* chk, int which, hipStream_t s) {
  hipLaunchKernelGGL(share_check_zero_kernel, dim3(1), dim3(64), 0, s, chk, which);
  return hipGetLastError();
}
hipError_t mmpl_launch_share_check_compare(unsigned long long* chk, hipStream_t s) {
  hipLaunchKernelGGL(share_check_compare_kernel, dim3(1), dim3(64), 0, s, chk);
  return hipGetLastError();
}
__global__ void zero_ints_kernel(int* p, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0;
}
// a kernel rather than hipMemsetAsync: inside a captured hipGraph it is an ordinary kernel node, ordered like every other launch
// (a memset NODE was seen to run unordered under rocprofv3's kernel tracing: garbage tile tickets -> faults / hangs)
hipError_t mmpl_launch_zero_ints(int* p, int n, hipStream_t s) {
  hipLaunchKernelGGL(zero_ints_kernel, dim3(1), dim3(64), 0, s, p, n);
  return hipGetLastError();
}
hipError_t mmpl_launch_sinusoid(const float* t, bf16_t* out, int F, int freq_dim, hipStream_t s) {
  const int n = F * (freq_dim / 2);
  hipLaunchKernelGGL(sinusoid_kernel, dim3((n + 255) / 256), dim3(256), 0, s, t, out, F, freq_dim);
  return hipGetLastError();
}
hipError_t mmpl_launch_silu(const bf16_t* x, bf16_t* y, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(silu_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, y, n);
  return hipGetLastError();
}
hipError_t mmpl_launch_unipc(const UniPCArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(unipc_kernel, dim3(grid_for(a.n)), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t mmpl_launch_unipc_table(const UniPCArgs& a, const UniPCStepDev* table, int* step, float* t_out, const float* t_tab,
                                   int n_t, int n_steps, hipStream_t s) {
  hipLaunchKernelGGL(unipc_table_kernel, dim3(grid_for(a.n)), dim3(256), 0, s, a, table, step, n_steps);
  hipLaunchKernelGGL(unipc_advance_kernel, dim3(1), dim3(64), 0, s, step, t_out, t_tab, n_t, n_steps);
  return hipGetLastError();
}
