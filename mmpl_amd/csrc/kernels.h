// Internal launcher interface between the C-ABI layer (api.hip) and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"

// ---------------------------------------------------------------- per-device launcher state (device_state.hip)
hipError_t mmpl_dyn_smem_once(const void* func, int bytes);   // hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel)
int mmpl_cus_per_xcd();                                       // CUs / 8 of the current device

// ---------------------------------------------------------------- GEMM (gemm.hip)
enum GemmEpi { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_SILU = 2, EPI_GATE_RES = 3, EPI_RES = 4, EPI_F32_SCALE = 5,
               EPI_BIAS_VPAGES = 6 };   // bias; columns >= v_col0 go to per-frame pages instead of C (the V third of the fused qkv)
struct GemmArgs {
  const bf16_t* A; int lda;      // [M, K]
  const bf16_t* W; int ldw;      // [N, K]  (nn.Linear weight)
  const bf16_t* bias;            // [N] or null
  bf16_t* C; int ldc;            // [M, N]
  int M, N, K;
  int epi;
  const bf16_t* res; int ldres;  // residual x for EPI_GATE_RES / EPI_RES (may alias C)
  const bf16_t* gate;            // per-frame gate vectors: gate[frame * gate_frame_stride + n]
  int gate_frame_stride;
  int rows_per_frame;
  float alpha;                   // EPI_F32_SCALE: C is float*, C = alpha * acc (no bias)
  int group;                     // M-tile group size of the block order (set by the launcher)
  int batch;                     // > 1: batched (blockIdx.y); element strides below; small-problem kernel only
  long sA, sW, sC;
  // EPI_BIAS_VPAGES: row m, column n >= v_col0 is written to v_dst[m / rows_per_frame] + (m % rows_per_frame) * v_ld + n - v_col0
  bf16_t* v_dst[8]; int v_col0, v_ld;
  int staged_epilogue;           // v6: LDS-staged 16-byte epilogue (set by the launcher when every pointer / stride allows it)
  // v6, optional: 8 device ints (one per XCD), zero when the launch starts and left zero by it.  With it the kernel is launched
  // once per CU and its blocks take tiles of their XCD's share from these tickets until none is left (gemm.hip); the
  // launches that share a counter must be ordered (one stream).  Null: one block per tile.
  int* tile_counter;
  int pf_dist;                   // v6: L2 prefetch distance in k-tiles (0 = off; set by the launcher)
  int sync_sweeps;               // v6: deal whole M-groups to the XCDs so that all of them sweep W's column panels together (set by the launcher)
  // v6, optional (needs tile_counter too): scratch for the split-K launch of the partial last round of tiles (gemm.hip).
  // splitk_ws: mmpl_gemm_splitk_ws_bytes() of fp32 partials; splitk_cnt: 256 device ints, zero when a launch starts and left zero.
  float* splitk_ws; int* splitk_cnt;
  int splitk_s, splitk_tb, splitk_per;   // set by the launcher: parts per tile (1 = no split), tail tiles per XCD (max), CUs per XCD
};
size_t mmpl_gemm_splitk_ws_bytes();
bool mmpl_xcd_dispatch_ok(bool may_probe);   // device_state.hip: workgroup b runs on XCD b & 7 (probed once per device, never inside a capture)
hipError_t mmpl_launch_gemm(const GemmArgs& g, hipStream_t s);

// ---------------------------------------------------------------- attention (attention.hip)
constexpr int MMPL_MAX_PAGES = 24;
struct AttnArgs {
  const bf16_t* q; int ldq;       // row r, head h at q + r*ldq + h*128
  bf16_t* o; int ldo;
  const bf16_t* k_pages[MMPL_MAX_PAGES];   // page p: row j, head h at k_pages[p] + j*ldk + h*128
  const bf16_t* v_pages[MMPL_MAX_PAGES];
  int ldk, ldv;
  int n_pages, page_rows;
  int page_rows_each[MMPL_MAX_PAGES];   // attn_w64_kernel only, filled by mmpl_launch_attention: rows of page p (merged runs of slots differ in length)
  // host side only (merge_contiguous_pages): which ALLOCATION page p lies in, as the caller knows it (the DiT forward: 0 = KV-cache
  // slots, 1 = the stage's scratch pages).  Pages are ordered by (group, address) and merged within a group only, so the order and
  // the tile boundaries the kernel sees -- hence the fp32 summation order, hence the bits -- do not depend on where the allocator
  // happened to put one allocation relative to another (a fresh process and a long-lived one must agree: tests/test_wavefront_gpu.py).
  unsigned char page_group[MMPL_MAX_PAGES];
  int Lq, H;
  float scale;                     // softmax scale (1/sqrt(128))
  int cross;                       // 1: text cross-attention launch (symbol tag only)
  float* split_ws; size_t split_ws_bytes;   // optional scratch for the split-KV tail round (nullptr: never split)
  int variant;                     // ATTN_* kernel selector (0 = auto)
  int q_prescaled;                 // q was already multiplied by scale * log2(e) by its producer (ATTN_W64 only)
  // attn_fwd_kernel with ONE page only: the page's last row stands for `last_row_copies` identical keys (same K row, same V row):
  // its score gets + ln(copies) / scale, i.e. its softmax weight is multiplied by copies.  0 / 1 = a plain row.
  int last_row_copies;
  // attn_w64_kernel, optional: five device counters {blocks run, blocks whose FAST pass failed and were redone by the GENERAL pass,
  // WAVES (64 query rows) that held a failing row themselves, blocks their history sent straight to the GENERAL pass, blocks whose
  // FAST pass held on the references their history remembered} (how data-dependent is the kernel's time on THIS input? -- bench.py --heavy-tail)
  unsigned long long* redo_stats;
  // attn_w64_kernel, optional: mmpl_attn_history_bytes(Lq, H) bytes carried by the caller from one launch to the next launch of the same
  // attention (same layer, CFG branch and stage shape); all zero = no history.  [H * ceil(Lq / 256) * 4] state bytes (head, query block,
  // split part), padded to 256 B, then per state byte 128 int16 lane references (attn_w64.hip).  nullptr: stateless.
  unsigned char* history;
  short* history_mem;              // set by mmpl_launch_attention: history + the padded state bytes
};
// Work item `local` of XCD `xcd` -> (head, query block) for attn_w64_kernel / attn_merge_kernel (the hardware deals workgroups
// round-robin to the 8 XCDs: blockIdx & 7).  H % 8 == 0: XCD x owns heads x, x+8, ...; any other head count (Wan 1.3B: 12):
// XCD x owns the x-th contiguous chunk of ceil(n_qb*H / 8) items of the head-major (head, query block) list, so the blocks
// sharing one L2 work on at most a few heads; false = past the end of the list (the last XCD's chunk can be short).
__device__ __forceinline__ bool mmpl_attn_item(int H, int n_qb, int xcd, int local, int& head, int& qb) {
  if ((H & 7) == 0) {
    head = xcd + 8 * (local / n_qb);
    qb = local % n_qb;
    return head < H;
  }
  const int total = n_qb * H, per = (total + 7) >> 3, item = xcd * per + local;
  if (local >= per || item >= total) return false;
  head = item / n_qb;
  qb = item % n_qb;
  return true;
}
enum { ATTN_AUTO = 0, ATTN_LOCKSTEP = 1, ATTN_W64 = 3 };      // (2 was the round-1 ping-pong kernel, removed)
// The kernel ATTN_AUTO resolves to for a self-attention launch whose producer can fold the softmax scale into q before q is
// rounded to bf16 (the DiT forward: qknorm_kernel's q_scale).  ATTN_W64 computes exp2(K.q) without a per-score multiply, so it
// wants q = bf16(q_fp32 * scale * log2(e)); handing it a bf16 q to prescale itself costs a second rounding of q, which shows
// on sharp softmax rows (large QK-norm gains) -- a raw-q ATTN_AUTO launch (the attention() seam) therefore takes ATTN_LOCKSTEP.
int mmpl_attention_self_variant();
// attn_w64.hip (4 waves x 64 query rows, one wave per SIMD)
int mmpl_attention_w64_smem();
const void* mmpl_attention_w64_symbol(int split);
void mmpl_launch_attention_w64(const AttnArgs& a, int blocks, int local_base, int sp, bool split, hipStream_t s);
size_t mmpl_attention_split_ws_bytes();     // upper bound of what a launch can use
inline size_t mmpl_attention_history_state_bytes(int Lq, int H) { return ((size_t)H * ((Lq + 255) / 256) * 4 + 255) & ~(size_t)255; }
inline size_t mmpl_attention_history_bytes(int Lq, int H) { return mmpl_attention_history_state_bytes(Lq, H) + (size_t)H * ((Lq + 255) / 256) * 4 * 128 * sizeof(short); }
hipError_t mmpl_launch_attention(const AttnArgs& a, hipStream_t s);

// ---------------------------------------------------------------- norms / rope / elementwise (elementwise.hip)
// LayerNorm (eps, fp32 stats) fused with per-frame modulation  y = bf16(bf16(LN(x)) * bf16(1+scale)) + shift
// or with an affine (w, b).  mod pointers index [frame * mod_frame_stride + col].
struct LnArgs {
  const bf16_t* x; int ldx;
  bf16_t* y; int ldy;
  int rows, d;
  float eps;
  const bf16_t* scale; const bf16_t* shift; int mod_frame_stride; int rows_per_frame;  // modulation form
  const bf16_t* w; const bf16_t* b;                                                    // affine form (if w != null)
};
hipError_t mmpl_launch_layernorm(const LnArgs& a, hipStream_t s);

// Full-dim RMSNorm (+ optional 3-axis RoPE, + optional K/V page write) on the fused qkv projection.
struct QkNormArgs {
  bf16_t* q; int ldq;                 // in/out (in place)
  const bf16_t* k; int ldk;           // in (null: q only)
  const bf16_t* v; int ldv;           // in
  const bf16_t* wq; const bf16_t* wk; // RMSNorm gains [d]
  int rows, d;
  float eps;
  float q_scale;                      // q is multiplied by this after RoPE, before its (single) rounding to bf16; 0 = 1
  int rope;                           // 0: no rope (cross-attn q / context k)
  const float* cos_tab; const float* sin_tab;  // [1024][64] fp32
  int frame_ids[8];                   // temporal rope position per local frame
  bf16_t* k_dst[8]; bf16_t* v_dst[8]; // per local frame destination page base (row stride = d); k_out for no-page mode
  int rows_per_frame, grid_w;
};
hipError_t mmpl_launch_qknorm(const QkNormArgs& a, hipStream_t s);

// rmsnorm of a plain [rows, d] matrix in place (context K)
// out_scale != 0: the normalised row is multiplied by it before its (single) rounding (a q that feeds ATTN_W64: scale * log2 e)
hipError_t mmpl_launch_rmsnorm(bf16_t* x, int ldx, const bf16_t* w, int rows, int d, float eps, hipStream_t s, float out_scale = 0.f);

// emod[l][f][k][:] = bf16(mod[l*mod_layer_stride + k*d + :] + e[f*e_frame_stride + (bcast ? : : k*d + :)])   (k < nmod)
hipError_t mmpl_launch_modulation(const bf16_t* mod, size_t mod_layer_stride, const bf16_t* e, int e_frame_stride, int bcast,
                                  bf16_t* emod, int n_layers, int n_frames, int nmod, int d, hipStream_t s);
// patchify: x[F, C, h, w] -> A[F*gh*gw, lda]  (col = c*4 + ph*2 + pw; columns >= 4 C are zero: K padded to a multiple of 64)
hipError_t mmpl_launch_patchify(const bf16_t* x, bf16_t* a, int lda, int F, int C, int h, int w, hipStream_t s);
// unpatchify: y[F*gh*gw, 4*C] (col = (ph*2+pw)*C + c) -> out[F, C, h, w]
// a = bf16(a + b), n elements (i2v.hip)
hipError_t mmpl_launch_add(bf16_t* a, const bf16_t* b, size_t n, hipStream_t s);
hipError_t mmpl_launch_unpatchify(const bf16_t* y, int ldy, bf16_t* out, int F, int C, int h, int w, hipStream_t s);
// sinusoidal timestep embedding (fp64 math): t[F] fp32 -> out[F, freq_dim] bf16 ([cos | sin])
hipError_t mmpl_launch_zero_ints(int* p, int n, hipStream_t s);
// MMPL_CHECK_SHARE=1 (api.hip: the guard of mmpl_dit_forward's share_in promise).  chk: 5 device uint64 = {producer fingerprint[2],
// consumer fingerprint[2], mismatch count}; a fingerprint is accumulated over a list of equally sized pages (K pages, then V pages).
struct PageList { const void* p[MMPL_MAX_PAGES]; int n; };
hipError_t mmpl_launch_share_check_zero(unsigned long long* chk, int which, hipStream_t s);
hipError_t mmpl_launch_pages_fingerprint(const PageList& pl, size_t bytes_per_page, unsigned long long* chk, int which, hipStream_t s);
hipError_t mmpl_launch_share_check_compare(unsigned long long* chk, hipStream_t s);
// flags[r] = (row r of x[rows, d] == row rows - 1, bitwise)
hipError_t mmpl_launch_rows_equal_last(const bf16_t* x, int ld, int rows, int d, int* flags, hipStream_t s);
hipError_t mmpl_launch_sinusoid(const float* t, bf16_t* out, int F, int freq_dim, hipStream_t s);
hipError_t mmpl_launch_silu(const bf16_t* x, bf16_t* y, size_t n, hipStream_t s);

// CFG combine + one FlowUniPC step (elementwise.hip); scalars computed on the host exactly as the reference does
struct UniPCArgs {
  const bf16_t* flow_c; const bf16_t* flow_u;   // flow_u == null: flow_c is already the combined flow
  float guidance;
  bf16_t* x;          // sample in / next sample out
  bf16_t* m0; bf16_t* m1; bf16_t* last_sample;   // solver state (updated in place)
  size_t n;
  float sigma_cur;
  int use_corrector, corr_order; float c_c1, c_c2, c_c3, c_inv_rk, c_rho0, c_rho_last;
  int pred_order; float p_c1, p_c2, p_c3, p_inv_rk;
};
hipError_t mmpl_launch_unipc(const UniPCArgs& a, hipStream_t s);
// Device-resident form for a hipGraph of a whole denoise step: the scalars of step *step come from table[*step] (same
// layout as MmplUniPCStep), then *step is advanced and the NEXT step's timestep is written to t_out[0..n_t) (the tensor
// the two DiT forwards of the next replay read).  `a` carries the pointers and n only.
struct UniPCStepDev {
  float guidance, sigma_cur; int use_corrector, corr_order; float c_c1, c_c2, c_c3, c_inv_rk, c_rho0, c_rho_last;
  int pred_order; float p_c1, p_c2, p_c3, p_inv_rk;
};
hipError_t mmpl_launch_unipc_table(const UniPCArgs& a, const UniPCStepDev* table, int* step, float* t_out, const float* t_tab,
                                   int n_t, int n_steps, hipStream_t s);
