// C ABI of libmmpl_hip.so (see include/mmpl_hip.h) and the host-side orchestration of one Wan DiT forward.
// The forward is a fixed sequence of kernel launches on the caller's stream with no host synchronisation and
// no allocation, so it is hipGraph-capturable per (stage, pass) shape.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <utility>
#include <vector>

#include "../../include/mmpl_hip.h"
#include "kernels.h"
#include "mmpl_config.h"

static thread_local std::string g_err;
static int fail(const char* where, const char* what) {
  g_err = std::string(where) + ": " + what;
  return 1;
}
int mmpl_set_error(const char* where, const char* what) { return fail(where, what); }
#define HIP_TRY(expr, where)                                     \
  do {                                                           \
    hipError_t e__ = (expr);                                     \
    if (e__ != hipSuccess) return fail(where, hipGetErrorString(e__)); \
  } while (0)

// ---- optional per-kernel-class timing (bench / profiling only; per host thread, off by default; launches recorded
// while the stream is being captured into a hipGraph are not timed -- an event pair inside a graph cannot be read back)
enum { K_GEMM, K_ATTN_SELF, K_ATTN_CROSS, K_LAYERNORM, K_QKNORM, K_ELEMENTWISE, K_UNIPC, K_VAE, K_NKINDS };
namespace {
struct Prof {
  bool on = false;
  std::vector<hipEvent_t> ev;
  std::vector<int> kind;
  std::vector<double> flops;
  size_t used = 0;  // pairs
  unsigned mask = ~0u;  // kinds that are timed (mmpl_profile_enable: on = 1 all kinds, on > 1 = bit mask << 1)
};
thread_local Prof g_prof;
struct ProfScope {
  hipStream_t s;
  bool active;
  ProfScope(int kind, double flops, hipStream_t st) : s(st), active(g_prof.on && (g_prof.mask >> kind & 1)) {
    if (active) {
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) active = false;
    }
    if (!active) return;
    if (g_prof.ev.size() < 2 * (g_prof.used + 1)) {
      hipEvent_t a, b;
      hipEventCreate(&a);
      hipEventCreate(&b);
      g_prof.ev.push_back(a);
      g_prof.ev.push_back(b);
      g_prof.kind.push_back(0);
      g_prof.flops.push_back(0);
    }
    g_prof.kind[g_prof.used] = kind;
    g_prof.flops[g_prof.used] = flops;
    hipEventRecord(g_prof.ev[2 * g_prof.used], s);
  }
  ~ProfScope() {
    if (!active) return;
    hipEventRecord(g_prof.ev[2 * g_prof.used + 1], s);
    ++g_prof.used;
  }
};
}  // namespace

enum {  // global weight slots
  G_PE_W, G_PE_B, G_TXT0_W, G_TXT0_B, G_TXT2_W, G_TXT2_B, G_TE0_W, G_TE0_B, G_TE2_W, G_TE2_B, G_TP_W, G_TP_B,
  G_HEAD_MOD, G_HEAD_W, G_HEAD_B, G_BLOCK_MOD, NG
};
enum {  // per-layer weight slots
  L_QKV_W, L_QKV_B, L_NQ, L_NK, L_O_W, L_O_B, L_N3_W, L_N3_B, L_CQ_W, L_CQ_B, L_CNQ, L_CK_W, L_CK_B, L_CNK, L_CV_W,
  L_CV_B, L_CO_W, L_CO_B, L_F0_W, L_F0_B, L_F2_W, L_F2_B, NL
};
static const char* kGlobalNames[NG] = {
    "patch_embedding.weight", "patch_embedding.bias", "text_embedding.0.weight", "text_embedding.0.bias",
    "text_embedding.2.weight", "text_embedding.2.bias", "time_embedding.0.weight", "time_embedding.0.bias",
    "time_embedding.2.weight", "time_embedding.2.bias", "time_projection.1.weight", "time_projection.1.bias",
    "head.modulation", "head.head.weight", "head.head.bias", "pack:blocks.*.modulation[L,6,dim]"};
static const char* kLayerNames[NL] = {
    "pack:self_attn.{q,k,v}.weight[3dim,dim]", "pack:self_attn.{q,k,v}.bias[3dim]", "self_attn.norm_q.weight",
    "self_attn.norm_k.weight", "self_attn.o.weight", "self_attn.o.bias", "norm3.weight", "norm3.bias",
    "cross_attn.q.weight", "cross_attn.q.bias", "cross_attn.norm_q.weight", "cross_attn.k.weight", "cross_attn.k.bias",
    "cross_attn.norm_k.weight", "cross_attn.v.weight", "cross_attn.v.bias", "cross_attn.o.weight", "cross_attn.o.bias",
    "ffn.0.weight", "ffn.0.bias", "ffn.2.weight", "ffn.2.bias"};

enum { I_K_W, I_K_B, I_V_W, I_V_B, I_NK, NI };   // per-layer slots of the Wan-I2V image stream (cross_attn.{k_img,v_img,norm_k_img})

struct MmplDit {
  MmplDitConfig cfg;
  int S, gh, gw;
  int pe_k;                    // K of the patch-embedding GEMM: 4 * in_dim rounded up to a multiple of 64 (64 | 192)
  // Wan-I2V (model_type 'i2v', model.py:595,615-616): per-layer image K / V of the current image, [num_layers, n_img, dim]
  const bf16_t* img_k = nullptr;
  const bf16_t* img_v = nullptr;
  int n_img = 0;
  float* cos_tab = nullptr;  // [1024][64]
  float* sin_tab = nullptr;
  std::vector<const bf16_t*> w;
  unsigned long long* attn_stats = nullptr;    // mmpl_dit_set_attn_stats: 4 counters of the self-attention launches (header)
  unsigned long long* share_chk = nullptr;     // MMPL_CHECK_SHARE=1: {producer fingerprint[2], consumer fingerprint[2], mismatches} (device)
  const bf16_t* G(int i) const { return w[i]; }
  const bf16_t* Lw(int l, int i) const { return w[NG + l * NL + i]; }
};

extern "C" {

int mmpl_profile_enable(int on) {
  g_prof.on = on != 0;
  g_prof.mask = on > 1 ? (unsigned)on >> 1 : ~0u;
  g_prof.used = 0;
  return 0;
}
int mmpl_profile_read(int n_kinds, double* ms, double* flops, long long* launches) {
  if (n_kinds > K_NKINDS) n_kinds = K_NKINDS;
  for (int i = 0; i < n_kinds; ++i) { ms[i] = 0; flops[i] = 0; launches[i] = 0; }
  if (hipDeviceSynchronize() != hipSuccess) return fail("mmpl_profile_read", "sync failed");
  for (size_t i = 0; i < g_prof.used; ++i) {
    float t = 0;
    if (hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) continue;
    const int k = g_prof.kind[i];
    if (k < n_kinds) { ms[k] += t; flops[k] += g_prof.flops[i]; launches[k] += 1; }
  }
  g_prof.used = 0;
  return 0;
}

const char* mmpl_last_error(void) { return g_err.c_str(); }
const char* mmpl_version(void) { return "mmpl_hip 0.1 (gfx950)"; }
int mmpl_dit_num_weights(const MmplDitConfig* cfg) { return NG + cfg->num_layers * NL; }
const char* mmpl_dit_weight_name(int slot, int per_layer) {
  if (per_layer) return (slot >= 0 && slot < NL) ? kLayerNames[slot] : nullptr;
  return (slot >= 0 && slot < NG) ? kGlobalNames[slot] : nullptr;
}

int mmpl_dit_create(const MmplDitConfig* cfg, MmplDit** out) {
  if (!cfg || !out) return fail("mmpl_dit_create", "null argument");
  if (cfg->dim % cfg->num_heads || cfg->dim / cfg->num_heads != 128)
    return fail("mmpl_dit_create", "head_dim must be 128 (both Wan2.1 models)");
  if (cfg->dim > 5120 || cfg->dim % 128 || cfg->ffn_dim % 64 || cfg->text_dim % 64 || cfg->freq_dim % 64)
    return fail("mmpl_dit_create", "unsupported dims");
  if (cfg->lat_h % 2 || cfg->lat_w % 2 || cfg->max_frames < 1 || cfg->max_frames > 8 || (cfg->in_dim != 16 && cfg->in_dim != 36) || cfg->out_dim != 16)
    return fail("mmpl_dit_create", "unsupported geometry");
  MmplDit* h = new MmplDit();
  h->cfg = *cfg;
  h->gh = cfg->lat_h / 2;
  h->gw = cfg->lat_w / 2;
  h->S = h->gh * h->gw;
  h->pe_k = (4 * cfg->in_dim + 63) / 64 * 64;
  // RoPE tables: rope_params(1024, d - 4*(d//6)) | (1024, 2*(d//6)) x2, theta 1e4, fp64 (model.py:29-36,
  // causal_fps_model.py:510-516), stored as fp32 cos / sin.
  const int d = 128, dims[3] = {d - 4 * (d / 6), 2 * (d / 6), 2 * (d / 6)};
  std::vector<float> c(1024 * 64), s(1024 * 64);
  for (int pos = 0; pos < 1024; ++pos) {
    int p = 0;
    for (int part = 0; part < 3; ++part)
      for (int j = 0; j < dims[part] / 2; ++j, ++p) {
        const double freq = 1.0 / pow(10000.0, (double)(2 * j) / (double)dims[part]);
        const double ang = (double)pos * freq;
        c[pos * 64 + p] = (float)cos(ang);
        s[pos * 64 + p] = (float)sin(ang);
      }
  }
  if (hipMalloc(&h->cos_tab, c.size() * 4) != hipSuccess || hipMalloc(&h->sin_tab, s.size() * 4) != hipSuccess) {
    delete h;
    return fail("mmpl_dit_create", "hipMalloc of RoPE tables failed (is a GPU present?)");
  }
  hipMemcpy(h->cos_tab, c.data(), c.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(h->sin_tab, s.data(), s.size() * 4, hipMemcpyHostToDevice);
  if (mmpl_config().check_share) {
    if (hipMalloc(&h->share_chk, 8 * sizeof(unsigned long long)) != hipSuccess || hipMemset(h->share_chk, 0, 8 * sizeof(unsigned long long)) != hipSuccess) {
      mmpl_dit_destroy(h);
      return fail("mmpl_dit_create", "hipMalloc of the share-check state failed");
    }
  }
  *out = h;
  return 0;
}

void mmpl_dit_destroy(MmplDit* h) {
  if (!h) return;
  if (h->cos_tab) hipFree(h->cos_tab);
  if (h->sin_tab) hipFree(h->sin_tab);
  if (h->share_chk) hipFree(h->share_chk);
  delete h;
}

int mmpl_dit_bind_weights(MmplDit* h, const void* const* ptrs, int n) {
  if (!h || !ptrs) return fail("mmpl_dit_bind_weights", "null argument");
  if (n != NG + h->cfg.num_layers * NL) return fail("mmpl_dit_bind_weights", "wrong pointer count");
  h->w.resize(n);
  for (int i = 0; i < n; ++i) {
    if (!ptrs[i]) return fail("mmpl_dit_bind_weights", "null weight pointer");
    h->w[i] = (const bf16_t*)ptrs[i];
  }
  return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ workspace
namespace {
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* b) : base((char*)b) {}
  bf16_t* take(size_t elems) {
    bf16_t* p = (bf16_t*)(base + off);
    off += (elems * 2 + 255) & ~(size_t)255;
    return p;
  }
};
constexpr int kTicketInts = 64 + 256;     // 8 per-XCD tile tickets (padded to 64 ints) + 256 split-K tile counters: what a forward zeroes
struct FwdWs {
  bf16_t *x, *xn, *big, *attn, *ksc, *vsc, *patch, *sinu, *t1, *e, *se, *e0, *emod, *emod_head, *yh;
  int* tile_counter;     // 8 ints: the per-XCD tile tickets of the large GEMMs (GemmArgs.tile_counter), zeroed at the start of a forward;
                         // ints 64..319: the split-K tile counters (GemmArgs.splitk_cnt)
  float* splitk;         // fp32 partials of the GEMMs' split-K tail launch (mmpl_gemm_splitk_ws_bytes())
  size_t bytes;
};
FwdWs carve_fwd(const MmplDit* h, int nF, void* base) {
  const MmplDitConfig& c = h->cfg;
  const size_t Lq = (size_t)nF * h->S, d = c.dim;
  const size_t bigw = (size_t)(3 * c.dim > c.ffn_dim ? 3 * c.dim : c.ffn_dim);
  Carver k(base);
  FwdWs w;
  // first: the same place for every stage shape, so no other shape's activations ever land on it.  take() counts bf16 elements
  static_assert(kTicketInts * sizeof(int) % sizeof(bf16_t) == 0, "ticket area");
  w.tile_counter = (int*)k.take(kTicketInts * sizeof(int) / sizeof(bf16_t));
  w.x = k.take(Lq * d);
  w.xn = k.take(Lq * d);
  w.big = k.take(Lq * bigw);
  w.attn = k.take(Lq * d);
  w.ksc = k.take(Lq * d);
  w.vsc = k.take(Lq * d);
  w.patch = k.take(Lq * (size_t)h->pe_k);
  w.sinu = k.take((size_t)nF * c.freq_dim);
  w.t1 = k.take((size_t)nF * d);
  w.e = k.take((size_t)nF * d);
  w.se = k.take((size_t)nF * d);
  w.e0 = k.take((size_t)nF * 6 * d);
  w.emod = k.take((size_t)c.num_layers * nF * 6 * d);
  w.emod_head = k.take((size_t)nF * 2 * d);
  w.yh = k.take(Lq * 64);
  w.splitk = (float*)k.take(mmpl_gemm_splitk_ws_bytes() / sizeof(bf16_t));
  w.bytes = k.off;
  return w;
}

int gemm(const bf16_t* A, int lda, const bf16_t* W, int ldw, const bf16_t* bias, bf16_t* C, int ldc, int M, int N, int K,
         int epi, const bf16_t* res, int ldres, const bf16_t* gate, int gfs, int rpf, hipStream_t s, int* tile_counter = nullptr,
         float* splitk_ws = nullptr) {
  GemmArgs g{A, lda, W, ldw, bias, C, ldc, M, N, K, epi, res, ldres, gate, gfs, rpf > 0 ? rpf : 1, 1.0f, 0, 0, 0, 0, 0};
  g.tile_counter = tile_counter;
  g.splitk_ws = splitk_ws;                               // with the forward's scratch: counters 64 ints behind the tile tickets
  g.splitk_cnt = splitk_ws ? tile_counter + 64 : nullptr;
  ProfScope ps(K_GEMM, 2.0 * M * (double)N * K, s);
  hipError_t e = mmpl_launch_gemm(g, s);
  if (e != hipSuccess) return fail("gemm", hipGetErrorString(e));
  return 0;
}
#define TRY(x) do { if ((x) != 0) return 1; } while (0)
}  // namespace

extern "C" {

size_t mmpl_dit_workspace_bytes(const MmplDit* h, int n_frames) { return carve_fwd(h, n_frames, nullptr).bytes; }
size_t mmpl_dit_context_workspace_bytes(const MmplDit* h) {
  return 2 * (((size_t)h->cfg.text_len * h->cfg.dim * 2 + 255) & ~(size_t)255) + (((size_t)h->cfg.text_len * sizeof(int) + 255) & ~(size_t)255);
}

int mmpl_dit_precompute_context(MmplDit* h, const void* context, void* cross_k, void* cross_v, void* workspace,
                                size_t workspace_bytes, int* distinct_rows, mmpl_stream_t stream) {
  if (!h || h->w.empty()) return fail("mmpl_dit_precompute_context", "weights not bound");
  if (workspace_bytes < mmpl_dit_context_workspace_bytes(h)) return fail("mmpl_dit_precompute_context", "workspace too small");
  const MmplDitConfig& c = h->cfg;
  hipStream_t s = (hipStream_t)stream;
  Carver k(workspace);
  bf16_t* t0 = k.take((size_t)c.text_len * c.dim);
  bf16_t* ctx = k.take((size_t)c.text_len * c.dim);
  const int T = c.text_len, d = c.dim;
  TRY(gemm((const bf16_t*)context, c.text_dim, h->G(G_TXT0_W), c.text_dim, h->G(G_TXT0_B), t0, d, T, d, c.text_dim,
           EPI_BIAS_GELU, nullptr, 0, nullptr, 0, 1, s));
  TRY(gemm(t0, d, h->G(G_TXT2_W), d, h->G(G_TXT2_B), ctx, d, T, d, d, EPI_BIAS, nullptr, 0, nullptr, 0, 1, s));
  // How many leading rows of the embedded context are distinct from its last row?  The reference zeroes the padded rows of the
  // T5 output (utils/wan_wrapper.py:46-47) and attends over all text_len of them unmasked (causal_fps_model.py:780, model.py:189):
  // after the text embedding the T - n padded rows are ONE repeated row, so are their K and V rows in every block, and
  // softmax over {k_0..k_{n-1}, (T - n) x k_pad} == softmax over {k_0..k_{n-1}, k_pad + ln(T - n)}: a forward told so (cross_rows)
  // attends over n + 1 keys (the bench's 64-token prompt: 2 KV tiles instead of 8).  Checked on the device per prompt (bitwise
  // row compare of what the K / V projections read); one 2 KiB read-back -- not inside a stream capture, where T is reported.
  // The count is handed to the CALLER: it describes the contents, so it travels with every copy of them (round 4 kept it in
  // the handle keyed by the cross_k pointer, which a re-used address could inherit).
  if (distinct_rows) {
    *distinct_rows = T;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cs);
    if (cs == hipStreamCaptureStatusNone && T >= 2) {
      int* flags = (int*)k.take(((size_t)T * sizeof(int) + 1) / sizeof(bf16_t));
      std::vector<int> hf(T);
      HIP_TRY(mmpl_launch_rows_equal_last(ctx, d, T, d, flags, s), "rows_equal_last");
      HIP_TRY(hipMemcpyAsync(hf.data(), flags, (size_t)T * sizeof(int), hipMemcpyDeviceToHost, s), "context flags");
      HIP_TRY(hipStreamSynchronize(s), "context flags");
      int n = T - 1;
      while (n > 0 && hf[n - 1]) --n;                      // rows n .. T-1 are identical
      if (T - n >= 2) *distinct_rows = n;
    }
  }
  for (int l = 0; l < c.num_layers; ++l) {
    bf16_t* kl = (bf16_t*)cross_k + (size_t)l * T * d;
    bf16_t* vl = (bf16_t*)cross_v + (size_t)l * T * d;
    TRY(gemm(ctx, d, h->Lw(l, L_CK_W), d, h->Lw(l, L_CK_B), kl, d, T, d, d, EPI_BIAS, nullptr, 0, nullptr, 0, 1, s));
    HIP_TRY(mmpl_launch_rmsnorm(kl, d, h->Lw(l, L_CNK), T, d, c.eps, s), "rmsnorm(ctx k)");
    TRY(gemm(ctx, d, h->Lw(l, L_CV_W), d, h->Lw(l, L_CV_B), vl, d, T, d, d, EPI_BIAS, nullptr, 0, nullptr, 0, 1, s));
  }
  return 0;
}

int mmpl_dit_set_image_kv(MmplDit* h, const void* img_k, const void* img_v, int n_img_tokens) {
  if (!h) return fail("mmpl_dit_set_image_kv", "null handle");
  if ((img_k == nullptr) != (img_v == nullptr)) return fail("mmpl_dit_set_image_kv", "need both of img_k / img_v, or neither");
  if (img_k && (n_img_tokens < 1 || n_img_tokens > 4096)) return fail("mmpl_dit_set_image_kv", "n_img_tokens out of range");
  h->img_k = (const bf16_t*)img_k;
  h->img_v = (const bf16_t*)img_v;
  h->n_img = img_k ? n_img_tokens : 0;
  return 0;
}

int mmpl_dit_set_attn_stats(MmplDit* h, void* stats_dev) {
  if (!h) return fail("mmpl_dit_set_attn_stats", "null handle");
  if (reinterpret_cast<uintptr_t>(stats_dev) % 8) return fail("mmpl_dit_set_attn_stats", "stats must be 8-byte aligned");
  h->attn_stats = (unsigned long long*)stats_dev;
  return 0;
}

int mmpl_dit_forward(MmplDit* h, const void* x_in, const float* t_dev, int nF, const int* frame_ids,
                     const int* write_slots, const int* visible_slots, int n_visible, void* k_cache, void* v_cache,
                     int n_slots, const void* cross_k, const void* cross_v, int cross_rows, void* share_out, const void* share_in,
                     void* attn_history, void* out, void* workspace, size_t workspace_bytes, mmpl_stream_t stream) {
  if (!h || h->w.empty()) return fail("mmpl_dit_forward", "weights not bound");
  const MmplDitConfig& c = h->cfg;
  if (nF < 1 || nF > c.max_frames) return fail("mmpl_dit_forward", "n_frames out of range");
  int n_write = 0;
  for (int i = 0; i < nF; ++i) {
    if (write_slots[i] >= n_slots) return fail("mmpl_dit_forward", "write slot out of range");
    if (write_slots[i] >= 0) ++n_write;
  }
  if (n_write != 0 && n_write != nF) return fail("mmpl_dit_forward", "write_slots must be all >= 0 or all -1");
  if (share_out && share_in) return fail("mmpl_dit_forward", "share_out and share_in are exclusive (producer or consumer of block 0's self-attention)");
  const bool persist = n_write == nF;
  const int n_pages = n_visible + (persist ? 0 : nF);
  if (n_pages < 1 || n_pages > MMPL_MAX_PAGES) return fail("mmpl_dit_forward", "too many / no visible KV pages");
  for (int i = 0; i < n_visible; ++i)
    if (visible_slots[i] < 0 || visible_slots[i] >= n_slots) return fail("mmpl_dit_forward", "visible slot out of range");
  FwdWs w = carve_fwd(h, nF, workspace);
  if (workspace_bytes < w.bytes) return fail("mmpl_dit_forward", "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int S = h->S, d = c.dim, f = c.ffn_dim, Lq = nF * S, H = c.num_heads, T = c.text_len;

  HIP_TRY(mmpl_launch_zero_ints(w.tile_counter, kTicketInts, s), "tile counter");                  // left zero by every GEMM that uses it
  int* const tc = w.tile_counter;
  // ---- embeddings (causal_fps_model.py:757-776)
  HIP_TRY(mmpl_launch_patchify((const bf16_t*)x_in, w.patch, h->pe_k, nF, c.in_dim, c.lat_h, c.lat_w, s), "patchify");
  TRY(gemm(w.patch, h->pe_k, h->G(G_PE_W), h->pe_k, h->G(G_PE_B), w.x, d, Lq, d, h->pe_k, EPI_BIAS, nullptr, 0, nullptr, 0, 1, s));
  HIP_TRY(mmpl_launch_sinusoid(t_dev, w.sinu, nF, c.freq_dim, s), "sinusoid");
  TRY(gemm(w.sinu, c.freq_dim, h->G(G_TE0_W), c.freq_dim, h->G(G_TE0_B), w.t1, d, nF, d, c.freq_dim, EPI_BIAS_SILU, nullptr,
           0, nullptr, 0, 1, s));
  TRY(gemm(w.t1, d, h->G(G_TE2_W), d, h->G(G_TE2_B), w.e, d, nF, d, d, EPI_BIAS, nullptr, 0, nullptr, 0, 1, s));
  HIP_TRY(mmpl_launch_silu(w.e, w.se, (size_t)nF * d, s), "silu");
  TRY(gemm(w.se, d, h->G(G_TP_W), d, h->G(G_TP_B), w.e0, 6 * d, nF, 6 * d, d, EPI_BIAS, nullptr, 0, nullptr, 0, 1, s));
  // e = bf16(modulation + e0) for every layer and for the head (causal_fps_model.py:338, 393)
  HIP_TRY(mmpl_launch_modulation(h->G(G_BLOCK_MOD), (size_t)6 * d, w.e0, 6 * d, 0, w.emod, c.num_layers, nF, 6, d, s), "modulation");
  HIP_TRY(mmpl_launch_modulation(h->G(G_HEAD_MOD), 0, w.e, d, 1, w.emod_head, 1, nF, 2, d, s), "head modulation");

  const float scale = 1.0f / sqrtf(128.0f);
  const int self_variant = mmpl_attention_self_variant();
  const bool prescale_q = self_variant == ATTN_W64;         // fold scale * log2(e) into q before it is rounded (kernels.h)
  const bool cross_w64_env = mmpl_config().cross_w64;
  const bool cross_w64 = cross_w64_env && prescale_q;       // text / image cross-attention on the 64-rows-per-wave kernel too
  const size_t layer_stride = (size_t)n_slots * S * d;
  // rows cross_rows .. T-1 of the text K / V are one repeated row (the caller's statement, see the header); -1 = attend over all T
  const int ctx_n = (cross_rows >= 0 && cross_rows <= T - 2) ? cross_rows : -1;
  const size_t hist_layer = mmpl_attn_history_bytes(Lq, H);
  // MMPL_CHECK_SHARE=1: the producer (share_out) and the consumer (share_in) of block 0's shared self-attention fingerprint the layer-0
  // K / V they attend to (after their own slot writes); the consumer compares.  The count of mismatches stays on the device
  // (mmpl_dit_share_check_failures); an eager consumer forward reads it right away and fails.
  const bool check_share = h->share_chk != nullptr && (share_out || share_in);
  auto fingerprint_layer0 = [&](int which) -> int {
    PageList pk{}, pv{};
    for (int i = 0; i < n_visible; ++i) {
      pk.p[i] = (const bf16_t*)k_cache + (size_t)visible_slots[i] * S * d;
      pv.p[i] = (const bf16_t*)v_cache + (size_t)visible_slots[i] * S * d;
    }
    pk.n = pv.n = n_visible;
    if (n_visible < 1) return 0;
    HIP_TRY(mmpl_launch_share_check_zero(h->share_chk, which, s), "share check");
    HIP_TRY(mmpl_launch_pages_fingerprint(pk, (size_t)S * d * sizeof(bf16_t), h->share_chk, which, s), "share check (K)");
    HIP_TRY(mmpl_launch_pages_fingerprint(pv, (size_t)S * d * sizeof(bf16_t), h->share_chk, which, s), "share check (V)");
    if (which == 1) HIP_TRY(mmpl_launch_share_check_compare(h->share_chk, s), "share check (compare)");
    return 0;
  };
  for (int l = 0; l < c.num_layers; ++l) {
    const bf16_t* em = w.emod + (size_t)l * nF * 6 * d;  // [nF][6][d]
    bf16_t* kc = (bf16_t*)k_cache + (size_t)l * layer_stride;
    bf16_t* vc = (bf16_t*)v_cache + (size_t)l * layer_stride;
    // Block 0's self-attention depends on nothing but the latents, the timestep and the cache contents -- none of which differ
    // between the two branches of classifier-free guidance.  share_in: the caller states exactly that for this forward and hands in
    // x as the OTHER branch's forward (share_out) left it after block 0's self-attention residual: the attention and its output
    // projection are skipped (this branch's own K / V slots are still written when the stage persists them).  Same kernels, same
    // inputs -> the bits the skipped launches would have produced.
    const bool take_shared = l == 0 && share_in != nullptr;
    if (!take_shared || persist) {
      // -- self attention (causal_fps_model.py:342-348)
      {
        LnArgs a{w.x, d, w.xn, d, Lq, d, c.eps, em + 1 * d, em + 0 * d, 6 * d, S, nullptr, nullptr};
        ProfScope ps(K_LAYERNORM, 0, s);
        HIP_TRY(mmpl_launch_layernorm(a, s), "norm1");
      }
      {
        // fused q|k|v projection; the V third goes straight into its KV-cache pages (V is cached unchanged,
        // causal_fps_model.py:217), so the norm / RoPE pass below only touches q and k
        GemmArgs g{w.xn, d, h->Lw(l, L_QKV_W), d, h->Lw(l, L_QKV_B), w.big, 3 * d, Lq, 3 * d, d, EPI_BIAS_VPAGES, nullptr, 0, nullptr, 0, S,
                   1.0f, 0, 0, 0, 0, 0};
        for (int i = 0; i < nF; ++i) g.v_dst[i] = persist ? vc + (size_t)write_slots[i] * S * d : w.vsc + (size_t)i * S * d;
        g.v_col0 = 2 * d;
        g.v_ld = d;
        g.tile_counter = tc;
        g.splitk_ws = w.splitk; g.splitk_cnt = tc + 64;
        ProfScope ps(K_GEMM, 2.0 * Lq * 3.0 * d * d, s);
        HIP_TRY(mmpl_launch_gemm(g, s), "qkv gemm");
      }
      {
        QkNormArgs a = {};
        a.q = w.big; a.ldq = 3 * d; a.k = w.big + d; a.ldk = 3 * d; a.v = nullptr; a.ldv = 3 * d;
        a.wq = h->Lw(l, L_NQ); a.wk = h->Lw(l, L_NK); a.rows = Lq; a.d = d; a.eps = c.eps; a.rope = 1;
        a.q_scale = prescale_q ? scale * 1.4426950408889634f : 0.f;
        a.cos_tab = h->cos_tab; a.sin_tab = h->sin_tab; a.rows_per_frame = S; a.grid_w = h->gw;
        for (int i = 0; i < nF; ++i) {
          a.frame_ids[i] = frame_ids[i];
          a.k_dst[i] = persist ? kc + (size_t)write_slots[i] * S * d : w.ksc + (size_t)i * S * d;
          a.v_dst[i] = persist ? vc + (size_t)write_slots[i] * S * d : w.vsc + (size_t)i * S * d;
        }
        ProfScope ps(K_QKNORM, 0, s);
        HIP_TRY(mmpl_launch_qknorm(a, s), "qk norm + rope + kv write");
      }
    }
    if (l == 0 && check_share) TRY(fingerprint_layer0(share_in ? 1 : 0));
    if (take_shared) {
      HIP_TRY(hipMemcpyAsync(w.x, share_in, (size_t)Lq * d * sizeof(bf16_t), hipMemcpyDeviceToDevice, s), "shared block-0 x");
    } else {
    {
        AttnArgs a = {};
        a.q = w.big; a.ldq = 3 * d; a.o = w.attn; a.ldo = d; a.ldk = d; a.ldv = d; a.page_rows = S; a.Lq = Lq; a.H = H; a.scale = scale;
        int np = 0;
        for (int i = 0; i < n_visible; ++i, ++np) {
          a.k_pages[np] = kc + (size_t)visible_slots[i] * S * d;
          a.v_pages[np] = vc + (size_t)visible_slots[i] * S * d;
        }
        if (!persist)
          for (int i = 0; i < nF; ++i, ++np) {
            a.k_pages[np] = w.ksc + (size_t)i * S * d;
            a.v_pages[np] = w.vsc + (size_t)i * S * d;
            a.page_group[np] = 1;                      // the workspace, not the cache allocation (order independent of their addresses)
          }
        a.n_pages = np;
        a.variant = self_variant;
        a.q_prescaled = prescale_q;
        a.split_ws = (float*)w.xn;                    // norm1's output is dead once the QKV GEMM has consumed it
        a.split_ws_bytes = (size_t)Lq * d * sizeof(bf16_t);
        a.redo_stats = h->attn_stats;
        a.history = attn_history ? (unsigned char*)attn_history + (size_t)l * hist_layer : nullptr;
        ProfScope ps(K_ATTN_SELF, 4.0 * Lq * (double)np * S * d, s);
        HIP_TRY(mmpl_launch_attention(a, s), "self attention");
      }
      TRY(gemm(w.attn, d, h->Lw(l, L_O_W), d, h->Lw(l, L_O_B), w.x, d, Lq, d, d, EPI_GATE_RES, w.x, d, em + 2 * d, 6 * d, S, s, tc, w.splitk));
      if (l == 0 && share_out) HIP_TRY(hipMemcpyAsync(share_out, w.x, (size_t)Lq * d * sizeof(bf16_t), hipMemcpyDeviceToDevice, s), "share block-0 x");
    }
    // -- cross attention (causal_fps_model.py:352-353, model.py:161-194)
    {
      LnArgs a{w.x, d, w.xn, d, Lq, d, c.eps, nullptr, nullptr, 0, S, h->Lw(l, L_N3_W), h->Lw(l, L_N3_B)};
      ProfScope ps(K_LAYERNORM, 0, s);
      HIP_TRY(mmpl_launch_layernorm(a, s), "norm3");
    }
    TRY(gemm(w.xn, d, h->Lw(l, L_CQ_W), d, h->Lw(l, L_CQ_B), w.big, d, Lq, d, d, EPI_BIAS, nullptr, 0, nullptr, 0, 1, s, tc, w.splitk));
    {
      ProfScope ps(K_QKNORM, 0, s);
      HIP_TRY(mmpl_launch_rmsnorm(w.big, d, h->Lw(l, L_CNQ), Lq, d, c.eps, s, cross_w64 ? scale * 1.4426950408889634f : 0.f), "cross q norm");
    }
    {
      AttnArgs a = {};
      a.q = w.big; a.ldq = d; a.o = w.attn; a.ldo = d; a.ldk = d; a.ldv = d; a.page_rows = T; a.Lq = Lq; a.H = H; a.scale = scale;
      a.n_pages = 1;
      a.cross = 1;
      if (cross_w64) { a.variant = ATTN_W64; a.q_prescaled = true; }
      else if (ctx_n >= 0) { a.page_rows = ctx_n + 1; a.last_row_copies = T - ctx_n; }   // padded tail = one key, weighted (precompute_context)
      a.k_pages[0] = (const bf16_t*)cross_k + (size_t)l * T * d;
      a.v_pages[0] = (const bf16_t*)cross_v + (size_t)l * T * d;
      ProfScope ps(K_ATTN_CROSS, 4.0 * Lq * (double)T * d, s);
      HIP_TRY(mmpl_launch_attention(a, s), "cross attention");
      if (h->img_k) {
        // Wan-I2V: the query also attends to the image tokens; the two outputs are summed in bf16 before the o-projection
        // (WanI2VCrossAttention.forward, model.py:254-263).  The scratch K pages are free here: they hold the image output.
        a.o = w.ksc;
        a.page_rows = h->n_img;
        a.last_row_copies = 0;                    // (the text stream's weighted padded key does not apply to the image tokens)
        a.k_pages[0] = h->img_k + (size_t)l * h->n_img * d;
        a.v_pages[0] = h->img_v + (size_t)l * h->n_img * d;
        ProfScope ps2(K_ATTN_CROSS, 4.0 * Lq * (double)h->n_img * d, s);
        HIP_TRY(mmpl_launch_attention(a, s), "image cross attention");
        HIP_TRY(mmpl_launch_add(w.attn, w.ksc, (size_t)Lq * d, s), "x + img_x");
      }
    }
    TRY(gemm(w.attn, d, h->Lw(l, L_CO_W), d, h->Lw(l, L_CO_B), w.x, d, Lq, d, d, EPI_RES, w.x, d, nullptr, 0, 1, s, tc, w.splitk));
    // -- FFN (causal_fps_model.py:354-360)
    {
      LnArgs a{w.x, d, w.xn, d, Lq, d, c.eps, em + 4 * d, em + 3 * d, 6 * d, S, nullptr, nullptr};
      ProfScope ps(K_LAYERNORM, 0, s);
      HIP_TRY(mmpl_launch_layernorm(a, s), "norm2");
    }
    TRY(gemm(w.xn, d, h->Lw(l, L_F0_W), d, h->Lw(l, L_F0_B), w.big, f, Lq, f, d, EPI_BIAS_GELU, nullptr, 0, nullptr, 0, 1, s, tc, w.splitk));
    TRY(gemm(w.big, f, h->Lw(l, L_F2_W), f, h->Lw(l, L_F2_B), w.x, d, Lq, d, f, EPI_GATE_RES, w.x, d, em + 5 * d, 6 * d, S, s, tc, w.splitk));
  }
  // ---- head + unpatchify (causal_fps_model.py:384-395, 1007-1030)
  {
    LnArgs a{w.x, d, w.xn, d, Lq, d, c.eps, w.emod_head + d, w.emod_head, 2 * d, S, nullptr, nullptr};
    HIP_TRY(mmpl_launch_layernorm(a, s), "head norm");
  }
  TRY(gemm(w.xn, d, h->G(G_HEAD_W), d, h->G(G_HEAD_B), w.yh, 64, Lq, 64, d, EPI_BIAS, nullptr, 0, nullptr, 0, 1, s));
  HIP_TRY(mmpl_launch_unpatchify(w.yh, 64, (bf16_t*)out, nF, c.out_dim, c.lat_h, c.lat_w, s), "unpatchify");
  if (check_share && share_in) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cs);
    if (cs == hipStreamCaptureStatusNone) {                  // eager: fail this very call (inside a capture the count stays on the device)
      long long n = 0;
      TRY(mmpl_dit_share_check_failures(h, &n, stream));
      if (n) return fail("mmpl_dit_forward", "MMPL_CHECK_SHARE: share_in was given but the layer-0 K / V this forward attends to differ from "
                                             "the share_out forward's (the two CFG caches are out of step)");
    }
  }
  return 0;
}

int mmpl_dit_share_check_failures(MmplDit* h, long long* count, mmpl_stream_t stream) {
  if (!h || !count) return fail("mmpl_dit_share_check_failures", "null argument");
  *count = 0;
  if (!h->share_chk) return 0;                               // MMPL_CHECK_SHARE not set when the handle was created: nothing is checked
  unsigned long long n = 0;
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipMemcpyAsync(&n, h->share_chk + 4, sizeof(n), hipMemcpyDeviceToHost, s), "share check read");
  HIP_TRY(hipStreamSynchronize(s), "share check read");
  if (n) HIP_TRY(hipMemsetAsync(h->share_chk + 4, 0, sizeof(n), s), "share check reset");
  *count = (long long)n;
  return 0;
}

size_t mmpl_dit_attn_history_bytes(const MmplDit* h, int n_frames) {
  return h ? (size_t)h->cfg.num_layers * mmpl_attn_history_bytes(n_frames * h->S, h->cfg.num_heads) : 0;
}
size_t mmpl_attn_history_bytes(int Lq, int num_heads) { return mmpl_attention_history_bytes(Lq, num_heads); }

int mmpl_attn_fwd_history(const void* q, int ldq, void* o, int ldo, const void* const* k_pages, const void* const* v_pages,
                          int ldk, int ldv, int n_pages, int page_rows, int Lq, int num_heads, float softmax_scale,
                          void* workspace, size_t workspace_bytes, void* history, void* stats_dev, mmpl_stream_t stream) {
  if (n_pages < 1 || n_pages > MMPL_MAX_PAGES) return fail("mmpl_attn_fwd_history", "n_pages out of range");
  if (reinterpret_cast<uintptr_t>(stats_dev) % 8) return fail("mmpl_attn_fwd_history", "stats must be 8-byte aligned");
  AttnArgs a = {};
  a.variant = ATTN_W64;
  a.q_prescaled = 1;
  a.q = (const bf16_t*)q; a.ldq = ldq; a.o = (bf16_t*)o; a.ldo = ldo; a.ldk = ldk; a.ldv = ldv; a.n_pages = n_pages;
  a.page_rows = page_rows; a.Lq = Lq; a.H = num_heads; a.scale = softmax_scale;
  a.split_ws = (float*)workspace; a.split_ws_bytes = workspace ? workspace_bytes : 0;
  a.history = (unsigned char*)history;
  a.redo_stats = (unsigned long long*)stats_dev;
  for (int i = 0; i < n_pages; ++i) { a.k_pages[i] = (const bf16_t*)k_pages[i]; a.v_pages[i] = (const bf16_t*)v_pages[i]; }
  HIP_TRY(mmpl_launch_attention(a, (hipStream_t)stream), "mmpl_attn_fwd_history");
  return 0;
}

// ------------------------------------------------------------------------------------------------ single kernels
size_t mmpl_attn_workspace_bytes(void) { return mmpl_attention_split_ws_bytes(); }

int mmpl_attn_fwd_variant(const void* q, int ldq, void* o, int ldo, const void* const* k_pages, const void* const* v_pages,
                          int ldk, int ldv, int n_pages, int page_rows, int Lq, int num_heads, float softmax_scale,
                          void* workspace, size_t workspace_bytes, int variant, int cross, mmpl_stream_t stream) {
  if (n_pages < 1 || n_pages > MMPL_MAX_PAGES) return fail("mmpl_attn_fwd", "n_pages out of range");
  if (variant != ATTN_AUTO && variant != ATTN_LOCKSTEP && variant != ATTN_W64 && variant != ATTN_W64 + 1)
    return fail("mmpl_attn_fwd", "unknown kernel variant");
  AttnArgs a = {};
  a.variant = variant > ATTN_W64 ? ATTN_W64 : variant;
  a.q_prescaled = variant == ATTN_W64 + 1;
  a.cross = cross != 0;
  a.q = (const bf16_t*)q; a.ldq = ldq; a.o = (bf16_t*)o; a.ldo = ldo; a.ldk = ldk; a.ldv = ldv; a.n_pages = n_pages;
  a.page_rows = page_rows; a.Lq = Lq; a.H = num_heads; a.scale = softmax_scale;
  a.split_ws = (float*)workspace; a.split_ws_bytes = workspace ? workspace_bytes : 0;
  for (int i = 0; i < n_pages; ++i) { a.k_pages[i] = (const bf16_t*)k_pages[i]; a.v_pages[i] = (const bf16_t*)v_pages[i]; }
  HIP_TRY(mmpl_launch_attention(a, (hipStream_t)stream), "mmpl_attn_fwd");
  return 0;
}

int mmpl_attn_fwd_ws(const void* q, int ldq, void* o, int ldo, const void* const* k_pages, const void* const* v_pages,
                     int ldk, int ldv, int n_pages, int page_rows, int Lq, int num_heads, float softmax_scale,
                     void* workspace, size_t workspace_bytes, mmpl_stream_t stream) {
  return mmpl_attn_fwd_variant(q, ldq, o, ldo, k_pages, v_pages, ldk, ldv, n_pages, page_rows, Lq, num_heads, softmax_scale, workspace,
                               workspace_bytes, ATTN_AUTO, 0, stream);
}

int mmpl_attn_fwd(const void* q, int ldq, void* o, int ldo, const void* const* k_pages, const void* const* v_pages,
                  int ldk, int ldv, int n_pages, int page_rows, int Lq, int num_heads, float softmax_scale,
                  mmpl_stream_t stream) {
  return mmpl_attn_fwd_ws(q, ldq, o, ldo, k_pages, v_pages, ldk, ldv, n_pages, page_rows, Lq, num_heads, softmax_scale, nullptr, 0,
                          stream);
}

int mmpl_gemm(const void* A, int lda, const void* W, int ldw, const void* bias, void* C, int ldc, int M, int N, int K,
              int epi, const void* res, int ldres, const void* gate, int gate_frame_stride, int rows_per_frame,
              mmpl_stream_t stream) {
  if (epi < EPI_BIAS || epi > EPI_F32_SCALE) return fail("mmpl_gemm", "unknown epilogue");
  if ((epi == EPI_GATE_RES && (!res || !gate)) || (epi == EPI_RES && !res)) return fail("mmpl_gemm", "missing epilogue operand");
  return gemm((const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)bias, (bf16_t*)C, ldc, M, N, K, epi,
              (const bf16_t*)res, ldres, (const bf16_t*)gate, gate_frame_stride, rows_per_frame, (hipStream_t)stream);
}

int mmpl_gemm_tickets(const void* A, int lda, const void* W, int ldw, const void* bias, void* C, int ldc, int M, int N, int K,
                      int epi, const void* res, int ldres, const void* gate, int gate_frame_stride, int rows_per_frame,
                      void* tile_counter, mmpl_stream_t stream) {
  if (epi < EPI_BIAS || epi > EPI_F32_SCALE) return fail("mmpl_gemm", "unknown epilogue");
  if ((epi == EPI_GATE_RES && (!res || !gate)) || (epi == EPI_RES && !res)) return fail("mmpl_gemm", "missing epilogue operand");
  return gemm((const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)bias, (bf16_t*)C, ldc, M, N, K, epi,
              (const bf16_t*)res, ldres, (const bf16_t*)gate, gate_frame_stride, rows_per_frame, (hipStream_t)stream, (int*)tile_counter);
}

size_t mmpl_gemm_scratch_bytes(void) { return 2048 + mmpl_gemm_splitk_ws_bytes(); }
int mmpl_device_xcd_round_robin(void) { return mmpl_xcd_dispatch_ok(true) ? 1 : 0; }

int mmpl_gemm_scratch(const void* A, int lda, const void* W, int ldw, const void* bias, void* C, int ldc, int M, int N, int K,
                      int epi, const void* res, int ldres, const void* gate, int gate_frame_stride, int rows_per_frame,
                      void* scratch, size_t scratch_bytes, mmpl_stream_t stream) {
  if (epi < EPI_BIAS || epi > EPI_F32_SCALE) return fail("mmpl_gemm", "unknown epilogue");
  if ((epi == EPI_GATE_RES && (!res || !gate)) || (epi == EPI_RES && !res)) return fail("mmpl_gemm", "missing epilogue operand");
  if (!scratch || scratch_bytes < mmpl_gemm_scratch_bytes()) return fail("mmpl_gemm_scratch", "scratch missing or smaller than mmpl_gemm_scratch_bytes()");
  if (reinterpret_cast<uintptr_t>(scratch) % 256) return fail("mmpl_gemm_scratch", "scratch must be 256-byte aligned");
  return gemm((const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)bias, (bf16_t*)C, ldc, M, N, K, epi,
              (const bf16_t*)res, ldres, (const bf16_t*)gate, gate_frame_stride, rows_per_frame, (hipStream_t)stream, (int*)scratch,
              (float*)((char*)scratch + 2048));
}

int mmpl_layernorm(const void* x, int ldx, void* y, int ldy, int rows, int d, float eps, const void* scale,
                   const void* shift, int mod_frame_stride, int rows_per_frame, const void* w, const void* b,
                   mmpl_stream_t stream) {
  if (!w && (!scale || !shift)) return fail("mmpl_layernorm", "need (scale, shift) or (w, b)");
  LnArgs a{(const bf16_t*)x, ldx, (bf16_t*)y, ldy, rows, d, eps, (const bf16_t*)scale, (const bf16_t*)shift, mod_frame_stride,
           rows_per_frame > 0 ? rows_per_frame : 1, (const bf16_t*)w, (const bf16_t*)b};
  HIP_TRY(mmpl_launch_layernorm(a, (hipStream_t)stream), "mmpl_layernorm");
  return 0;
}

int mmpl_qknorm_rope(MmplDit* h, void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* wq,
                     const void* wk, int n_frames, const int* frame_ids, void* const* k_dst, void* const* v_dst,
                     mmpl_stream_t stream) {
  if (!h) return fail("mmpl_qknorm_rope", "null handle");
  if (n_frames < 1 || n_frames > 8) return fail("mmpl_qknorm_rope", "n_frames out of range");
  QkNormArgs a = {};
  a.q = (bf16_t*)q; a.ldq = ldq; a.k = (const bf16_t*)k; a.ldk = ldk; a.v = (const bf16_t*)v; a.ldv = ldv;
  a.wq = (const bf16_t*)wq; a.wk = (const bf16_t*)wk; a.rows = n_frames * h->S; a.d = h->cfg.dim; a.eps = h->cfg.eps; a.rope = 1;
  a.cos_tab = h->cos_tab; a.sin_tab = h->sin_tab; a.rows_per_frame = h->S; a.grid_w = h->gw;
  for (int i = 0; i < n_frames; ++i) { a.frame_ids[i] = frame_ids[i]; a.k_dst[i] = (bf16_t*)k_dst[i]; a.v_dst[i] = (bf16_t*)v_dst[i]; }
  HIP_TRY(mmpl_launch_qknorm(a, (hipStream_t)stream), "mmpl_qknorm_rope");
  return 0;
}

int mmpl_cfg_unipc_step(const void* flow_cond, const void* flow_uncond, void* x, void* m0, void* m1, void* last_sample,
                        size_t n, const MmplUniPCStep* st, mmpl_stream_t stream) {
  if (!st) return fail("mmpl_cfg_unipc_step", "null step");
  UniPCArgs a = {};
  a.flow_c = (const bf16_t*)flow_cond; a.flow_u = (const bf16_t*)flow_uncond; a.guidance = st->guidance; a.x = (bf16_t*)x;
  a.m0 = (bf16_t*)m0; a.m1 = (bf16_t*)m1; a.last_sample = (bf16_t*)last_sample; a.n = n; a.sigma_cur = st->sigma_cur;
  a.use_corrector = st->use_corrector; a.corr_order = st->corr_order; a.c_c1 = st->c_c1; a.c_c2 = st->c_c2; a.c_c3 = st->c_c3;
  a.c_inv_rk = st->c_inv_rk; a.c_rho0 = st->c_rho0; a.c_rho_last = st->c_rho_last; a.pred_order = st->pred_order;
  a.p_c1 = st->p_c1; a.p_c2 = st->p_c2; a.p_c3 = st->p_c3; a.p_inv_rk = st->p_inv_rk;
  ProfScope ps(K_UNIPC, 0, (hipStream_t)stream);
  HIP_TRY(mmpl_launch_unipc(a, (hipStream_t)stream), "mmpl_cfg_unipc_step");
  return 0;
}

int mmpl_cfg_unipc_step_table(const void* flow_cond, const void* flow_uncond, void* x, void* m0, void* m1, void* last_sample,
                              size_t n, const MmplUniPCStep* table_dev, int* step_dev, float* timestep_dev,
                              const float* timestep_table_dev, int n_timestep, int n_steps, mmpl_stream_t stream) {
  if (!table_dev || !step_dev || !timestep_dev || !timestep_table_dev || n_steps < 1 || n_timestep < 1)
    return fail("mmpl_cfg_unipc_step_table", "bad arguments");
  static_assert(sizeof(MmplUniPCStep) == sizeof(UniPCStepDev), "table layout");
  UniPCArgs a = {};
  a.flow_c = (const bf16_t*)flow_cond; a.flow_u = (const bf16_t*)flow_uncond; a.x = (bf16_t*)x;
  a.m0 = (bf16_t*)m0; a.m1 = (bf16_t*)m1; a.last_sample = (bf16_t*)last_sample; a.n = n;
  HIP_TRY(mmpl_launch_unipc_table(a, (const UniPCStepDev*)table_dev, step_dev, timestep_dev, timestep_table_dev, n_timestep, n_steps,
                                  (hipStream_t)stream), "mmpl_cfg_unipc_step_table");
  return 0;
}

}  // extern "C"
