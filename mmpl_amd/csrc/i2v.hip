// Wan-I2V image conditioning of the text cross-attention: MLPProj over the CLIP features
// (MMPL_t2v/wan/modules/model.py:469-481) and WanI2VCrossAttention (model.py:224-266): the query attends separately to
// the 257 projected image tokens and to the 512 text tokens, the two outputs are summed before the o-projection.
// Host orchestration over the DiT kernels (gemm.hip, attention.hip, elementwise.hip) + two elementwise kernels; C ABI
// mmpl_i2v_*.  The FPS wrapper of the reference never instantiates this module (its I2V mode conditions through the
// VAE-encoded first frame only), so nothing in the denoising loop calls it; parity is against the module run standalone.
#include <math.h>

#include "../../include/mmpl_hip.h"
#include "kernels.h"

extern int mmpl_set_error(const char* where, const char* what);  // api.hip

namespace {

// torch.nn.GELU() (erf form) on a bf16 tensor: fp32 math, one rounding
__global__ void gelu_erf_kernel(bf16_t* x, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = bf2f(x[i]);
    x[i] = f2bf(0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)));
  }
}
// a = bf16(a + b)   (x = x + img_x, model.py:262)
__global__ void add_kernel(bf16_t* a, const bf16_t* b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    a[i] = f2bf(bf2f(a[i]) + bf2f(b[i]));
}

hipError_t linear(const bf16_t* x, int K, const bf16_t* w, const bf16_t* b, bf16_t* y, int M, int N, hipStream_t s) {
  GemmArgs g = {x, K, w, K, b, y, N, M, N, K, EPI_BIAS, nullptr, 0, nullptr, 0, 1, 1.0f, 0, 0, 0, 0, 0};
  return mmpl_launch_gemm(g, s);
}
hipError_t attend(const bf16_t* q, bf16_t* o, const bf16_t* k, const bf16_t* v, int n_kv, int Lq, int dim, hipStream_t s) {
  AttnArgs a = {};
  a.q = q; a.ldq = dim; a.o = o; a.ldo = dim;
  a.k_pages[0] = k; a.v_pages[0] = v; a.ldk = dim; a.ldv = dim;
  a.n_pages = 1; a.page_rows = n_kv; a.Lq = Lq; a.H = dim / 128; a.scale = 1.0f / sqrtf(128.0f); a.cross = 1;
  return mmpl_launch_attention(a, s);
}
int blocks_for(size_t n) { return (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256); }

#define I2V_TRY(expr, where)                                                  \
  do {                                                                        \
    hipError_t e_ = (expr);                                                   \
    if (e_ != hipSuccess) return mmpl_set_error(where, hipGetErrorString(e_)); \
  } while (0)

}  // namespace

hipError_t mmpl_launch_add(bf16_t* a, const bf16_t* b, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(add_kernel, dim3(blocks_for(n)), dim3(256), 0, s, a, b, n);
  return hipGetLastError();
}

extern "C" {

size_t mmpl_i2v_img_proj_workspace_bytes(int n_tok, int clip_dim, int dim) {
  return ((size_t)2 * n_tok * clip_dim + (size_t)n_tok * dim) * sizeof(bf16_t) + 1024;
}

int mmpl_i2v_img_proj(const void* clip_fea, int n_tok, int clip_dim, int dim, const void* const* w, void* out, void* workspace,
                      size_t workspace_bytes, mmpl_stream_t stream) {
  if (!clip_fea || !w || !out || !workspace) return mmpl_set_error("mmpl_i2v_img_proj", "null argument");
  if (n_tok < 1 || clip_dim % 128 || dim % 128 || clip_dim > 5120 || dim > 5120) return mmpl_set_error("mmpl_i2v_img_proj", "unsupported dims");
  if (workspace_bytes < mmpl_i2v_img_proj_workspace_bytes(n_tok, clip_dim, dim)) return mmpl_set_error("mmpl_i2v_img_proj", "workspace too small");
  for (int i = 0; i < 8; ++i)
    if (!w[i]) return mmpl_set_error("mmpl_i2v_img_proj", "null weight pointer");
  hipStream_t s = (hipStream_t)stream;
  bf16_t* t0 = (bf16_t*)workspace;
  bf16_t* t1 = t0 + (size_t)n_tok * clip_dim;
  bf16_t* t2 = t1 + (size_t)n_tok * clip_dim;
  const bf16_t* const* W = (const bf16_t* const*)w;   // proj.0.{weight,bias}, proj.1.{weight,bias}, proj.3.{weight,bias}, proj.4.{weight,bias}
  LnArgs l0 = {(const bf16_t*)clip_fea, clip_dim, t0, clip_dim, n_tok, clip_dim, 1e-5f, nullptr, nullptr, 0, 1, W[0], W[1]};
  I2V_TRY(mmpl_launch_layernorm(l0, s), "mmpl_i2v_img_proj: layernorm 0");
  I2V_TRY(linear(t0, clip_dim, W[2], W[3], t1, n_tok, clip_dim, s), "mmpl_i2v_img_proj: fc1");
  gelu_erf_kernel<<<blocks_for((size_t)n_tok * clip_dim), 256, 0, s>>>(t1, (size_t)n_tok * clip_dim);
  I2V_TRY(linear(t1, clip_dim, W[4], W[5], t2, n_tok, dim, s), "mmpl_i2v_img_proj: fc2");
  LnArgs l1 = {t2, dim, (bf16_t*)out, dim, n_tok, dim, 1e-5f, nullptr, nullptr, 0, 1, W[6], W[7]};
  I2V_TRY(mmpl_launch_layernorm(l1, s), "mmpl_i2v_img_proj: layernorm 1");
  I2V_TRY(hipGetLastError(), "mmpl_i2v_img_proj");
  return 0;
}

int mmpl_i2v_img_kv(const void* ctx_img, int n_tok, int dim, const void* wk, const void* bk, const void* wv, const void* bv,
                    const void* norm_k_img_w, float eps, void* k_out, void* v_out, mmpl_stream_t stream) {
  if (!ctx_img || !wk || !bk || !wv || !bv || !norm_k_img_w || !k_out || !v_out) return mmpl_set_error("mmpl_i2v_img_kv", "null argument");
  if (n_tok < 1 || dim % 128 || dim > 5120) return mmpl_set_error("mmpl_i2v_img_kv", "unsupported dims");
  hipStream_t s = (hipStream_t)stream;
  I2V_TRY(linear((const bf16_t*)ctx_img, dim, (const bf16_t*)wk, (const bf16_t*)bk, (bf16_t*)k_out, n_tok, dim, s), "mmpl_i2v_img_kv: k_img");
  I2V_TRY(mmpl_launch_rmsnorm((bf16_t*)k_out, dim, (const bf16_t*)norm_k_img_w, n_tok, dim, eps, s), "mmpl_i2v_img_kv: norm_k_img");
  I2V_TRY(linear((const bf16_t*)ctx_img, dim, (const bf16_t*)wv, (const bf16_t*)bv, (bf16_t*)v_out, n_tok, dim, s), "mmpl_i2v_img_kv: v_img");
  return 0;
}

size_t mmpl_i2v_cross_attn_workspace_bytes(int Lq, int dim) { return (size_t)3 * Lq * dim * sizeof(bf16_t) + 1024; }

int mmpl_i2v_cross_attn(const void* x, int Lq, int dim, const void* wq, const void* bq, const void* norm_q_w, float eps,
                        const void* k_txt, const void* v_txt, int n_txt, const void* k_img, const void* v_img, int n_img,
                        const void* wo, const void* bo, void* out, void* workspace, size_t workspace_bytes, mmpl_stream_t stream) {
  if (!x || !wq || !bq || !norm_q_w || !k_txt || !v_txt || !k_img || !v_img || !wo || !bo || !out || !workspace)
    return mmpl_set_error("mmpl_i2v_cross_attn", "null argument");
  if (Lq < 1 || n_txt < 1 || n_img < 1 || dim % 128 || dim > 5120) return mmpl_set_error("mmpl_i2v_cross_attn", "unsupported dims");
  if (workspace_bytes < mmpl_i2v_cross_attn_workspace_bytes(Lq, dim)) return mmpl_set_error("mmpl_i2v_cross_attn", "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  bf16_t* q = (bf16_t*)workspace;
  bf16_t* at = q + (size_t)Lq * dim;
  bf16_t* ai = at + (size_t)Lq * dim;
  I2V_TRY(linear((const bf16_t*)x, dim, (const bf16_t*)wq, (const bf16_t*)bq, q, Lq, dim, s), "mmpl_i2v_cross_attn: q");
  I2V_TRY(mmpl_launch_rmsnorm(q, dim, (const bf16_t*)norm_q_w, Lq, dim, eps, s), "mmpl_i2v_cross_attn: norm_q");
  I2V_TRY(attend(q, ai, (const bf16_t*)k_img, (const bf16_t*)v_img, n_img, Lq, dim, s), "mmpl_i2v_cross_attn: image attention");
  I2V_TRY(attend(q, at, (const bf16_t*)k_txt, (const bf16_t*)v_txt, n_txt, Lq, dim, s), "mmpl_i2v_cross_attn: text attention");
  add_kernel<<<blocks_for((size_t)Lq * dim), 256, 0, s>>>(at, ai, (size_t)Lq * dim);
  I2V_TRY(hipGetLastError(), "mmpl_i2v_cross_attn: x + img_x");
  I2V_TRY(linear(at, dim, (const bf16_t*)wo, (const bf16_t*)bo, (bf16_t*)out, Lq, dim, s), "mmpl_i2v_cross_attn: o");
  I2V_TRY(hipGetLastError(), "mmpl_i2v_cross_attn");
  return 0;
}

// ---------------------------------------------------------------- CLIP ViT-H/14 vision tower (Wan-I2V's image encoder)
// VisionTransformer.forward(x, use_31_block=True) (MMPL_t2v/wan/modules/clip.py:209-327, called by CLIPModel.visual :527-542):
// patch embedding (no bias: pre_norm) -> [cls; patches] + pos -> pre_norm -> n_blocks pre-norm blocks (clip.py:120-155: LayerNorm,
// fused qkv, attention, proj, LayerNorm, Linear - GELU(erf) - Linear) -> the tokens themselves (no post_norm, no head).
// The real tower has 16 heads of 80: every head is padded to 128 columns by the host when it packs the weights (zero rows in
// to_qkv, zero columns in proj), so the DiT's 128-wide attention kernel serves it unchanged with softmax_scale = 1/sqrt(80).
// patches: dev [n_patch, pk] = im2col of the normalised image, (c, ky, kx) order, K zero-padded to a multiple of 64.
// gw: patch_embedding.weight [dim, pk], cls_embedding [dim], pos_embedding [n_patch + 1, dim], pre_norm.{weight, bias}.
// lw: per block 12 pointers: norm1.{weight,bias}, to_qkv.{weight [3 H 128, dim], bias}, proj.{weight [dim, H 128], bias},
//     norm2.{weight,bias}, mlp.0.{weight,bias}, mlp.2.{weight,bias}.
size_t mmpl_clip_visual_workspace_bytes(int n_tok, int dim, int mlp_dim, int heads) {
  const size_t wide = (size_t)(3 * heads * 128 > mlp_dim ? 3 * heads * 128 : mlp_dim);
  return ((size_t)n_tok * dim * 2 + (size_t)n_tok * wide + (size_t)n_tok * heads * 128) * sizeof(bf16_t) + 4096;
}

int mmpl_clip_visual(const void* patches, int n_patch, int pk, int dim, int mlp_dim, int heads, int head_dim, int n_blocks,
                     const void* const* gw, const void* const* lw, float eps, void* out, void* workspace, size_t workspace_bytes,
                     mmpl_stream_t stream) {
  if (!patches || !gw || !lw || !out || !workspace) return mmpl_set_error("mmpl_clip_visual", "null argument");
  if (n_patch < 1 || pk % 64 || dim % 64 || mlp_dim % 64 || dim > 5120 || heads < 1 || head_dim < 1 || head_dim > 128 || n_blocks < 0)
    return mmpl_set_error("mmpl_clip_visual", "unsupported dims");
  const int n = n_patch + 1, hw = heads * 128;
  if (workspace_bytes < mmpl_clip_visual_workspace_bytes(n, dim, mlp_dim, heads)) return mmpl_set_error("mmpl_clip_visual", "workspace too small");
  for (int i = 0; i < 5; ++i)
    if (!gw[i]) return mmpl_set_error("mmpl_clip_visual", "null weight pointer");
  for (int i = 0; i < 12 * n_blocks; ++i)
    if (!lw[i]) return mmpl_set_error("mmpl_clip_visual", "null weight pointer");
  hipStream_t s = (hipStream_t)stream;
  const bf16_t* const* G = (const bf16_t* const*)gw;
  const bf16_t* const* L = (const bf16_t* const*)lw;
  const size_t wide = (size_t)(3 * hw > mlp_dim ? 3 * hw : mlp_dim);
  auto al = [](size_t e) { return (e + 127) & ~(size_t)127; };
  bf16_t* x = (bf16_t*)workspace;                 // [n, dim] the residual stream
  bf16_t* hbuf = x + al((size_t)n * dim);         // [n, dim] LayerNorm output
  bf16_t* big = hbuf + al((size_t)n * dim);       // [n, wide] qkv | mlp hidden
  bf16_t* att = big + al((size_t)n * wide);       // [n, H 128]
  // x = [cls; patches . W^T] + pos
  I2V_TRY(hipMemcpyAsync(x, G[1], (size_t)dim * sizeof(bf16_t), hipMemcpyDeviceToDevice, s), "mmpl_clip_visual: cls");
  {
    GemmArgs g = {(const bf16_t*)patches, pk, G[0], pk, nullptr, x + dim, dim, n_patch, dim, pk, EPI_BIAS, nullptr, 0, nullptr, 0, 1, 1.0f, 0, 0, 0, 0, 0};
    I2V_TRY(mmpl_launch_gemm(g, s), "mmpl_clip_visual: patch embedding");
  }
  add_kernel<<<blocks_for((size_t)n * dim), 256, 0, s>>>(x, G[2], (size_t)n * dim);
  I2V_TRY(hipGetLastError(), "mmpl_clip_visual: + pos_embedding");
  {
    LnArgs l = {x, dim, hbuf, dim, n, dim, eps, nullptr, nullptr, 0, 1, G[3], G[4]};
    I2V_TRY(mmpl_launch_layernorm(l, s), "mmpl_clip_visual: pre_norm");
    I2V_TRY(hipMemcpyAsync(x, hbuf, (size_t)n * dim * sizeof(bf16_t), hipMemcpyDeviceToDevice, s), "mmpl_clip_visual: pre_norm copy");
  }
  for (int b = 0; b < n_blocks; ++b) {
    const bf16_t* const* W = L + 12 * b;
    LnArgs l1 = {x, dim, hbuf, dim, n, dim, eps, nullptr, nullptr, 0, 1, W[0], W[1]};
    I2V_TRY(mmpl_launch_layernorm(l1, s), "mmpl_clip_visual: norm1");
    I2V_TRY(linear(hbuf, dim, W[2], W[3], big, n, 3 * hw, s), "mmpl_clip_visual: to_qkv");
    {
      AttnArgs a = {};
      a.q = big; a.ldq = 3 * hw; a.o = att; a.ldo = hw;
      a.k_pages[0] = big + hw; a.v_pages[0] = big + 2 * hw; a.ldk = 3 * hw; a.ldv = 3 * hw;
      a.n_pages = 1; a.page_rows = n; a.Lq = n; a.H = heads; a.scale = 1.0f / sqrtf((float)head_dim); a.cross = 1;
      I2V_TRY(mmpl_launch_attention(a, s), "mmpl_clip_visual: attention");
    }
    {
      GemmArgs g = {att, hw, W[4], hw, W[5], x, dim, n, dim, hw, EPI_RES, x, dim, nullptr, 0, 1, 1.0f, 0, 0, 0, 0, 0};
      I2V_TRY(mmpl_launch_gemm(g, s), "mmpl_clip_visual: proj + residual");
    }
    LnArgs l2 = {x, dim, hbuf, dim, n, dim, eps, nullptr, nullptr, 0, 1, W[6], W[7]};
    I2V_TRY(mmpl_launch_layernorm(l2, s), "mmpl_clip_visual: norm2");
    I2V_TRY(linear(hbuf, dim, W[8], W[9], big, n, mlp_dim, s), "mmpl_clip_visual: mlp.0");
    gelu_erf_kernel<<<blocks_for((size_t)n * mlp_dim), 256, 0, s>>>(big, (size_t)n * mlp_dim);
    I2V_TRY(hipGetLastError(), "mmpl_clip_visual: gelu");
    {
      GemmArgs g = {big, mlp_dim, W[10], mlp_dim, W[11], x, dim, n, dim, mlp_dim, EPI_RES, x, dim, nullptr, 0, 1, 1.0f, 0, 0, 0, 0, 0};
      I2V_TRY(mmpl_launch_gemm(g, s), "mmpl_clip_visual: mlp.2 + residual");
    }
  }
  I2V_TRY(hipMemcpyAsync(out, x, (size_t)n * dim * sizeof(bf16_t), hipMemcpyDeviceToDevice, s), "mmpl_clip_visual: out");
  I2V_TRY(hipGetLastError(), "mmpl_clip_visual");
  return 0;
}

}  // extern "C"
