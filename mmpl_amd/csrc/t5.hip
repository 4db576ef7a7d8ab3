// umT5 encoder (the text encoder in front of the denoising path): kernels + host orchestration + C ABI (mmpl_t5_*).
// Follows MMPL_t2v/wan/modules/t5.py (T5Encoder :267-312 with per-layer relative position bias, T5Attention :70-121,
// T5FeedForward :124-145, T5LayerNorm :52-67) behind WanTextEncoder (utils/wan_wrapper.py:15-51).  The linears run on
// the DiT MFMA GEMM kernels (gemm.hip; the 64 heads of QK^T and PV as one batched launch each); the kernels here are
// the HBM-bound glue with the reference's bf16 rounding points.
#include <math.h>

#include <vector>

#include "../../include/mmpl_hip.h"
#include "kernels.h"

extern int mmpl_set_error(const char* where, const char* what);  // api.hip

namespace {

__global__ void t5_gather_kernel(const int* ids, const bf16_t* emb, bf16_t* out, int L, int dim) {
  const int chunks = dim >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)L * chunks; i += (long)gridDim.x * blockDim.x) {
    const int l = (int)(i / chunks), c = (int)(i % chunks);
    *reinterpret_cast<u32x4*>(out + (size_t)l * dim + c * 8) = *reinterpret_cast<const u32x4*>(emb + (size_t)ids[l] * dim + c * 8);
  }
}

// scores fp32 [H][L][L] -> P bf16: attn = bf16(bf16(q.k) + bias); softmax in fp32; bias = per-layer embedding of the
// relative-position bucket, or finfo(bf16).min for padded keys (t5.py:104-116)
__global__ __launch_bounds__(256) void t5_softmax_kernel(const float* sc, const bf16_t* pos_emb, const int* bucket, const int* mask,
                                                         bf16_t* p, int H, int L) {
  const int row = blockIdx.x, h = row / L, i = row - h * L;
  const float* sp = sc + (size_t)row * L;
  bf16_t* pp = p + (size_t)row * L;
  __shared__ float red[4];
  const float kMin = -3.3895313892515355e38f;   // torch.finfo(torch.bfloat16).min
  auto val = [&](int j) {
    const float b = mask[j] ? bf2f(pos_emb[bucket[j - i + L - 1] * H + h]) : kMin;
    return rbf(rbf(sp[j]) + b);
  };
  float mx = -INFINITY;
  for (int j = threadIdx.x; j < L; j += 256) mx = fmaxf(mx, val(j));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int j = threadIdx.x; j < L; j += 256) sum += __expf(val(j) - mx);
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
  for (int j = threadIdx.x; j < L; j += 256) pp[j] = f2bf(__expf(val(j) - mx) * inv);
}

// v [L][H*c] (head h at column h*c) -> vt [H][c][L]
__global__ void t5_transpose_kernel(const bf16_t* v, int ld, bf16_t* vt, int L, int c) {
  __shared__ bf16_t tile[32][33];
  const int h = blockIdx.z, r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, cc = c0 + tx;
    tile[k][tx] = (r < L && cc < c) ? v[(size_t)r * ld + h * c + cc] : (bf16_t)0;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int cc = c0 + k, r = r0 + tx;
    if (cc < c && r < L) vt[((size_t)h * c + cc) * L + r] = tile[tx][k];
  }
}

// h = fc1 * GELU(gate), every tensor op of t5.py:45-49,139 rounding to bf16
__global__ void t5_gated_kernel(bf16_t* f, const bf16_t* g, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float x = bf2f(g[i]);
    const float p3 = rbf(x * x * x);
    const float u = rbf(x + rbf(0.044715f * p3));
    const float th = rbf(tanhf(rbf(0.7978845608028654f * u)));
    const float gl = rbf(rbf(0.5f * x) * rbf(1.0f + th));
    f[i] = f2bf(bf2f(f[i]) * gl);
  }
}

__global__ void t5_zero_pad_kernel(bf16_t* out, const int* mask, int L, int dim) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)L * dim; i += (long)gridDim.x * blockDim.x)
    if (!mask[i / dim]) out[i] = 0;
}

inline int grid_for(size_t n) {
  size_t g = (n + 255) / 256;
  return (int)(g > 8192 ? 8192 : (g == 0 ? 1 : g));
}

struct Carve {
  char* base;
  size_t off = 0;
  void* take(size_t bytes) {
    void* p = base + off;
    off += (bytes + 255) & ~(size_t)255;
    return p;
  }
};
struct Ws {
  bf16_t *x, *xn, *qkv, *p, *vt, *attn, *g, *f;
  float* sc;
  size_t bytes;
};
Ws carve(const MmplT5Config& c, void* base) {
  Carve k{(char*)base};
  const size_t L = c.text_len;
  Ws w;
  w.x = (bf16_t*)k.take(L * c.dim * 2);
  w.xn = (bf16_t*)k.take(L * c.dim * 2);
  w.qkv = (bf16_t*)k.take(L * 3 * c.dim_attn * 2);
  w.sc = (float*)k.take((size_t)c.num_heads * L * L * 4);
  w.p = (bf16_t*)k.take((size_t)c.num_heads * L * L * 2);
  w.vt = (bf16_t*)k.take((size_t)c.dim_attn * L * 2);
  w.attn = (bf16_t*)k.take(L * c.dim_attn * 2);
  w.g = (bf16_t*)k.take(L * c.dim_ffn * 2);
  w.f = (bf16_t*)k.take(L * c.dim_ffn * 2);
  w.bytes = k.off;
  return w;
}

}  // namespace

struct MmplT5 {
  MmplT5Config cfg;
  std::vector<const bf16_t*> w;
};
enum { T_EMB, T_NORM, TG };
enum { TL_N1, TL_QKV, TL_O, TL_POS, TL_N2, TL_GATE, TL_FC1, TL_FC2, TLN };

extern "C" {

int mmpl_t5_num_weights(const MmplT5Config* c) { return TG + c->num_layers * TLN; }

int mmpl_t5_create(const MmplT5Config* c, MmplT5** out) {
  if (!c || !out) return mmpl_set_error("mmpl_t5_create", "null argument");
  if (c->dim % 128 || c->dim > 5120 || c->dim_attn % c->num_heads || (c->dim_attn / c->num_heads) % 64 || c->dim_ffn % 64 ||
      c->text_len % 64 || c->dim_attn % 64)
    return mmpl_set_error("mmpl_t5_create", "unsupported dims (dim % 128, head_dim % 64, text_len % 64)");
  MmplT5* h = new MmplT5();
  h->cfg = *c;
  *out = h;
  return 0;
}
void mmpl_t5_destroy(MmplT5* h) { delete h; }

int mmpl_t5_bind_weights(MmplT5* h, const void* const* ptrs, int n) {
  if (!h || !ptrs || n != mmpl_t5_num_weights(&h->cfg)) return mmpl_set_error("mmpl_t5_bind_weights", "wrong pointer count");
  h->w.resize(n);
  for (int i = 0; i < n; ++i) {
    if (!ptrs[i]) return mmpl_set_error("mmpl_t5_bind_weights", "null weight pointer");
    h->w[i] = (const bf16_t*)ptrs[i];
  }
  return 0;
}

size_t mmpl_t5_workspace_bytes(const MmplT5* h) { return carve(h->cfg, nullptr).bytes; }

int mmpl_t5_encode(MmplT5* h, const int* ids, const int* mask, const int* bucket, void* out, void* workspace, size_t workspace_bytes,
                   mmpl_stream_t stream) {
  if (!h || h->w.empty()) return mmpl_set_error("mmpl_t5_encode", "weights not bound");
  const MmplT5Config& c = h->cfg;
  Ws w = carve(c, workspace);
  if (workspace_bytes < w.bytes) return mmpl_set_error("mmpl_t5_encode", "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int L = c.text_len, d = c.dim, da = c.dim_attn, df = c.dim_ffn, H = c.num_heads, hc = da / H;
  hipError_t err = hipSuccess;
  auto chk = [&](hipError_t e) { if (e != hipSuccess && err == hipSuccess) err = e; };
  auto gemm = [&](const bf16_t* A, int lda, const bf16_t* W, int ldw, void* C, int ldc, int M, int N, int K, int epi, const bf16_t* res,
                  int ldres, int batch = 0, long sA = 0, long sW = 0, long sC = 0) {
    GemmArgs g{A, lda, W, ldw, nullptr, (bf16_t*)C, ldc, M, N, K, epi, res, ldres, nullptr, 0, 1, 1.0f, 0, batch, sA, sW, sC};
    chk(mmpl_launch_gemm(g, s));
  };
  auto norm = [&](const bf16_t* x, const bf16_t* wt, bf16_t* y) {
    chk(hipMemcpyAsync(y, x, (size_t)L * d * 2, hipMemcpyDeviceToDevice, s));
    chk(mmpl_launch_rmsnorm(y, d, wt, L, d, c.eps, s));
  };
  hipLaunchKernelGGL(t5_gather_kernel, dim3(grid_for((size_t)L * d / 8)), dim3(256), 0, s, ids, h->w[T_EMB], w.x, L, d);
  for (int l = 0; l < c.num_layers; ++l) {
    const bf16_t* const* lw = &h->w[TG + l * TLN];
    norm(w.x, lw[TL_N1], w.xn);
    gemm(w.xn, d, lw[TL_QKV], d, w.qkv, 3 * da, L, 3 * da, d, EPI_BIAS, nullptr, 0);
    // scores[h] = q_h . k_h^T (fp32), one batched launch over the heads
    gemm(w.qkv, 3 * da, w.qkv + da, 3 * da, w.sc, L, L, L, hc, EPI_F32_SCALE, nullptr, 0, H, hc, hc, (long)L * L);
    hipLaunchKernelGGL(t5_softmax_kernel, dim3(H * L), dim3(256), 0, s, w.sc, lw[TL_POS], bucket, mask, w.p, H, L);
    hipLaunchKernelGGL(t5_transpose_kernel, dim3((L + 31) / 32, (hc + 31) / 32, H), dim3(256), 0, s, w.qkv + 2 * da, 3 * da, w.vt, L, hc);
    gemm(w.p, L, w.vt, L, w.attn, da, L, hc, L, EPI_BIAS, nullptr, 0, H, (long)L * L, (long)hc * L, hc);
    gemm(w.attn, da, lw[TL_O], da, w.x, d, L, d, da, EPI_RES, w.x, d);
    norm(w.x, lw[TL_N2], w.xn);
    gemm(w.xn, d, lw[TL_GATE], d, w.g, df, L, df, d, EPI_BIAS, nullptr, 0);
    gemm(w.xn, d, lw[TL_FC1], d, w.f, df, L, df, d, EPI_BIAS, nullptr, 0);
    hipLaunchKernelGGL(t5_gated_kernel, dim3(grid_for((size_t)L * df)), dim3(256), 0, s, w.f, w.g, (size_t)L * df);
    gemm(w.f, df, lw[TL_FC2], df, w.x, d, L, d, df, EPI_RES, w.x, d);
  }
  norm(w.x, h->w[T_NORM], (bf16_t*)out);
  hipLaunchKernelGGL(t5_zero_pad_kernel, dim3(grid_for((size_t)L * d)), dim3(256), 0, s, (bf16_t*)out, mask, L, d);
  chk(hipGetLastError());
  if (err != hipSuccess) return mmpl_set_error("mmpl_t5_encode", hipGetErrorString(err));
  return 0;
}

}  // extern "C"
